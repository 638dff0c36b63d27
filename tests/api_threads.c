/* tests/api_threads.c -- TEST INFRASTRUCTURE: many threads running the reference's public API sequence, unchanged, against
 * libj40hip.so. Every thread does what dj40.c does (/root/reference/dj40.c:29-50): j40_from_memory -> j40_output_format ->
 * j40_next_frame -> j40_current_frame -> j40_frame_pixels_u8x4 -> rows -> j40_error -> j40_free, image after image. With
 * several threads inside the API at once the library serves them from one pipeline (j40_amd/csrc/api.cpp); the pixels must be
 * those of a lone call. The first decode of every file is kept (and optionally written out raw, for the comparison with the
 * reference's pixels in tests/test_api_threads.py); every later decode of that file must equal it byte for byte.
 *
 *   api_threads <threads> <images per thread> [--dump DIR] [--warm N] [--verify-every K] file.jxl [file.jxl ...]
 *
 * --verify-every K: compare only every K-th decode of a thread with the file's first decode (the comparison reads 133 MB per 8K image:
 * a throughput measurement on a box with few CPUs should not spend them there; the tests verify every decode)
 *
 * prints one JSON line: wall time of the timed part, Mpixel/s, the latency of the calls, errors, mismatches. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "j40.h"

typedef struct { char *path; void *data; size_t size; uint8_t *first; int32_t w, h; pthread_mutex_t m; char err[8]; } file_t;

static file_t *files; static int nfiles, nthreads, per_thread, warm, verify_every = 1;
static const char *dump_dir;
static pthread_barrier_t start_line;
static double *latency_ms; static long mismatches, errors; static double pixels_done;
static pthread_mutex_t tally = PTHREAD_MUTEX_INITIALIZER;

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }

/* one image through the public API; returns 0 when it decoded and agreed with the file's first decode */
static int decode_one(file_t *f, double *ms, int verify) {
	j40_image image;
	const double t0 = now_ms();
	j40_from_memory(&image, f->data, f->size, NULL);
	j40_output_format(&image, J40_RGBA, J40_U8X4);
	int bad = 0;
	if (j40_next_frame(&image)) {
		j40_frame frame = j40_current_frame(&image);
		j40_pixels_u8x4 px = j40_frame_pixels_u8x4(&frame, J40_RGBA);
		*ms = now_ms() - t0;
		if (((uintptr_t) px.data & 31) || (px.stride_bytes & 31) || px.stride_bytes < px.width * 4 + 1) bad = 1;   /* the reference's layout (j40.h:1061-1065, 7939) */
		pthread_mutex_lock(&f->m);
		if (!f->first) {
			f->w = px.width; f->h = px.height;
			f->first = malloc((size_t) px.width * 4 * (size_t) px.height);
			for (int32_t y = 0; y < px.height; ++y) memcpy(f->first + (size_t) y * px.width * 4, j40_row_u8x4(px, y), (size_t) px.width * 4);
			pthread_mutex_unlock(&f->m);
		} else {
			pthread_mutex_unlock(&f->m);
			if (!verify) ;
			else if (px.width != f->w || px.height != f->h) bad = 1;
			else for (int32_t y = 0; y < px.height && !bad; ++y) if (memcmp(f->first + (size_t) y * px.width * 4, j40_row_u8x4(px, y), (size_t) px.width * 4)) bad = 1;
		}
		pthread_mutex_lock(&tally); pixels_done += (double) px.width * px.height; mismatches += bad; pthread_mutex_unlock(&tally);
	} else *ms = now_ms() - t0;
	if (j40_error(&image)) {
		j40_err e = j40_error(&image);
		pthread_mutex_lock(&f->m);
		f->err[0] = (char) (e >> 24); f->err[1] = (char) (e >> 16); f->err[2] = (char) (e >> 8); f->err[3] = (char) e; f->err[4] = 0;
		pthread_mutex_unlock(&f->m);
		pthread_mutex_lock(&tally); ++errors; pthread_mutex_unlock(&tally);
		bad = 1;
	}
	j40_free(&image);
	return bad;
}

static void *thread_main(void *arg) {
	const int t = (int) (intptr_t) arg;
	double ms;
	for (int i = 0; i < warm; ++i) decode_one(&files[(t + i) % nfiles], &ms, 1);
	pthread_barrier_wait(&start_line);
	pthread_barrier_wait(&start_line);   /* (the main thread resets the tallies and starts the clock in between) */
	for (int i = 0; i < per_thread; ++i) { decode_one(&files[(t + warm + i) % nfiles], &ms, (i + t) % verify_every == 0); latency_ms[(size_t) t * per_thread + i] = ms; }
	return NULL;
}

static int cmp_double(const void *a, const void *b) { const double x = *(const double *) a, y = *(const double *) b; return x < y ? -1 : x > y; }

int main(int argc, char **argv) {
	if (argc < 4) { fprintf(stderr, "usage: %s threads images_per_thread [--dump DIR] [--warm N] file.jxl ...\n", argv[0]); return 2; }
	nthreads = atoi(argv[1]); per_thread = atoi(argv[2]);
	int a = 3;
	while (a + 1 < argc && argv[a][0] == '-') {
		if (!strcmp(argv[a], "--dump")) dump_dir = argv[a + 1];
		else if (!strcmp(argv[a], "--warm")) warm = atoi(argv[a + 1]);
		else if (!strcmp(argv[a], "--verify-every")) verify_every = atoi(argv[a + 1]) > 0 ? atoi(argv[a + 1]) : 1;
		else { fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
		a += 2;
	}
	nfiles = argc - a;
	if (nthreads < 1 || per_thread < 1 || nfiles < 1) return 2;
	files = calloc((size_t) nfiles, sizeof *files);
	for (int i = 0; i < nfiles; ++i) {
		file_t *f = &files[i];
		f->path = argv[a + i];
		FILE *fp = fopen(f->path, "rb");
		if (!fp) { perror(f->path); return 2; }
		fseek(fp, 0, SEEK_END); f->size = (size_t) ftell(fp); fseek(fp, 0, SEEK_SET);
		f->data = malloc(f->size ? f->size : 1);
		if (fread(f->data, 1, f->size, fp) != f->size) { perror(f->path); return 2; }
		fclose(fp);
		pthread_mutex_init(&f->m, NULL);
	}
	latency_ms = calloc((size_t) nthreads * per_thread, sizeof *latency_ms);
	pthread_barrier_init(&start_line, NULL, (unsigned) nthreads + 1);
	pthread_t *th = calloc((size_t) nthreads, sizeof *th);
	for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, thread_main, (void *) (intptr_t) t);
	pthread_barrier_wait(&start_line);
	const long warm_errors = errors, warm_mismatches = mismatches;
	pixels_done = 0;
	const double t0 = now_ms();
	pthread_barrier_wait(&start_line);
	for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
	const double seconds = (now_ms() - t0) / 1e3;
	const size_t n = (size_t) nthreads * per_thread;
	qsort(latency_ms, n, sizeof *latency_ms, cmp_double);
	if (dump_dir) for (int i = 0; i < nfiles; ++i) if (files[i].first) {
		char path[4096];
		snprintf(path, sizeof path, "%s/%d_%dx%d.rgba", dump_dir, i, files[i].w, files[i].h);
		FILE *fp = fopen(path, "wb");
		if (!fp || fwrite(files[i].first, 1, (size_t) files[i].w * 4 * files[i].h, fp) != (size_t) files[i].w * 4 * files[i].h) { perror(path); return 2; }
		fclose(fp);
	}
	printf("{\"threads\": %d, \"images\": %zu, \"verify_every\": %d, \"files\": %d, \"seconds\": %.4f, \"mpixels_per_s\": %.1f, \"latency_ms\": {\"min\": %.2f, \"median\": %.2f, \"p90\": %.2f, \"max\": %.2f}, \"errors\": %ld, \"mismatches\": %ld, \"warm_errors\": %ld, \"warm_mismatches\": %ld, \"file_errors\": [",
		nthreads, n, verify_every, nfiles, seconds, pixels_done / seconds / 1e6, latency_ms[0], latency_ms[n / 2], latency_ms[n * 9 / 10], latency_ms[n - 1], errors - warm_errors, mismatches - warm_mismatches, warm_errors, warm_mismatches);
	for (int i = 0; i < nfiles; ++i) printf("%s\"%s\"", i ? ", " : "", files[i].err);
	printf("]}\n");
	return 0;
}
