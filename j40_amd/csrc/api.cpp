// j40_amd/csrc/api.cpp -- the public j40 C API (include/j40.h) on top of the host parser and the HIP
// hot path. Handle states, magic numbers, error strings and the placeholder image follow the
// reference's observable behaviour (j40.h:7970-8119, 8245-8477).
#define J40_API __attribute__((visibility("default")))
#include <algorithm>
#include <sys/stat.h>
#include "../../include/j40.h"
#include "capi.hpp"
#include <atomic>
#include <new>
#include <thread>
#include <mutex>
#include <memory>
#include <condition_variable>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace j40hip;

namespace {

enum : uint32_t {
	IMAGE_MAGIC = 0x7867ae21u, IMAGE_ERR_MAGIC = 0xb26a48aau, IMAGE_OPEN_ERR_MAGIC = 0x02c2eb6du,
	FRAME_MAGIC = 0x08a296b3u, FRAME_ERR_MAGIC = 0x16351564u, INNER_MAGIC = 0x5009e1c4u,
};

enum Origin { O_NONE = 0, O_NEXT, O_from_file, O_from_memory, O_output_format, O_next_frame, O_current_frame, O_frame_pixels, O_error_string, O_free, O_LAST_ALT = O_from_memory };
const char *const ORIGIN_NAMES[] = {"(unknown)", nullptr, "from_file", "from_memory", "output_format", "next_frame", "current_frame", "frame_pixels_*", "error_string", "free"};

const struct { const char *err, *msg; } ERROR_STRINGS[] = {
	{"Upt0", "`path` parameter is NULL"}, {"Ubf0", "`buf` parameter is NULL"}, {"Uch?", "Bad `channel` parameter"},
	{"Ufm?", "Bad `format` parameter"}, {"Uof?", "Bad `channel` and `format` combination"}, {"Urnd", "Frame is not yet rendered"},
	{"Ufre", "Trying to reuse already freed image"}, {"!mem", "Out of memory"}, {"!jxl", "The JPEG XL signature is not found"},
	{"open", "Failed to open file"}, {"bigg", "Image dimensions are too large to handle"}, {"flen", "File is too lengthy to handle"},
	{"shrt", "Premature end of file"}, {"slim", "Image size limit reached"}, {"elim", "Extra channel number limit reached"},
	{"xlim", "Modular transform limit reached"}, {"tlim", "Meta-adaptive tree size or depth limit reached"},
	{"plim", "ICC profile length limit reached"}, {"fbpp", "Given bits per pixel value is disallowed"},
	{"fblk", "Black extra channel is disallowed"}, {"fm32", "32-bit buffers for modular encoding are disallowed"},
	{"TODO", "Unimplemented feature encountered"}, {"TEST", "Testing-only error occurred"},
	{"!gpu", "No usable HIP device (the hot path has no CPU fallback)"},
};

uint32_t code4(const char *s) { return ((uint32_t) (uint8_t) s[0] << 24) | ((uint32_t) (uint8_t) s[1] << 16) | ((uint32_t) (uint8_t) s[2] << 8) | (uint32_t) (uint8_t) s[3]; }

} // namespace

struct j40__inner {
	uint32_t magic;
	int origin;
	j40_err err;
	int saved_errno;
	char errbuf[256];
	void *buf; size_t size; j40_memory_free_func freefunc;   // borrowed input
	void *owned;                                             // file contents (from_file), read by the first j40_next_frame
	FILE *fp;                                                // from_file: the open file until then
	j40hip_frame *frame;
	int decoded, rendered;
	uint8_t *pixels; size_t pixels_bytes; int32_t width, height, stride_bytes;   // image-owned plane: pinned host memory from the library's pool
};

namespace {

j40_err set_alt_magic(j40_err err, int saved_errno, int origin, j40_image *image) {  // j40.h:8083
	if (err == code4("open")) { image->magic = IMAGE_OPEN_ERR_MAGIC ^ (uint32_t) origin; image->u.saved_errno = saved_errno; return err; }
	image->magic = IMAGE_ERR_MAGIC ^ (uint32_t) origin;
	return image->u.err = err;
}

j40_err check_image(j40_image *image, int neworigin, j40__inner **out) {  // j40.h:8103
	*out = nullptr;
	if (!image) return code4("Uim0");
	if (image->magic != IMAGE_MAGIC) {
		uint32_t origin = image->magic ^ IMAGE_ERR_MAGIC;
		if (0 < origin && origin <= O_LAST_ALT) {
			if (origin == O_NEXT && neworigin) image->magic = IMAGE_ERR_MAGIC ^ (uint32_t) neworigin;
			return image->u.err;
		}
		origin = image->magic ^ IMAGE_OPEN_ERR_MAGIC;
		if (0 < origin && origin <= O_LAST_ALT) return code4("open");
		return code4("Uim?");
	}
	if (!image->u.inner || image->u.inner->magic != INNER_MAGIC) return code4("Uim?");
	*out = image->u.inner;
	return image->u.inner->err;
}

void free_inner(j40__inner *inner) {
	if (inner->frame) j40hip_frame_free(inner->frame);
	if (inner->freefunc && inner->buf) inner->freefunc(inner->buf);
	free(inner->owned);
	if (inner->fp) fclose(inner->fp);
	if (inner->pixels) j40hip_pinned_release(inner->pixels, inner->pixels_bytes);
	inner->magic = 0;
	free(inner);
}

j40__inner *new_inner() {
	j40__inner *inner = (j40__inner *) calloc(1, sizeof(j40__inner));
	if (inner) inner->magic = INNER_MAGIC;
	return inner;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The image-owned pixel plane in the reference's layout: rows of stride = 32 * ceil((4 * width + 1) / 32) bytes, 32-byte aligned
// (forced padding, j40.h:1061-1065, 7939). Pinned host memory, so that the copy from the device runs at the link's rate.
uint32_t make_plane(j40__inner *inner, int64_t width, int64_t height) {
	const int64_t stride = (width * 4 + 1 + 31) / 32 * 32;
	if (width >= INT32_MAX / 4 || stride > INT32_MAX || height > INT32_MAX) return code4("bigg");
	inner->width = (int32_t) width; inner->height = (int32_t) height; inner->stride_bytes = (int32_t) stride;
	inner->pixels_bytes = (size_t) stride * (size_t) height;
	inner->pixels = (uint8_t *) j40hip_pinned_acquire(inner->pixels_bytes);
	return inner->pixels ? 0 : code4("!mem");
}

void *serve_alloc(void *ctx, int64_t width, int64_t height, size_t *stride_bytes) {   // (j40hip_output_alloc: called on a pipeline thread)
	j40__inner *inner = (j40__inner *) ctx;
	if (make_plane(inner, width, height)) return nullptr;
	*stride_bytes = (size_t) inner->stride_bytes;
	return inner->pixels;
}

// How j40_next_frame decodes. The reference decodes on the calling thread and so does the LATENCY path here: host parse, upload,
// one section per wavefront, copy back -- the shortest way for one image. When several threads are inside the API at once their
// images are handed to the process-wide pipeline of the device instead (SERVING: j40hip_pipeline_run, device/pipeline.hip) and share
// its batches -- one entropy launch for all of them, the copies back at the link's rate; the calling thread sleeps meanwhile.
// J40HIP_SERVE=1 always serves, =0 never does; default: serve when another call is in progress, or was within the last 200 ms.
std::atomic<int> g_inside{0};
std::atomic<int64_t> g_last_overlap_ms{-1000000};
int serve_policy() { static const int v = [] { const char *e = getenv("J40HIP_SERVE"); return !e || !*e ? 2 : atoi(e) != 0 ? 1 : 0; }(); return v; }
int device_index() { const char *e = getenv("J40HIP_DEVICE"); return e ? atoi(e) : 0; }

// ---- j40_from_file's source (the reference's refillable file source and backing buffer, j40.h:1220-1386,
// 1676-1812). j40_from_file opens the file and reads nothing (j40.h:8342-8361); the bytes are read by the first j40_next_frame, as
// the reference's are (j40__file_source_read, j40.h:1241-1256: fread until the end of the file; a failing read raises `read` with
// the errno kept; what a truncated file lacks surfaces as `shrt` from whoever needs the bytes). A regular file is read by a thread
// of its own WHILE the calling thread parses it (j40hip_frame_parse_streamed: headers and TOC on the prefix that has arrived, LfGlobal,
// HfGlobal and each LfGroup section as its bytes are due): the host's part of the decode overlaps the arrival of the pass-group
// sections, which are most of the file and which only the device reads. J40HIP_STREAM=0, a source without a size (a pipe) or a
// call that is served by the pipeline: the whole file first.
struct FileSource {
	FILE *fp = nullptr; uint8_t *data = nullptr; size_t size = 0;
	std::mutex m; std::condition_variable cv;
	size_t have = 0; bool done = false; bool failed = false; int saved_errno = 0;
	std::thread reader;
	void run() {
		const int saved = errno;
		size_t at = 0;
		while (at < size) {
			errno = 0;
			const size_t n = fread(data + at, 1, std::min<size_t>(size - at, (size_t) 1 << 20), fp);
			if (n == 0) { if (!feof(fp)) { std::lock_guard<std::mutex> lock(m); failed = true; saved_errno = errno; } break; }   // (a file that ends early: what is missing stays zero, a truncated stream)
			at += n;
			{ std::lock_guard<std::mutex> lock(m); have = at; }
			cv.notify_all();
		}
		errno = saved;
		{ std::lock_guard<std::mutex> lock(m); done = true; }
		cv.notify_all();
	}
	static void need(void *ctx, size_t upto) { FileSource *s = (FileSource *) ctx; std::unique_lock<std::mutex> lock(s->m); s->cv.wait(lock, [&] { return s->done || s->have >= upto; }); }
	static size_t have_now(void *ctx) { FileSource *s = (FileSource *) ctx; std::lock_guard<std::mutex> lock(s->m); return s->done ? s->size : s->have; }
};
int stream_policy() { static const int v = [] { const char *e = getenv("J40HIP_STREAM"); return !e || !*e || atoi(e) != 0 ? 1 : 0; }(); return v; }

// the whole file, now (the served path, pipes, J40HIP_STREAM=0); 0 or the error code
uint32_t read_whole_file(j40__inner *inner) {
	FILE *fp = inner->fp;
	inner->fp = nullptr;
	const int saved = errno;
	errno = 0;
	size_t cap = 1 << 16, size = 0;
	struct stat sb;
	if (fstat(fileno(fp), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) cap = (size_t) sb.st_size + 1;   // (one allocation; a source that cannot say grows its buffer)
	uint8_t *data = (uint8_t *) malloc(cap);
	uint32_t ferr = data ? 0 : code4("!mem");
	while (!ferr) {
		if (size == cap) {
			uint8_t *more = (uint8_t *) realloc(data, cap * 2);
			if (!more) { ferr = code4("!mem"); break; }
			data = more; cap *= 2;
		}
		const size_t n = fread(data + size, 1, cap - size, fp);
		if (n > 0) { size += n; continue; }
		if (!feof(fp)) { inner->saved_errno = errno; ferr = code4("read"); }
		break;
	}
	fclose(fp);
	errno = saved;
	if (ferr) { free(data); return ferr; }
	inner->owned = data; inner->buf = data; inner->size = size; inner->freefunc = nullptr;
	return 0;
}

// the whole decode: RGBA into the image-owned plane
j40_err advance(j40__inner *inner, int origin) {
	if (inner->decoded) return 0;
	struct Inside { int n; Inside() : n(++g_inside) {} ~Inside() { --g_inside; } } inside;
	const int policy = serve_policy();
	const int64_t now = (int64_t) now_ms();
	if (inside.n > 1) g_last_overlap_ms.store(now);
	const bool serve = policy == 1 || (policy == 2 && (inside.n > 1 || now - g_last_overlap_ms.load() < 200));
	static const bool timing = getenv("J40HIP_API_TIMING") != nullptr;
	uint32_t err = 0;
	std::unique_ptr<FileSource> src;
	if (inner->fp) {
		struct stat sb;
		if (!serve && stream_policy() && fstat(fileno(inner->fp), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
			src.reset(new (std::nothrow) FileSource());
			uint8_t *data = src ? (uint8_t *) calloc((size_t) sb.st_size, 1) : nullptr;
			if (!data) { src.reset(); err = code4("!mem"); }
			else {
				src->fp = inner->fp; src->data = data; src->size = (size_t) sb.st_size;
				inner->fp = nullptr; inner->owned = data; inner->buf = data; inner->size = src->size; inner->freefunc = nullptr;
				try { src->reader = std::thread([s = src.get()] { s->run(); }); }
				catch (const std::exception &) { src->run(); }   // (no thread to be had: read it here)
			}
		} else err = read_whole_file(inner);
		if (err) { inner->origin = origin; inner->err = err; return err; }
	}
	// (whatever happens below, the reader has finished and the file is closed before this call returns)
	struct Joined { FileSource *s; ~Joined() { if (s) { if (s->reader.joinable()) s->reader.join(); fclose(s->fp); } } } joined{src.get()};
	auto read_failed = [&]() -> bool {   // the file could not be read to its end: `read`, whatever the parse made of what there was
		if (!src) return false;
		FileSource::need(src.get(), src->size);
		if (!src->failed) return false;
		inner->saved_errno = src->saved_errno; inner->origin = origin; inner->err = code4("read");
		return true;
	};
	if (serve) {
		j40hip_pipeline *p = j40hip_serve_pipeline(device_index(), &err);
		if (p) err = j40hip_pipeline_run(p, inner->buf, inner->size, serve_alloc, inner);
		if (err) { inner->origin = origin; inner->err = err; return err; }
		inner->decoded = 1;
		return 0;
	}
	const double t0 = now_ms();
	double t1 = t0, t2 = t0, t3 = t0;
	// (LfGroup sections are independent: an 8K frame has twelve; their tail -- dequantisation, smoothing, LLF coefficients -- runs on
	// the device at upload, flags = 1)
	// (no more threads than the container's CPU quota: a process over its quota has all its threads throttled, the HIP runtime's too;
	// frames with fewer LfGroups and groups than that get a smaller team: parse_frame, build_vardct_plan)
	static const int parse_threads = [] { const char *e = getenv("J40HIP_PARSE_THREADS"); return e && atoi(e) > 0 ? atoi(e) : std::max(1, std::min(12, j40hip_cpu_quota())); }();
	inner->frame = src ? j40hip_frame_parse_streamed(inner->buf, inner->size, parse_threads, 1u, FileSource::need, FileSource::have_now, src.get(), &err)
	                   : j40hip_frame_parse_ex(inner->buf, inner->size, parse_threads, 1u, &err);
	if (read_failed()) return inner->err;   // (also: the rest of the file is there from here on -- the upload copies the codestream)
	t1 = now_ms();
	if (!err && j40hip_device_count() <= device_index()) err = code4("!gpu");   // (before the plane: pinned memory needs the device too)
	if (!err) {
		int64_t info[32];
		j40hip_frame_info(inner->frame, info);
		err = make_plane(inner, info[0], info[1]);
	}
	t2 = now_ms();
	if (!err) err = j40hip_frame_upload(inner->frame, device_index());
	t3 = now_ms();
	if (!err) err = j40hip_frame_decode_to_host(inner->frame, inner->pixels, (size_t) inner->stride_bytes);
	if (!err) err = j40hip_frame_after_frame_status(inner->frame);   // bytes behind the frame (j40__no_more_bytes, j40.h:8215)
	if (timing) fprintf(stderr, "[j40 api] %d x %d: parse %.2f ms, plane %.2f ms, plan + upload %.2f ms, decode + copy back %.2f ms\n", inner->width, inner->height, t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
	if (err) { inner->origin = origin; inner->err = err; return err; }
	inner->decoded = 1;
	return 0;
}

} // namespace

extern "C" {

j40_err j40_error(const j40_image *image) {
	j40__inner *inner;
	return check_image((j40_image *) image, O_NONE, &inner);
}

const char *j40_error_string(const j40_image *image) {  // j40.h:8251
	static char static_errbuf[256];
	uint32_t origin = O_NONE; j40_err err = 0; char *buf = nullptr; int saved_errno = 0; bool corrupted = false;
	if (!image) { snprintf(static_errbuf, sizeof static_errbuf, "`image` parameter is NULL during j40_error_string"); return static_errbuf; }
	if (image->magic == IMAGE_MAGIC) {
		if (image->u.inner && image->u.inner->magic == INNER_MAGIC) { origin = (uint32_t) image->u.inner->origin; err = image->u.inner->err; buf = image->u.inner->errbuf; saved_errno = image->u.inner->saved_errno; }
		else corrupted = true;
	} else {
		origin = image->magic ^ IMAGE_ERR_MAGIC;
		if (0 < origin && origin <= O_LAST_ALT) { err = image->u.err; buf = static_errbuf; if (origin == O_NEXT) origin = O_error_string; }
		else {
			origin = image->magic ^ IMAGE_OPEN_ERR_MAGIC;
			if (0 < origin && origin <= O_LAST_ALT) { err = code4("open"); buf = static_errbuf; saved_errno = image->u.saved_errno; }
			else corrupted = true;
		}
	}
	if (corrupted) { snprintf(static_errbuf, sizeof static_errbuf, "`image` parameter is found corrupted during j40_error_string"); return static_errbuf; }
	const char *msg = nullptr;
	for (const auto &e : ERROR_STRINGS) if (err == code4(e.err)) { msg = e.msg; break; }
	if (!msg) snprintf(buf, 256, "Decoding failed (%c%c%c%c) during j40_%s", err >> 24 & 0xff, err >> 16 & 0xff, err >> 8 & 0xff, err & 0xff, ORIGIN_NAMES[origin]);
	else if (saved_errno) snprintf(buf, 256, "%s during j40_%s: %s", msg, ORIGIN_NAMES[origin], strerror(saved_errno));
	else snprintf(buf, 256, "%s during j40_%s", msg, ORIGIN_NAMES[origin]);
	return buf;
}

j40_err j40_from_memory(j40_image *image, void *buf, size_t size, j40_memory_free_func freefunc) {
	if (!image) return code4("Uim0");
	if (!buf) return set_alt_magic(code4("Ubf0"), 0, O_from_memory, image);
	j40__inner *inner = new_inner();
	if (!inner) return set_alt_magic(code4("!mem"), 0, O_from_memory, image);
	inner->buf = buf; inner->size = size; inner->freefunc = freefunc;
	image->magic = IMAGE_MAGIC; image->u.inner = inner;
	return 0;
}

j40_err j40_from_file(j40_image *image, const char *path) {
	if (!image) return code4("Uim0");
	if (!path) return set_alt_magic(code4("Upt0"), 0, O_from_file, image);
	j40__inner *inner = new_inner();
	if (!inner) return set_alt_magic(code4("!mem"), 0, O_from_file, image);
	int saved = errno;
	errno = 0;
	FILE *fp = fopen(path, "rb");
	if (!fp) { int e = errno; errno = saved; free_inner(inner); return set_alt_magic(code4("open"), e, O_from_file, image); }
	errno = saved;
	inner->fp = fp;   // (read by j40_next_frame: j40.h:8354 opens the source and reads nothing either)
	image->magic = IMAGE_MAGIC; image->u.inner = inner;
	return 0;
}

j40_err j40_output_format(j40_image *image, int32_t channel, int32_t format) {
	j40__inner *inner;
	j40_err err = check_image(image, O_output_format, &inner);
	if (err) return err;
	if (channel != J40_RGBA) { inner->origin = O_output_format; return inner->err = code4("Uch?"); }
	if (format != J40_U8X4) { inner->origin = O_output_format; return inner->err = code4("Ufm?"); }
	return 0;
}

int j40_next_frame(j40_image *image) {
	j40__inner *inner;
	if (check_image(image, O_next_frame, &inner)) return 0;
	if (advance(inner, O_next_frame)) return 0;
	if (inner->rendered) return 0;  // single-frame images: the second call reports "no more frames" (j40.h:8390)
	inner->rendered = 1;
	return 1;
}

j40_frame j40_current_frame(j40_image *image) {
	j40__inner *inner;
	j40_frame frame;
	j40_err err = check_image(image, O_current_frame, &inner);
	frame.magic = FRAME_ERR_MAGIC; frame.reserved = 0; frame.inner = inner;
	if (err) return frame;
	if (!inner->rendered) { if (!j40_next_frame(image)) { if (inner->err) return frame; } }
	frame.magic = FRAME_MAGIC;
	return frame;
}

j40_pixels_u8x4 j40_frame_pixels_u8x4(const j40_frame *frame, int32_t channel) {
	// placeholder shown on error: "ERR" on red, 21 x 7 (same picture as j40.h:8432-8441); 1 = opaque
	static const char *const ERR_ROWS[7] = {
		"111111111111111111111", "100011111111111111111", "101111111111111111111", "100010001000100010001",
		"101110111011101010111", "100010111011100010111", "111111111111111111111"};
	static uint8_t error_pixels[21 * 7 * 4];
	static const bool error_pixels_ready = [] {   // (a function-local static: filled once, also with many threads in here)
		for (int y = 0; y < 7; ++y) for (int x = 0; x < 21; ++x) { uint8_t *p = error_pixels + (y * 21 + x) * 4; p[0] = 255; p[1] = 0; p[2] = 0; p[3] = ERR_ROWS[y][x] == '1' ? 255 : 0; }
		return true;
	}();
	(void) error_pixels_ready;
	const j40_pixels_u8x4 ERROR_PIXELS = {21, 7, 21 * 4, error_pixels};
	if (!frame || frame->magic != FRAME_MAGIC) return ERROR_PIXELS;
	j40__inner *inner = frame->inner;
	if (!inner || inner->magic != INNER_MAGIC) return ERROR_PIXELS;
	if (channel != J40_RGBA) return ERROR_PIXELS;
	if (!inner->rendered) { inner->origin = O_frame_pixels; inner->err = code4("Urnd"); return ERROR_PIXELS; }
	j40_pixels_u8x4 px;
	px.width = inner->width; px.height = inner->height; px.stride_bytes = inner->stride_bytes; px.data = inner->pixels;
	return px;
}

const j40_u8x4 *j40_row_u8x4(j40_pixels_u8x4 pixels, int32_t y) {
	return (const j40_u8x4 *) ((const char *) pixels.data + (size_t) pixels.stride_bytes * (size_t) y);
}

void j40_free(j40_image *image) {
	j40__inner *inner;
	check_image(image, O_free, &inner);
	if (inner) free_inner(inner);
	if (!image) return;
	image->magic = IMAGE_ERR_MAGIC ^ O_NEXT;
	image->u.err = code4("Ufre");
}

} // extern "C"
