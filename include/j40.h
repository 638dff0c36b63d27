/*
 * include/j40.h -- public C API of j40-hip: a drop-in for the public API of lifthrasiir/j40.
 *
 * Same ten functions, struct layouts, constants and error conventions as the reference's header
 * (declarations at /root/reference/j40.h:168-272, implementation j40.h:8245-8477), so a program
 * written against the reference (e.g. its dj40.c:29-50 or extra/j40-fuzz.c:5-14) links against
 * libj40hip.so unchanged. Unlike the reference this is a declarations-only header: the
 * implementation lives in the shared library and runs the per-group hot path on an MI355X.
 *
 * The reference's opt-in macros are accepted and ignored so that existing sources compile as is.
 */
#ifndef J40_HIP_PUBLIC_H_
#define J40_HIP_PUBLIC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef J40_API
#define J40_API
#endif

#define J40_VERSION 2270 /* API level of the reference this header mirrors (j40.h:80) */

/* 0 = no error; otherwise a four-character code packed big-endian (j40.h:171, 482) */
typedef uint32_t j40_err;
#define J40_MIN_RESERVED_ERR (j40_err) (1 << 24)

struct j40__inner;

/* caller-allocated handle; needs no initialisation before j40_from_* (j40.h:174-182) */
typedef struct {
	uint32_t magic;
	union {
		struct j40__inner *inner;
		j40_err err;
		int saved_errno;
	} u;
} j40_image;

/* j40.h:184-188 */
typedef struct {
	uint32_t magic;
	uint32_t reserved;
	struct j40__inner *inner;
} j40_frame;

typedef void (*j40_memory_free_func)(void *data);

#define J40_U8X4 0x0f33 /* j40.h:202 */
#define J40_RGBA 0x1755 /* j40.h:228 */

typedef uint8_t j40_u8x4[4];
typedef float j40_f32x4[4]; /* j40.h:256: declared there ahead of a float API that does not exist yet; kept so that sources naming it compile */

/* j40.h:244-251 */
typedef struct {
	int32_t width, height;
	int32_t stride_bytes;
	const void *data;
} j40_pixels_u8x4;

J40_API j40_err j40_error(const j40_image *image);                                 /* j40.h:233 / 8245 */
J40_API const char *j40_error_string(const j40_image *image);                      /* j40.h:234 / 8251 */

/* `buf` is borrowed until j40_free, which calls freefunc(buf) when freefunc is not NULL */
J40_API j40_err j40_from_memory(j40_image *image, void *buf, size_t size, j40_memory_free_func freefunc); /* j40.h:236 / 8321 */
J40_API j40_err j40_from_file(j40_image *image, const char *path);                 /* j40.h:237 / 8342 */

J40_API j40_err j40_output_format(j40_image *image, int32_t channel, int32_t format); /* j40.h:239 / 8363 */

/* 1 when a frame has been decoded and rendered, 0 on error or when there is no further frame */
J40_API int j40_next_frame(j40_image *image);                                      /* j40.h:241 / 8377 */
J40_API j40_frame j40_current_frame(j40_image *image);                             /* j40.h:242 / 8403 */

/* rows are 32-byte aligned, stride_bytes = 32 * ceil((4 * width + 1) / 32); memory belongs to the image */
J40_API j40_pixels_u8x4 j40_frame_pixels_u8x4(const j40_frame *frame, int32_t channel); /* j40.h:250 / 8425 */
J40_API const j40_u8x4 *j40_row_u8x4(j40_pixels_u8x4 pixels, int32_t y);          /* j40.h:251 / 8464 */

J40_API void j40_free(j40_image *image);                                           /* j40.h:272 / 8471 */

#ifdef __cplusplus
}
#endif
#endif
