/*
 * include/j40hip.h -- thin C-ABI between the C host (container / header / TOC / LF parsing) and the
 * HIP hot path, plus stage-level entry points used by the parity tests.
 *
 * The reference has no plugin / FFI layer: its public API (include/j40.h) is the
 * drop-in boundary. This header is the *internal* seam the north star asks for -- "the C host parses
 * the container, frame header and group TOC, then hands the per-group hot path to HIP through a
 * thin C-ABI layer" -- cut where the reference calls j40__pass_group (j40.h:7007, driver loop
 * j40.h:8202-8204), j40__inverse_transform(&f->gmodular) (j40.h:8209), j40__combine_vardct
 * (j40.h:7862 / 8210) and j40__render_to_u8x4_rgba (j40.h:7910 / 8393). Plain pointers and sizes
 * only; every device pointer is a raw HIP device address, every stream a hipStream_t passed as void*.
 */
#ifndef J40HIP_H_
#define J40HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define J40HIP_API __attribute__((visibility("default")))

typedef struct j40hip_frame j40hip_frame;

/* ---- host side: replaces the reference's parse up to the section loop (j40.h:8175-8192 and the
 *      LF-group half of j40__lf_or_pass_group_in_section, j40.h:7840-7846) ---- */

/* Parses container, headers, TOC, LfGlobal, HfGlobal and all LfGroup sections of the single frame
 * in `buf` (borrowed until j40hip_frame_free). `threads` = host threads for LfGroup sections.
 * Returns NULL and sets *err (4-char code, same values as the reference's j40_err) on failure. */
J40HIP_API j40hip_frame *j40hip_frame_parse(const void *buf, size_t size, int threads, uint32_t *err);
J40HIP_API void j40hip_frame_free(j40hip_frame *f);

/* out[0..20] = width, height, is_modular, num_lf_groups, num_groups, num_passes, nb_block_ctx,
 * block_ctx_size, num_hf_presets, global_scale, quant_lf, x_qm_scale, b_qm_scale, nb_qf_thr,
 * nb_lf_thr[0..2], group_size_shift, bpp, num_extra_channels, xyb_encoded */
J40HIP_API void j40hip_frame_info(const j40hip_frame *f, int64_t *out);
/* bytes of the (re-assembled) codestream and number of pass-group sections */
J40HIP_API size_t j40hip_frame_codestream_size(const j40hip_frame *f);
J40HIP_API int64_t j40hip_frame_num_sections(const j40hip_frame *f);

/* stage accessors for parity tests (mirror j40__lf_group_st, j40.h:6360-6390) */
J40HIP_API void j40hip_frame_lf_group_info(const j40hip_frame *f, int64_t gg, int32_t *out9);
/* which: 0 blocks (i32 w8*h8), 1 lfindices (u8 w8*h8), 2 xfromy (i16 w64*h64), 3 bfromy */
J40HIP_API int j40hip_frame_lf_group_plane(const j40hip_frame *f, int64_t gg, int which, void *out);
J40HIP_API void j40hip_frame_varblocks(const j40hip_frame *f, int64_t gg, int32_t *coeffoff_qfidx, float *hfmul_inv);
J40HIP_API void j40hip_frame_llf(const j40hip_frame *f, int64_t gg, int c, float *out);
J40HIP_API int32_t j40hip_frame_dq_matrix(const j40hip_frame *f, int idx, float *out_n_by_3);
J40HIP_API int32_t j40hip_frame_order(const j40hip_frame *f, int pass, int idx, int c, int32_t *out);
J40HIP_API int32_t j40hip_frame_block_ctx_map(const j40hip_frame *f, uint8_t *out);
/* Modular planes decoded on the host inside LfGlobal (j40.h:6334-6336): returns 0 and fills w*h int16 */
J40HIP_API int j40hip_frame_global_plane(const j40hip_frame *f, int c, int16_t *out, int32_t *w, int32_t *h);

/* host table builders exposed for known-answer tests against the reference's internals */
J40HIP_API int32_t j40hip_kat_natural_order(int32_t log_rows, int32_t log_columns, int32_t *out);          /* j40.h:4980 */
J40HIP_API int32_t j40hip_kat_library_dq_matrix(int idx, float *out_n_by_3);                              /* j40.h:4828 */
J40HIP_API void j40hip_kat_forward_llf(float *buf, int32_t log_rows, int32_t log_columns);                /* j40.h:5944 */
J40HIP_API float j40hip_kat_half_secant(int i);                                                           /* j40.h:5690 */
J40HIP_API float j40hip_kat_lf2llf_scale(int i);                                                          /* j40.h:5739 */

/* ---- device side ---- */

/* Number of visible HIP devices (0 when there is none); never throws. */
J40HIP_API int j40hip_device_count(void);

/* Uploads the frame plan (codestream, code specs, orders, dequant tables, LF bundle) to `device`
 * and allocates the working buffers. Returns 0 or a 4-char error ("!gpu": no device / HIP error). */
J40HIP_API uint32_t j40hip_frame_upload(j40hip_frame *f, int device);

/* Section subset decoded by this process (multi-GPU sharding by pass-group section):
 * groups [first_group, first_group + num_groups) of every pass. Default: all. */
J40HIP_API uint32_t j40hip_frame_set_group_range(j40hip_frame *f, int64_t first_group, int64_t num_groups);

/* Runs the hot path on `stream`: entropy decode of every pass-group section, dequantisation,
 * chroma-from-luma, inverse transforms, XYB -> sRGB and RGBA u8x4 packing (or the Modular
 * equivalents) into device memory `rgba_dev` with `stride_bytes` per row. Asynchronous. */
J40HIP_API uint32_t j40hip_frame_decode(j40hip_frame *f, void *rgba_dev, size_t stride_bytes, void *stream);

/* After the stream has been synchronised: first failing section in TOC order -> its 4-char code
 * ("coef", "shrt", "excs", "ans?" ...), 0 if every section decoded cleanly (j40.h:530-534). */
J40HIP_API uint32_t j40hip_frame_status(j40hip_frame *f);

/* Convenience for the public API: decode + copy to host rows of `stride_bytes`. Synchronous. */
J40HIP_API uint32_t j40hip_frame_decode_to_host(j40hip_frame *f, void *rgba_host, size_t stride_bytes);

/* Stage dumps for parity tests (device -> host copies, synchronous):
 *   quantised HF coefficients of LF group gg, channel c (f32[w8*h8*64], as j40__hf_coeffs leaves them) */
J40HIP_API uint32_t j40hip_frame_read_coeffs(j40hip_frame *f, int64_t gg, int c, float *out);
/*   int16 sample planes (Modular frames) before packing: channel c, w*h */
J40HIP_API uint32_t j40hip_frame_read_plane_i16(j40hip_frame *f, int c, int16_t *out);

/* per-kernel device time of the last j40hip_frame_decode_timed call, measured with HIP events on
 * the launch stream: ms[0] = entropy decode, ms[1] = coefficients -> pixels, ms[2] = other */
J40HIP_API uint32_t j40hip_frame_decode_timed(j40hip_frame *f, void *rgba_dev, size_t stride_bytes, void *stream, float *ms3);

#ifdef __cplusplus
}
#endif
#endif
