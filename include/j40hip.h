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
/* flags & 1: leave the tail of every LfGroup -- dequantisation of the LF samples, adaptive smoothing, LLF coefficients (j40.h:6544-6590,
 * 6492, 5944) -- to the device: j40hip_frame_upload runs it there (device/lf_tail_kernels.hip). Same results; what the pipeline uses. */
J40HIP_API j40hip_frame *j40hip_frame_parse_ex(const void *buf, size_t size, int threads, uint32_t flags, uint32_t *err);
/* Streaming input (what replaces the reference's refillable source and backing buffer, j40.h:1220-1386, 1676-1812, for a buffer
 * that is still being filled -- a file being read, j40_from_file): `buf` has room for the `size` bytes the stream will have, and
 * the parse runs while they arrive. need(ctx, n) is called -- also from the parse's worker threads -- before any byte below n is
 * read and returns once bytes [0, n) are there (or the source has ended: what is missing is then a truncated stream, "shrt" where
 * the reference raises it); have(ctx) says how many are there now. The parse asks for the signature first; a bare codestream's
 * headers and TOC are parsed on growing prefixes, then LfGlobal, HfGlobal and each LfGroup section as its bytes are due, so the
 * host's part of the decode overlaps the arrival of the pass-group sections (most of the file), which only the device reads:
 * wait for the rest before j40hip_frame_upload. A container is parsed once it is complete. Same frame as j40hip_frame_parse_ex. */
typedef void (*j40hip_need_bytes)(void *ctx, size_t upto);
typedef size_t (*j40hip_have_bytes)(void *ctx);
J40HIP_API j40hip_frame *j40hip_frame_parse_streamed(const void *buf, size_t size, int threads, uint32_t flags, j40hip_need_bytes need, j40hip_have_bytes have, void *ctx, uint32_t *err);
/* ... with the LfGroup streams (LF coefficients, HF metadata: j40.h:6722-6790) decoded on HIP device `device` on `stream` (a
 * hipStream_t) instead of the host; `flags` must include 1. The call sleeps while the device works -- meant for many parsing
 * threads per CPU (the pipeline). Frames outside what the device decoder takes are parsed on the host, same results.
 * j40hip_frame_lf_on_device tells which it was. */
J40HIP_API j40hip_frame *j40hip_frame_parse_on(const void *buf, size_t size, int threads, uint32_t flags, int device, void *stream, uint32_t *err);
J40HIP_API int j40hip_frame_lf_on_device(const j40hip_frame *f);
J40HIP_API void j40hip_frame_free(j40hip_frame *f);

/* out[0..20] = width, height, is_modular, num_lf_groups, num_groups, num_passes, nb_block_ctx,
 * block_ctx_size, num_hf_presets, global_scale, quant_lf, x_qm_scale, b_qm_scale, nb_qf_thr,
 * nb_lf_thr[0..2], group_size_shift, bpp, num_extra_channels, xyb_encoded */
J40HIP_API void j40hip_frame_info(const j40hip_frame *f, int64_t *out);
/* bytes of the (re-assembled) codestream and number of pass-group sections */
J40HIP_API size_t j40hip_frame_codestream_size(const j40hip_frame *f);
J40HIP_API int64_t j40hip_frame_num_sections(const j40hip_frame *f);
/* bytes of every pass-group section as the TOC lists them (j40.h:5529), pass-major; out may be NULL; returns how many */
J40HIP_API int64_t j40hip_frame_section_sizes(const j40hip_frame *f, int64_t *out);
/* Modular frames: sections decoded by the wave-cooperative form of the section kernel (diagnostic; -1 when the frame has no Modular plan) */
J40HIP_API int32_t j40hip_frame_coop_sections(j40hip_frame *f, int32_t *total);
J40HIP_API int32_t j40hip_frame_quad_sections(j40hip_frame *f);   /* ... of those, decoded four to a wavefront */
/* sections whose MA tree looks only at a sample's position (properties 0-3, no weighted predictor -- what fast lossless encoders write): decoded
 * in two passes, the stream's tokens first, the prediction behind them (modular_split.hip); J40HIP_NO_SPLIT=1 leaves them to the one-pass kernels */
J40HIP_API int32_t j40hip_frame_split_sections(j40hip_frame *f);

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

/* ---- plan views: what crosses the seam, as plain C arrays (mirrors of j40__frame_st, j40.h:5061-5122,
 *      j40__lf_group_st, j40.h:6360-6390, j40__code_spec, j40.h:2486-2495, j40__modular, j40.h:3553-3564).
 *      Pointers stay valid until j40hip_frame_free. Used by the CPU oracle (oracle/hotpath_oracle.c) so
 *      that it can be driven behind exactly the same boundary as the HIP kernels. ---- */

typedef struct {
	int32_t split_exp, msb_in_token, lsb_in_token;   /* hybrid integer config (j40.h:2283) */
	const int16_t *D;              /* ANS: distribution, 1 << log_alpha_size entries summing to 4096 */
	const uint8_t *lengths;        /* prefix code: code length per symbol */
	int32_t alphabet_size;         /* prefix code: number of symbols */
} j40hip_cluster_view;

typedef struct {
	int32_t num_dist, num_clusters;
	int32_t lz77_enabled, use_prefix_code, min_symbol, min_length, log_alpha_size;
	int32_t lz_len_split_exp, lz_len_msb, lz_len_lsb;
	const uint8_t *cluster_map;    /* [num_dist] */
	const j40hip_cluster_view *clusters;
} j40hip_codespec_view;

typedef struct {
	int32_t left, top, width, height, width8, height8, width64, height64, nb_varblocks;
	const int32_t *blocks;         /* [height8 * width8], (DctSelect + 2) << 20 | varblock or 1 << 20 | varblock */
	const uint8_t *lfindices;      /* [height8 * width8] */
	const float *llfcoeffs[3];     /* [height8 * width8] */
	const int32_t *coeffoff_qfidx; /* [nb_varblocks] */
	const float *hfmul_inv;        /* [nb_varblocks] */
	const int16_t *xfromy, *bfromy;/* [height64 * width64] */
} j40hip_lf_group_view;

typedef struct { uint32_t byte_off, size, bit_off; int32_t ggidx, gx_in_gg, gy_in_gg, gw, gh; } j40hip_section_view;

typedef struct {
	int32_t width, height, num_passes, num_groups, num_lf_groups;
	int32_t nb_block_ctx, nb_qf_thr, nb_lf_thr[3], num_hf_presets, bpp;
	int32_t sections_have_trailer;   /* extra channels: a Modular sub-image follows the coefficients of each section */
	uint32_t single_declared_end;    /* single-section frames: the TOC's end of the section (byte offset); the section itself is readable to the
	                                    end of the codestream, ending short of this is shrt, past it excs (j40.h:7796-7803) */
	int32_t check_section_end;       /* single-section frames only: zero padding + no bytes left at the section's end (the reference checks
	                                    nothing in frames with several sections, j40.h:7778-7795) */
	int32_t global_scale, x_qm_scale, b_qm_scale, x_factor_lf, b_factor_lf;
	float quant_bias[3], quant_bias_num, base_corr_x, base_corr_b, inv_colour_factor;
	float opsin_inv_mat[9], opsin_bias[3], intensity_target;
	const uint8_t *codestream; size_t codestream_size;
	const uint8_t *block_ctx_map; int32_t block_ctx_size;
	const j40hip_codespec_view *coeff_specs;     /* [num_passes] */
	const int32_t *orders[11 * 13 * 3];           /* [pass][order][channel] -> coefficient order or NULL */
	const float *dq_matrix[17]; int32_t dq_size[17];  /* [size][3] dequantisation weights or NULL */
	const j40hip_lf_group_view *lf_groups;       /* [num_lf_groups] */
	const j40hip_section_view *sections;         /* [num_passes * num_groups] */
} j40hip_vardct_view;

typedef struct { int32_t prop, value, a, b; } j40hip_tree_node;   /* see j40_amd/csrc/modular.hpp */
/* kind 0 RCT, 1 Palette, 2 one Squeeze step (channels [begin_c, begin_c + num_c) halved along rows if `horizontal`, else along
 * columns; residual channels right behind them if `in_place`, else at the end of the list) */
typedef struct { int32_t kind, begin_c, rct_type, num_c, nb_colours, nb_deltas, d_pred, horizontal, in_place; } j40hip_transform_view;
typedef struct {
	uint32_t byte_off, size, bit_off; int32_t gx, gy, gw, gh, sidx, first_channel, num_channels;
	int8_t wp[12];                 /* weighted predictor parameters p1, p2, p3[5], w[4] */
	/* the section's MA tree = tree[tree_off .. tree_off + tree_nodes) and its code spec = codespec[spec_idx]:
	 * the global pair, or the section's own when its header says use_global_tree = 0 (j40.h:3740-3746) */
	uint32_t tree_off; int32_t tree_nodes, spec_idx;
	/* RCTs of the section's own header (j40.h:3757), undone over its rectangle before the global transforms
	 * (j40.h:7030): pairs {begin_c relative to first_channel, rct_type} at local_rct + 2 * local_off */
	int32_t local_off, local_count;
	/* a section whose own header lists a palette decodes into a sub-image of its own (the reference's j40__pass_group does that
	 * for every section, j40.h:7024-7032): num_channels planes sub_w/h/meta[sub_off ..], its transforms
	 * sub_transforms[sub_tr_off .. sub_tr_off + sub_tr_count) undone there, then the channels pasted over the rectangle unless
	 * sub_paste is 0 (an earlier pass of a multi-pass frame). sub_off < 0: no sub-image, channels first_channel ... of the frame */
	int32_t sub_off, sub_tr_off, sub_tr_count, sub_paste;
	uint32_t preset_status;        /* != 0: the section's own Modular header did not parse; this is its status, nothing is decoded */
	/* frames whose channels differ in size (after a Squeeze): the section's channels as explicit rectangles,
	 * chan_rects[6 * (chan_off + i)] = {channel, x0, y0, width, height, hshift | vshift << 8}; -1: first_channel ... over gx, gy, gw, gh */
	int32_t chan_off;
	/* the LZ77 distance multiplier of the section's stream + 1 (j40.h:3840-3844), or 0: the widest non-meta channel among the section's
	 * own channels. LfGlobal's section takes the frame-wide image's, which also counts channels the section does not code */
	int32_t dist_mult_p1;
} j40hip_modular_section_view;

typedef struct {
	int32_t width, height, bpp, num_channels, num_sections, num_transforms, num_tree_nodes, alpha_channel;
	int32_t check_section_end;     /* as in j40hip_vardct_view */
	uint32_t single_declared_end;
	const uint8_t *codestream; size_t codestream_size;
	const j40hip_codespec_view *codespec;     /* [num_codespecs] */
	const j40hip_tree_node *tree;              /* [num_tree_nodes]: every tree in use, back to back */
	int32_t num_codespecs;
	const int32_t *channel_w, *channel_h, *channel_meta;   /* coded channels */
	const j40hip_transform_view *transforms;
	const j40hip_modular_section_view *sections;
	const int32_t *local_rct;
	const int32_t *sub_w, *sub_h, *sub_meta;
	const j40hip_transform_view *sub_transforms;
	int8_t global_wp[12];
	const int32_t *chan_rects;
} j40hip_modular_view;

/* The seam for a host that parses the bitstream itself (a patched j40): a frame handle built from the
 * view instead of from a bitstream; then j40hip_frame_upload / j40hip_frame_decode* as usual. Everything is copied. (The view does
 * not carry the global MA tree, so for frames with extra channels the sub-images behind the coefficients are not validated.) */
J40HIP_API j40hip_frame *j40hip_frame_from_vardct_view(const j40hip_vardct_view *v, uint32_t *err);
/* The same seam across processes (the "LF bundle" one rank parses and the others receive): the view flattened
 * into one relocatable blob. j40hip_frame_lf_bundle returns the bytes it needs and writes them when `capacity` suffices (call it
 * with out = NULL first); j40hip_frame_from_lf_bundle builds a frame handle from a received blob (everything is copied and
 * bounds-checked; "rnge" for a malformed blob). */
J40HIP_API size_t j40hip_frame_lf_bundle(j40hip_frame *f, void *out, size_t capacity, uint32_t *err);
J40HIP_API j40hip_frame *j40hip_frame_from_lf_bundle(const void *blob, size_t size, uint32_t *err);

/* What the reference reports after a frame that decoded cleanly: "excs" when bytes follow the frame and the reference gets to see
 * them -- always for single-section frames; for frames with several sections only if the frame ends within the first 64 KB its main
 * buffer holds (j40__seek_buffer, j40.h:1745-1760: otherwise the buffer is emptied, not refilled, and j40__no_more_bytes passes).
 * Bare codestreams only; 0 otherwise. The public API (api.cpp) applies it after a successful decode. */
J40HIP_API uint32_t j40hip_frame_after_frame_status(const j40hip_frame *f);

/* fill the views for a parsed frame; return 0 or a 4-char code ("TODO" for frames the hot path does not cover) */
J40HIP_API uint32_t j40hip_frame_vardct_view(j40hip_frame *f, j40hip_vardct_view *out);
J40HIP_API uint32_t j40hip_frame_modular_view(j40hip_frame *f, j40hip_modular_view *out);

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

/* Single-pass frames keep their quantised coefficients as per-block event lists sized from the section sizes. A section
 * with more non-zero coefficients than its region holds reports "evof" (j40hip_frame_status); j40hip_frame_decode_to_host
 * and the public API then repeat the decode with dense planes on their own, callers of the asynchronous entry points
 * call this and upload / decode again. */
J40HIP_API void j40hip_frame_force_dense(j40hip_frame *f, int dense);

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
/* How j40hip_frame_decode_to_host last decoded the frame: k > 0 -- in two phases, the k longest pass-group sections on a stream of
 * their own beside the others, the image on its way over the link while they finish (device/runtime.hip; J40HIP_TWO_PHASE=0: never);
 * 0: in one; -1: not decoded to the host since its upload. Same pixels, same codes either way. */
J40HIP_API int32_t j40hip_frame_two_phase_sections(const j40hip_frame *f);

/* Stage dumps for parity tests (device -> host copies, synchronous):
 *   quantised HF coefficients of LF group gg, channel c (f32[w8*h8*64], as j40__hf_coeffs leaves them) */
J40HIP_API uint32_t j40hip_frame_read_coeffs(j40hip_frame *f, int64_t gg, int c, float *out);
/*   int16 sample planes (Modular frames) before packing: channel c, w*h */
J40HIP_API uint32_t j40hip_frame_read_plane_i16(j40hip_frame *f, int c, int16_t *out);

/* ---- restoration filters: Gaborish and the edge-preserving filter between the inverse transforms and the colour
 *      conversion, over the whole picture. The reference parses the frame header's RestorationFilter bundle (j40.h:5339-5366) and never
 *      looks at it again -- its j40__gaborish (j40.h:7271) and j40__epf (j40.h:7578) are defined and never called -- so by DEFAULT
 *      nothing runs and the decode matches j40. mode 1 (or J40HIP_RESTORATION=1 in the environment): the filters a VarDCT frame
 *      signals run, stated as those two routines state them (j40_amd/csrc/device/restore_dev.h); mode 2 (J40HIP_RESTORATION=j40):
 *      bit for bit what the routines compute as they stand, including the edge-preserving filter's aliased line buffers (X and Y
 *      read the next channel's rows on half of the lines; the routine only runs at all under an allocator with slack). Their own
 *      complaints -- "gab0", "epf0" (the DEFAULT sharpness table starts with 0: every frame that keeps it), "shrp" -- are reported
 *      behind the sections' codes (j40hip_frame_status) and the picture is then left unfiltered. Single-frame entry points and the
 *      public API; a pipeline decodes such frames on its single-frame path. Modular frames: not filtered (the reference's routines
 *      take float planes only). ---- */
typedef struct {
	int32_t gab_enabled; float gab_weights[3][2];
	int32_t epf_iters; float epf_sharp_lut[8], epf_channel_scale[3], epf_quant_mul, epf_pass0_sigma_scale, epf_pass2_sigma_scale, epf_border_sad_mul, epf_sigma_for_modular;
} j40hip_restoration;
J40HIP_API void j40hip_frame_restoration(const j40hip_frame *f, j40hip_restoration *out);   /* as parsed (mirrors j40__frame_st::gab / ::epf, j40.h:5085-5100) */
J40HIP_API void j40hip_frame_set_restoration(j40hip_frame *f, int mode);                    /* -1: as the environment says (default), 0 off, 1 on, 2 as j40's routines stand */
J40HIP_API int j40hip_frame_sharpness(const j40hip_frame *f, int64_t gg, int16_t *out);     /* LfGroup gg's sharpness map as decoded (i16 w8*h8; j40__lf_group_st::sharpness) */
/* after a decode that ran the filters, stream synchronised: stage 0 the XYB samples as the inverse transforms left them, 1 the filtered
 * ones ([3][height][width] floats); 2 the reciprocal-sigma plane of the edge-preserving filter (w8*h8 floats, < 0: the cell is skipped) */
J40HIP_API uint32_t j40hip_frame_read_xyb(j40hip_frame *f, int stage, float *out);
J40HIP_API float j40hip_frame_restoration_ms(const j40hip_frame *f);   /* device time of the filter kernels in the last decode (J40HIP_RESTORATION_TIMING=1) */
/* known-answer hook: the filter kernels on caller-supplied planes (xyb: [3][h][w] floats, host, in place), a w8*h8 sharpness map and the
 * HfMul reciprocal of the varblock covering each cell; mode 1 or 2; sigma_out optional. 0 or "gab0" / "epf0" / "shrp" / "!gpu". */
J40HIP_API uint32_t j40hip_kat_device_restoration(float *xyb, int32_t w, int32_t h, const int16_t *sharpness, const float *hfmul_inv, const j40hip_restoration *r, int mode, int device, float *sigma_out);

/* per-kernel device time of the last j40hip_frame_decode_timed call, measured with HIP events on
 * the launch stream: ms[0] = entropy decode, ms[1] = coefficients -> pixels, ms[2] = other */
J40HIP_API uint32_t j40hip_frame_decode_timed(j40hip_frame *f, void *rgba_dev, size_t stride_bytes, void *stream, float *ms3);

/* known-answer hook for the renderer's per-sample tail on the device: linear sample -> sRGB transfer -> u8
 * (j40.h:7213-7240, 7925-7935); host arrays in and out */
J40HIP_API uint32_t j40hip_kat_device_srgb_u8(const float *v_host, size_t n, uint8_t *out_host);

/* ---- batches: throughput mode ----
 * A batch is a set of uploaded VarDCT frames decoded together: ONE entropy launch with one pass-group
 * section per wavefront LANE (64 sections per wavefront, every frame's sections side by side), then the
 * coefficients -> pixels kernels. Frames stay owned by the caller and must outlive the batch; the same
 * frame may also be decoded alone (latency mode: one section per wavefront, scalarised) with
 * j40hip_frame_decode. rgba_dev[i] / stride_bytes[i] belong to frames[i]. Per-frame result:
 * j40hip_frame_status(frames[i]) after the stream has been synchronised. */
typedef struct j40hip_batch j40hip_batch;
J40HIP_API j40hip_batch *j40hip_batch_create(j40hip_frame *const *frames, int64_t n, uint32_t *err);
J40HIP_API void j40hip_batch_free(j40hip_batch *b);
J40HIP_API uint32_t j40hip_batch_decode(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream);
/* ms3 as j40hip_frame_decode_timed, for the whole batch */
J40HIP_API uint32_t j40hip_batch_decode_timed(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream, float *ms3);
/* asynchronous timed decode: records the stage events in `slot` (0..4095) and returns without waiting, so that
 * batches on different streams overlap (one batch's entropy kernel with another's pixel kernels);
 * j40hip_batch_elapsed reads a slot's ms3 once its stream has been synchronised */
J40HIP_API uint32_t j40hip_batch_decode_recorded(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream, int32_t slot);
J40HIP_API uint32_t j40hip_batch_elapsed(j40hip_batch *b, int32_t slot, float *ms3);
/* `stream` waits for stage 1 (cleared), 2 (entropy decoded) or 3 (pixels written) of the decode recorded in `slot` */
J40HIP_API uint32_t j40hip_batch_wait_stage(j40hip_batch *b, int32_t slot, int32_t stage, void *stream);

/* a batch object re-used for other members (keeps its device arrays, streams and events); the previous members' decodes must be complete */
J40HIP_API uint32_t j40hip_batch_reset(j40hip_batch *b, j40hip_frame *const *frames, int64_t n);

/* ---- building blocks for pipelines ---- */
/* j40hip_frame_upload with the copies enqueued on `stream` (the plan is staged in pinned memory owned by the calling thread, so the
 * copy is a real asynchronous DMA beside other streams' kernels); returns when they have completed */
J40HIP_API uint32_t j40hip_frame_upload_on(j40hip_frame *f, int device, void *stream);
/* frees the calling thread's pinned staging buffer (call before a thread that uploaded frames exits) */
J40HIP_API void j40hip_thread_release(void);
/* Takes the library's process-wide state down: stops and joins its service threads, frees the cached device memory, pooled events
 * and cached tables. Optional (a process may simply end); nothing of the library may be running, and objects created before must
 * have been freed. The library can be used again afterwards. */
J40HIP_API void j40hip_shutdown(void);
/* j40hip_frame_status in two halves: `begin` enqueues the copy of the status words on `stream`, `end` -- once the caller has waited
 * for that stream -- reduces them to the frame's verdict without touching the device. VarDCT frames without extra channels. */
J40HIP_API uint32_t j40hip_frame_status_begin(j40hip_frame *f, void *stream);
J40HIP_API uint32_t j40hip_frame_status_end(j40hip_frame *f);
/* the caller guarantees that nothing is pending on the frame's device memory: j40hip_frame_free then skips its device-wide wait */
J40HIP_API void j40hip_frame_mark_idle(j40hip_frame *f);

/* ---- whole-frame throughput pipeline (j40_amd/csrc/device/pipeline.hip): codestreams in host memory -> RGBA u8x4, every stage of
 *      many frames in flight. `host_threads` workers parse what precedes the LfGroup sections of a frame and copy it to the device
 *      without waiting for it; one thread enqueues, per batch of `batch_frames` frames, the LfGroup streams, the plan build, the
 *      entropy decode and the pixel kernels on the batch's stream, up to `max_in_flight` batches on their own streams; host output
 *      is copied back on the batch's stream behind its kernels. Frames outside that path (Modular, one section, extra channels)
 *      are decoded by a worker through the single-frame entry points. The serving shape of j40_from_memory + j40_next_frame +
 *      j40_frame_pixels_u8x4 for many images. ---- */
typedef struct j40hip_pipeline j40hip_pipeline;
/* host_threads < 1: the container's CPU quota (cgroup cpu.max) less two, at most 16 -- never the visible CPU count. Keep the busy threads
 * of the process below its quota: a cgroup that runs into it throttles every thread of the process, the HIP runtime's included, and
 * the copies back to the host then crawl. With the LfGroup streams on the GPU a frame's host stage is 1-2 ms: two to four threads. */
J40HIP_API j40hip_pipeline *j40hip_pipeline_create(int device, int host_threads, int batch_frames, int max_in_flight, uint32_t *err);
/* flags bits 0-1: who decodes the LfGroup streams (j40.h:6722-6790) of the batched frames: 0 decided frame by frame (the host
 * threads keep them while the device has batches queued up, else the device takes them), 1 always the device (k_lf_groups),
 * 2 always the host threads. Bit 2: tune the process's malloc for many threads freeing multi-megabyte blocks (mallopt: mmap
 * threshold, trim threshold, top pad) -- process-wide, hence opt-in. Bit 3: do not size the device memory cache for the pipeline's full
 * depth at its second full batch (a serving pipeline whose batches fill up only now and then). */
J40HIP_API j40hip_pipeline *j40hip_pipeline_create_ex(int device, int host_threads, int batch_frames, int max_in_flight, uint32_t flags, uint32_t *err);
J40HIP_API int64_t j40hip_pipeline_lf_device_frames(j40hip_pipeline *p);   /* frames whose LfGroup streams the device decoded (since the last reset) */
J40HIP_API void j40hip_pipeline_free(j40hip_pipeline *p);
/* frames still queued when the pipeline is freed are dropped; frames already prepared or in flight are decoded first */
/* queues one image. buf is borrowed until the ticket is done. rgba: `stride_bytes` * height bytes of host memory (pinned memory for
 * full copy speed) or, with device_output != 0, of device memory (no copy back). */
J40HIP_API uint32_t j40hip_pipeline_submit(j40hip_pipeline *p, const void *buf, size_t size, void *rgba, size_t stride_bytes, int device_output, int64_t *ticket);
/* One image, synchronously: queued like a submitted one, the calling thread sleeps until it is done; returns 0 or the image's 4-char
 * code. The pixel memory is asked for through `alloc` once the image's size is known (called once, on a pipeline thread, before
 * anything is written; it returns host memory -- pinned for full copy speed -- of *stride_bytes * height bytes, *stride_bytes >=
 * 4 * width, or NULL: "!mem"). Any number of threads may call this at once: their images share the pipeline's batches. This is what
 * j40_next_frame (include/j40.h) runs on when several threads are inside the public API at once. */
typedef void *(*j40hip_output_alloc)(void *ctx, int64_t width, int64_t height, size_t *stride_bytes);
J40HIP_API uint32_t j40hip_pipeline_run(j40hip_pipeline *p, const void *buf, size_t size, j40hip_output_alloc alloc, void *ctx);
/* > 0: a prepared image waits at most `ms` for its batch to fill up while the device has a free batch slot (serving: images arrive
 * one by one); 0 (default): a partial batch is launched only when nothing else is queued */
J40HIP_API void j40hip_pipeline_set_max_wait_ms(j40hip_pipeline *p, double ms);
/* waits until everything submitted so far is done */
J40HIP_API uint32_t j40hip_pipeline_drain(j40hip_pipeline *p);
/* 0 or the image's 4-char error code, as j40_error would give it ("rnge" for an unknown or unfinished ticket) */
J40HIP_API uint32_t j40hip_pipeline_result(j40hip_pipeline *p, int64_t ticket);
/* out8: [0] ms the worker threads spent in the host stage of batched frames and [1] in whole single-frame decodes (summed over the
 * threads), [2] images completed, [3] ms from first submit to last completion, [4] entropy-stage and [5] pixel-stage ms summed over
 * the batch launches (HIP events on their streams), [6] batch launches, [7] images in them */
J40HIP_API void j40hip_pipeline_stats(j40hip_pipeline *p, double *out8);
/* out12: as above, then [8] ms of the batches' first stage (LfGroup streams + plan build + LfGroup tail), [9] images whose LfGroup
 * streams the device decoded, [10] images decoded on the single-frame path, [11] ms of k_hf_lanes itself summed over the launches,
 * as the device recorded them (first wavefront's start to last one's end: the duration rocprofv3 reports; [4] also counts what the
 * launch waited for behind other kernels) */
J40HIP_API void j40hip_pipeline_stats_ex(j40hip_pipeline *p, double *out12);
/* out5: the launches of the LfGroup lane decoder (k_lf_rows / k_lf_lanes; j40__lf_group's two Modular streams, j40.h:6722-6790) since
 * the last reset: [0] their durations summed, ms, as the device recorded them (first wavefront's start to last one's end), [1]
 * launches, [2] frames, [3] LfGroup sections and [4] wavefronts in them */
J40HIP_API void j40hip_pipeline_lf_stats(j40hip_pipeline *p, double *out5);
J40HIP_API void j40hip_pipeline_reset_stats(j40hip_pipeline *p);

/* The SDMA engine the pipeline's copies back to host memory go to on `device` (hsa_amd_memory_async_copy_on_engine; -1: none, the
 * copies are hipMemcpyAsync). An MI355X has sixteen engines of very different device-to-host rates (57 ... 7 GB/s) and hipMemcpyAsync
 * takes whichever is free; the library measures them once per process and device (32 MB each, at its first pipeline or at this call)
 * and keeps the fastest. gbps16 (optional): the measured GB/s per engine, 0 = not measured, < 0 = failed; masks2 (optional, THREE
 * words): the runtime's masks {free, recommended} for the direction and the engines set aside for the worker threads' uploads (an
 * SDMA engine works on one copy at a time: an upload sharing the engine of the copies back waits behind 133 MB transfers). J40HIP_COPY_ENGINE=hip: never; =<n>: engine n, unmeasured. */
J40HIP_API int j40hip_copy_engine(int device, double *gbps16, uint32_t *masks2);

/* ---- stage dump of the pipeline's DEVICE stages, for parity tests (tests/test_device_stages.py): one image goes through the same
 *      enqueue a batch of one takes -- front parse on the host; the LfGroup streams on the device (k_lf_lanes) when `lf_on_device`
 *      and the frame's tables allow it, else by the host decoder; plan build (k_plan_place / _scan / _emit), LfGroup tail, entropy
 *      decode, pixel kernels, verdict -- and what every stage produced is copied back and handed out in the layout of the reference's
 *      j40__lf_group_st (j40.h:6360-6390), like the j40hip_frame_lf_group_* accessors do for the host parse. NULL + "TODO" for
 *      images the batched path does not take. ---- */
typedef struct j40hip_stage_dump j40hip_stage_dump;
J40HIP_API j40hip_stage_dump *j40hip_stage_dump_create(const void *buf, size_t size, int device, int lf_on_device, uint32_t *err);
J40HIP_API void j40hip_stage_dump_free(j40hip_stage_dump *d);
/* out8: [0] the frame's verdict (k_plan_verdict: 0 or the first failing section's 4-char code, LfGroup and pass-group sections alike),
 * [1] flags: bit 0 an LfGroup section the device decoder leaves to the host, bit 1 an event region overflowed, bit 2 the LfGroup
 * streams were decoded on the device, [2] LfGroups, [3] groups, [4] width, [5] height, [6] union of the DctSelect values placed, [7] varblocks */
J40HIP_API void j40hip_stage_dump_info(const j40hip_stage_dump *d, uint32_t *out8);
/* out10: left, top, width, height, width8, height8, width64, height64, varblocks placed, the section's status (4-char code or 0) */
J40HIP_API int j40hip_stage_dump_lf_group_info(const j40hip_stage_dump *d, int64_t gg, int32_t *out10);
/* which: 0 blocks (i32 w8*h8, the reference's encoding, rebuilt from the device's varblock records), 1 lfindices (u8 w8*h8, from the
 * decoded LF integers with the frame's thresholds, j40.h:6566-6570), 2 xfromy / 3 bfromy (i16 w64*h64), 4 sharpness (i16 w8*h8; only
 * when the device decoded the streams), 5 / 6 / 7 the raw LF integers of X / Y / B (i16 w8*h8) */
J40HIP_API int j40hip_stage_dump_plane(const j40hip_stage_dump *d, int64_t gg, int which, void *out);
/* varblocks in placement order (= the reference's varblock index): coeffoff_qfidx, hfmul.inv, and (optional) x8, y8, DctSelect; returns how many */
J40HIP_API int j40hip_stage_dump_varblocks(const j40hip_stage_dump *d, int64_t gg, int32_t *coeffoff_qfidx, float *hfmul_inv, int32_t *x8_y8_dctsel);
J40HIP_API int j40hip_stage_dump_llf(const j40hip_stage_dump *d, int64_t gg, int c, float *out);
/* the entropy kernel's block list of one group (frame-wide group index), in j40__hf_coeffs' visiting order: per block
 * coeffoff_qfidx, pos_dct (y8 * 32 + x8 inside the group | DctSelect << 10), bctx3 (block context of Y | X << 4 | B << 8); returns the count */
J40HIP_API int64_t j40hip_stage_dump_group_blocks(const j40hip_stage_dump *d, int64_t group, uint32_t *out3, int64_t capacity);
/* the pixel kernels' work list (sorted by DctSelect; class_start28[27] = varblocks): out8 = px, py, effw, effh, dctsel, blk, llf_base,
 * coeff_base; out3 = mult1, kx_hf, kb_hf; returns the count */
J40HIP_API int64_t j40hip_stage_dump_sorted_varblocks(const j40hip_stage_dump *d, int32_t *out8, float *out3, int32_t *class_start28, int64_t capacity);
J40HIP_API int j40hip_stage_dump_rgba(const j40hip_stage_dump *d, uint8_t *out);   /* width * 4 bytes per row */

#ifdef __cplusplus
}
#endif
#endif
