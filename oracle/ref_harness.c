/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * A thin harness translation unit around the *unmodified* reference decoder, compiled straight
 * from where it lies (/root/reference/j40.h) into oracle/_ref/libj40ref.so by oracle/Makefile.
 * Nothing of the reference is copied into this repository: this file only #includes it.
 *
 * It exposes
 *   - the whole-path decode through the reference's public API (ref_decode_rgba), used as the
 *     golden oracle and as bench.py's `cpu_baseline` (kind = "reference");
 *   - a staged decode (ref_stage_*) that stops after all sections are parsed but before
 *     j40__combine_vardct (j40.h:8209-8210), so tests can diff intermediate products
 *     (block maps, LLF coefficients, quantised HF coefficients, dequant tables, orders);
 *   - known-answer entry points for single internal functions (IDCT family, natural order, ...).
 *
 * Build flags: -O3 without -march=native / -mfma (the VarDCT float path is contraction sensitive,
 * see j40.h:5834 and SURVEY.md section 0 fact 7).
 */
#include <stdlib.h>
#include <string.h>
#ifdef REF_ZEROED_ALLOC
/* _ref/libj40ref_zalloc.so: the same sources with the reference's allocator hooks (J40_MALLOC ..., j40.h:413-417) pointed at an
 * allocator that zero-fills and leaves slack on both sides of every block. Needed to run j40__epf_step at all (the KATs of the
 * restoration filters below; nothing else uses this build): the routine -- never called by the reference -- sets up the mirrored
 * borders of its line buffer with `c * cstride` added to row pointers that already include it (j40.h:7484-7488), so channel 1's
 * borders land in channel 2's slots, channel 2's past the end of the buffer, and the first two picture rows then read border
 * slots of channels 1 and 2 that were never written; and its rotating row 0 writes one float in front of the buffer
 * (`lines[3][c][-2]` once `linebuf + 1` has rotated into slot 3, j40.h:7512). Under malloc that is heap corruption ("double free or
 * corruption" at the routine's own j40__free); under this allocator it is deterministic: the unwritten slots read 0.0f. */
#define REF_SLACK 256
static void *ref_zmalloc(size_t n) { char *p = (char *) calloc(1, n * 2 + 4096 + 2 * REF_SLACK); return p ? p + REF_SLACK : NULL; }
static void ref_zfree(void *q) { if (q) free((char *) q - REF_SLACK); }
static void *ref_zrealloc(void *q, size_t n) { char *p = (char *) realloc(q ? (char *) q - REF_SLACK : NULL, n * 2 + 4096 + 2 * REF_SLACK); return p ? p + REF_SLACK : NULL; }
#define J40_MALLOC ref_zmalloc
#define J40_CALLOC(n, s) ref_zmalloc((size_t) (n) * (size_t) (s))
#define J40_REALLOC ref_zrealloc
#define J40_FREE ref_zfree
#endif
#define J40_CONFIRM_THAT_THIS_IS_EXPERIMENTAL_AND_POTENTIALLY_UNSAFE
#define J40_IMPLEMENTATION
#include "j40.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* whole path through the public API                                                          */

/* Decodes `buf` and writes tightly packed RGBA (w*h*4 bytes) into a malloc'ed buffer.
 * Returns 0 on success or the j40_err code. */
REF_API uint32_t ref_decode_rgba(const void *buf, size_t size, uint8_t **out, int32_t *w, int32_t *h) {
	j40_image image;
	uint32_t err;
	*out = NULL; *w = *h = 0;
	j40_from_memory(&image, (void *) buf, size, NULL);
	j40_output_format(&image, J40_RGBA, J40_U8X4);
	if (j40_next_frame(&image)) {
		j40_frame frame = j40_current_frame(&image);
		j40_pixels_u8x4 pixels = j40_frame_pixels_u8x4(&frame, J40_RGBA);
		int32_t y;
		*w = pixels.width; *h = pixels.height;
		*out = (uint8_t *) malloc((size_t) pixels.width * (size_t) pixels.height * 4);
		for (y = 0; y < pixels.height; ++y) {
			memcpy(*out + (size_t) y * (size_t) pixels.width * 4, j40_row_u8x4(pixels, y), (size_t) pixels.width * 4);
		}
	}
	err = j40_error(&image);
	j40_free(&image);
	if (err) { free(*out); *out = NULL; }
	return err;
}

REF_API void ref_free(void *p) { free(p); }

/* Same as above but decodes into a caller buffer and returns nothing but the error; used for
 * timing (the copy out of the reference's padded plane is part of "RGBA rows resident"). */
REF_API uint32_t ref_decode_into(const void *buf, size_t size, uint8_t *out, size_t outcap) {
	j40_image image;
	uint32_t err;
	j40_from_memory(&image, (void *) buf, size, NULL);
	j40_output_format(&image, J40_RGBA, J40_U8X4);
	if (j40_next_frame(&image)) {
		j40_frame frame = j40_current_frame(&image);
		j40_pixels_u8x4 pixels = j40_frame_pixels_u8x4(&frame, J40_RGBA);
		int32_t y;
		if ((size_t) pixels.width * (size_t) pixels.height * 4 <= outcap) {
			for (y = 0; y < pixels.height; ++y) {
				memcpy(out + (size_t) y * (size_t) pixels.width * 4, j40_row_u8x4(pixels, y), (size_t) pixels.width * 4);
			}
		}
	}
	err = j40_error(&image);
	j40_free(&image);
	return err;
}

REF_API const char *ref_error_string_for(const void *buf, size_t size) {
	static char out[256];
	j40_image image;
	j40_from_memory(&image, (void *) buf, size, NULL);
	j40_output_format(&image, J40_RGBA, J40_U8X4);
	if (j40_next_frame(&image)) j40_current_frame(&image);
	snprintf(out, sizeof out, "%s", j40_error(&image) ? j40_error_string(&image) : "");
	j40_free(&image);
	return out;
}

/* ------------------------------------------------------------------------------------------ */
/* staged decode: everything up to (not including) the inverse transforms / combine            */

typedef struct {
	j40_image image;
	j40__inner *inner;
	int combined;
} ref_stage;

/* mirrors the statement sequence of j40__advance (j40.h:8171-8207) without the coroutine */
static j40_err ref_stage_run(j40__inner *inner) {
	j40__st stbuf, *st = &stbuf;
	j40__frame_st *f;
	j40__init_state(st, inner);
	f = st->frame;
	J40__TRY(j40__init_buffer(st, 0, INT64_MAX));
	J40__TRY(j40__signature(st));
	J40__TRY(j40__image_metadata(st));
	if (st->image->want_icc) J40__TRY(j40__icc(st));
	J40__TRY(j40__frame_header(st));
	J40__SHOULD(f->is_last, "TODO");
	J40__SHOULD(f->type == J40__FRAME_REGULAR, "TODO");
	J40__TRY(j40__read_toc(st, &inner->toc));
	J40__TRY(j40__lf_global_in_section(st, &inner->toc));
	J40__TRY(j40__hf_global_in_section(st, &inner->toc));
	J40__TRY(j40__allocate_lf_groups(st, &inner->lf_groups));
	if (inner->toc.single_size) {
		J40__TRY(j40__lf_group(st, &inner->lf_groups[0]));
		J40__TRY(j40__prepare_dq_matrices(st));
		J40__TRY(j40__prepare_orders(st));
		J40__TRY(j40__pass_group(st, 0, 0, 0, f->width, f->height, 0, &inner->lf_groups[0]));
		J40__TRY(j40__zero_pad_to_byte(st));
	} else {
		while (inner->toc.nsections_read < inner->toc.nsections) {
			J40__TRY(j40__lf_or_pass_group_in_section(st, &inner->toc, inner->lf_groups));
		}
	}
	J40__TRY(j40__end_of_frame(st, &inner->toc));
J40__ON_ERROR:
	j40__save_state(st, inner, J40__ORIGIN_next_frame);
	return st->err;
}

REF_API ref_stage *ref_stage_open(const void *buf, size_t size, uint32_t *err) {
	ref_stage *s = (ref_stage *) calloc(1, sizeof(ref_stage));
	*err = j40_from_memory(&s->image, (void *) buf, size, NULL);
	if (*err) { free(s); return NULL; }
	s->inner = s->image.u.inner;
	*err = ref_stage_run(s->inner);
	if (*err) { j40_free(&s->image); free(s); return NULL; }
	return s;
}

REF_API void ref_stage_close(ref_stage *s) {
	if (!s) return;
	j40_free(&s->image);
	free(s);
}

/* frame-level scalars: out[0..] = width,height,is_modular,num_lf_groups,num_groups,num_passes,
 * nb_block_ctx,block_ctx_size,num_hf_presets,global_scale,quant_lf,x_qm_scale,b_qm_scale,
 * nb_qf_thr,nb_lf_thr[0..2], group_size_shift, bpp, num_extra_channels, xyb_encoded */
REF_API void ref_stage_frame_info(ref_stage *s, int64_t *out) {
	j40__frame_st *f = &s->inner->frame;
	j40__image_st *im = &s->inner->image;
	int i = 0;
	out[i++] = f->width; out[i++] = f->height; out[i++] = f->is_modular;
	out[i++] = f->num_lf_groups; out[i++] = f->num_groups; out[i++] = f->num_passes;
	out[i++] = f->nb_block_ctx; out[i++] = f->block_ctx_size; out[i++] = f->num_hf_presets;
	out[i++] = f->global_scale; out[i++] = f->quant_lf; out[i++] = f->x_qm_scale; out[i++] = f->b_qm_scale;
	out[i++] = f->nb_qf_thr; out[i++] = f->nb_lf_thr[0]; out[i++] = f->nb_lf_thr[1]; out[i++] = f->nb_lf_thr[2];
	out[i++] = f->group_size_shift; out[i++] = im->bpp; out[i++] = im->num_extra_channels; out[i++] = im->xyb_encoded;
}

/* LF group geometry: out = left, top, width, height, width8, height8, width64, height64, nb_varblocks */
REF_API void ref_stage_lf_group_info(ref_stage *s, int64_t ggidx, int32_t *out) {
	j40__lf_group_st *gg = &s->inner->lf_groups[ggidx];
	out[0] = gg->left; out[1] = gg->top; out[2] = gg->width; out[3] = gg->height;
	out[4] = gg->width8; out[5] = gg->height8; out[6] = gg->width64; out[7] = gg->height64;
	out[8] = gg->nb_varblocks;
}

static void ref_copy_plane(const j40__plane *p, void *out, size_t elemsize) {
	int32_t y;
	for (y = 0; y < p->height; ++y) {
		memcpy((char *) out + (size_t) y * (size_t) p->width * elemsize,
			(const char *) p->pixels + (size_t) p->stride_bytes * (size_t) y, (size_t) p->width * elemsize);
	}
}

/* which: 0 blocks (i32, w8*h8), 1 lfindices (u8, w8*h8), 2 xfromy (i16, w64*h64), 3 bfromy (i16),
 * 4 sharpness (i16, w8*h8) */
REF_API int ref_stage_lf_group_plane(ref_stage *s, int64_t ggidx, int which, void *out) {
	j40__lf_group_st *gg = &s->inner->lf_groups[ggidx];
	switch (which) {
	case 0: ref_copy_plane(&gg->blocks, out, 4); return 0;
	case 1: ref_copy_plane(&gg->lfindices, out, 1); return 0;
	case 2: if (gg->xfromy.type != J40__PLANE_I16) return -1; ref_copy_plane(&gg->xfromy, out, 2); return 0;
	case 3: if (gg->bfromy.type != J40__PLANE_I16) return -1; ref_copy_plane(&gg->bfromy, out, 2); return 0;
	case 4: if (gg->sharpness.type != J40__PLANE_I16) return -1; ref_copy_plane(&gg->sharpness, out, 2); return 0;
	}
	return -1;
}

/* varblocks: coeffoff_qfidx (i32[nb]) and hfmul.inv (f32[nb]) */
REF_API void ref_stage_varblocks(ref_stage *s, int64_t ggidx, int32_t *coeffoff_qfidx, float *hfmul_inv) {
	j40__lf_group_st *gg = &s->inner->lf_groups[ggidx];
	int32_t i;
	for (i = 0; i < gg->nb_varblocks; ++i) {
		coeffoff_qfidx[i] = gg->varblocks[i].coeffoff_qfidx;
		hfmul_inv[i] = gg->varblocks[i].hfmul.inv;
	}
}

/* c in 0..2 (X,Y,B): llf -> f32[w8*h8], coeffs -> f32[w8*h8*64] (quantised integers as floats
 * before ref_stage_combine, dequantised after) */
REF_API void ref_stage_llf(ref_stage *s, int64_t ggidx, int c, float *out) {
	j40__lf_group_st *gg = &s->inner->lf_groups[ggidx];
	memcpy(out, gg->llfcoeffs[c], sizeof(float) * (size_t) (gg->width8 * gg->height8));
}
REF_API void ref_stage_coeffs(ref_stage *s, int64_t ggidx, int c, float *out) {
	j40__lf_group_st *gg = &s->inner->lf_groups[ggidx];
	memcpy(out, gg->coeffs[c], sizeof(float) * (size_t) (gg->width8 * gg->height8 * 64));
}

/* dequantisation table idx (0..16) after j40__load_dq_matrix; out = f32[rows*cols][3]; returns
 * rows*cols or 0 if that table was never loaded */
REF_API int32_t ref_stage_dq_matrix(ref_stage *s, int idx, float *out) {
	j40__dq_matrix *m = &s->inner->frame.dq_matrix[idx];
	int32_t n, i;
	if (m->mode != J40__DQ_ENC_RAW || !m->params) return 0;
	n = (int32_t) m->n * (int32_t) m->m;
	for (i = 0; i < n; ++i) { out[i * 3] = m->params[i][0]; out[i * 3 + 1] = m->params[i][1]; out[i * 3 + 2] = m->params[i][2]; }
	return n;
}

/* coefficient order for (pass, order idx, channel); returns size or 0 if not loaded */
REF_API int32_t ref_stage_order(ref_stage *s, int pass, int idx, int c, int32_t *out) {
	j40__frame_st *f = &s->inner->frame;
	int32_t size = 1 << (J40__LOG_ORDER_SIZE[idx][0] + J40__LOG_ORDER_SIZE[idx][1]);
	if (!((f->order_loaded >> idx) & 1) || !f->orders[pass][idx][c]) return 0;
	memcpy(out, f->orders[pass][idx][c], sizeof(int32_t) * (size_t) size);
	return size;
}

REF_API int32_t ref_stage_block_ctx_map(ref_stage *s, uint8_t *out) {
	j40__frame_st *f = &s->inner->frame;
	if (!f->block_ctx_map) return 0;
	memcpy(out, f->block_ctx_map, (size_t) f->block_ctx_size);
	return f->block_ctx_size;
}

/* runs the rest: global inverse transforms, dequant + combine, render (j40.h:8209-8210, 8393) */
REF_API uint32_t ref_stage_combine(ref_stage *s) {
	j40__st stbuf, *st = &stbuf;
	j40__inner *inner = s->inner;
	j40__frame_st *f;
	j40__init_state(st, inner);
	f = st->frame;
	if (s->combined) return 0;
	J40__TRY(j40__inverse_transform(st, &f->gmodular));
	if (!f->is_modular) J40__TRY(j40__combine_vardct(st, inner->lf_groups));
	J40__TRY(j40__render_to_u8x4_rgba(st, &inner->rendered_rgba));
	inner->rendered = 1;
	s->combined = 1;
J40__ON_ERROR:
	return st->err;
}

/* after ref_stage_combine: int16 sample planes of the frame (channel c), tightly packed */
REF_API int ref_stage_plane_i16(ref_stage *s, int c, int16_t *out) {
	j40__frame_st *f = &s->inner->frame;
	if (c < 0 || c >= f->gmodular.num_channels) return -1;
	if (f->gmodular.channel[c].type != J40__PLANE_I16) return -1;
	ref_copy_plane(&f->gmodular.channel[c], out, 2);
	return 0;
}
REF_API int ref_stage_num_planes(ref_stage *s) { return s->inner->frame.gmodular.num_channels; }
REF_API void ref_stage_plane_size(ref_stage *s, int c, int32_t *w, int32_t *h) {
	*w = s->inner->frame.gmodular.channel[c].width; *h = s->inner->frame.gmodular.channel[c].height;
}
REF_API void ref_stage_rgba(ref_stage *s, uint8_t *out) {
	j40__plane *p = &s->inner->rendered_rgba;
	ref_copy_plane(p, out, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* known-answer entry points for single internal functions                                     */

/* in-place 2-D inverse DCT of a (1<<log_rows) x (1<<log_columns) block in j40's canonical
 * coefficient layout (short side = rows); result row-major rows x columns (j40.h:5972) */
REF_API void ref_kat_inverse_dct2d(float *buf, int32_t log_rows, int32_t log_columns) {
	float *scratch = (float *) malloc(sizeof(float) * ((size_t) 1 << (log_rows + log_columns)));
	j40__inverse_dct2d(buf, scratch, log_rows, log_columns);
	free(scratch);
}

/* the 8x8 transforms selected by DctSelect (same dispatch as j40.h:7178-7191) */
REF_API void ref_kat_inverse_by_dctsel(float *buf, float *scratch2, int32_t dctsel) {
	const j40__dct_select *dct = &J40__DCT_SELECT[dctsel];
	switch (dctsel) {
	case 1: j40__inverse_hornuss(buf); break;
	case 2: j40__inverse_dct11(buf); break;
	case 3: j40__inverse_dct22(buf); break;
	case 12: j40__inverse_dct23(buf); break;
	case 13: j40__inverse_dct32(buf); break;
	case 14: j40__inverse_afv(buf, 0, 0); break;
	case 15: j40__inverse_afv(buf, 1, 0); break;
	case 16: j40__inverse_afv(buf, 0, 1); break;
	case 17: j40__inverse_afv(buf, 1, 1); break;
	default: j40__inverse_dct2d(buf, scratch2, dct->log_rows, dct->log_columns); break;
	}
}

/* 1-D inverse DCT of length 1<<t on `rep` interleaved columns (j40.h:5921); both clobbered */
REF_API void ref_kat_inverse_dct(float *out, float *in, int32_t t, int32_t rep) { j40__inverse_dct(out, in, t, rep); }

/* LF -> LLF forward transform (j40.h:5944); buf holds (1<<log_rows) x (1<<log_columns) samples */
REF_API void ref_kat_forward_llf(float *buf, int32_t log_rows, int32_t log_columns) {
	float scratch[1024];
	j40__forward_dct2d_scaled_for_llf(buf, scratch, log_rows, log_columns);
}

REF_API int32_t ref_kat_natural_order(int32_t log_rows, int32_t log_columns, int32_t *out) {
	j40__st stbuf, *st = &stbuf;
	int32_t *order = NULL, size = 1 << (log_rows + log_columns);
	memset(st, 0, sizeof *st);
	if (j40__natural_order(st, log_rows, log_columns, &order)) return 0;
	memcpy(out, order, sizeof(int32_t) * (size_t) size);
	j40__free(order);
	return size;
}

/* library (default) dequantisation table idx (0..16): out = f32[rows*cols][3] */
REF_API int32_t ref_kat_library_dq_matrix(int idx, float *out) {
	j40__st stbuf, *st = &stbuf;
	j40__dq_matrix m;
	int32_t n, i;
	memset(st, 0, sizeof *st);
	memset(&m, 0, sizeof m);
	m.mode = J40__DQ_ENC_LIBRARY;
	if (j40__load_dq_matrix(st, idx, &m)) return 0;
	n = (int32_t) m.n * (int32_t) m.m;
	for (i = 0; i < n; ++i) { out[i * 3] = m.params[i][0]; out[i * 3 + 1] = m.params[i][1]; out[i * 3 + 2] = m.params[i][2]; }
	j40__free(m.params);
	return n;
}

REF_API float ref_kat_half_secant(int i) { return J40__HALF_SECANTS[i]; }
REF_API float ref_kat_lf2llf_scale(int i) { return J40__LF2LLF_SCALES[i]; }

/* the sRGB transfer + int16 quantisation line of j40.h:7233-7235, for transfer-function KATs */
REF_API int16_t ref_kat_srgb_i16(float v, int bpp) {
	v = (v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f);
	return (int16_t) ((float) ((1 << bpp) - 1) * v + 0.5f);
}
REF_API float ref_kat_cbrtf(float v) { return cbrtf(v); }

/* inverse RCT on three int16 planes of w*h (j40.h:4318) */
REF_API void ref_kat_inverse_rct16(int16_t *p0, int16_t *p1, int16_t *p2, int32_t w, int32_t h, int32_t type) {
	j40__st stbuf, *st = &stbuf;
	j40__modular m;
	j40__transform tr;
	j40__plane ch[3];
	int16_t *src[3];
	int i, y;
	memset(st, 0, sizeof *st);
	memset(&m, 0, sizeof m);
	src[0] = p0; src[1] = p1; src[2] = p2;
	for (i = 0; i < 3; ++i) {
		if (j40__init_plane(st, J40__PLANE_I16, w, h, 0, &ch[i])) return;
		for (y = 0; y < h; ++y) memcpy(J40__I16_PIXELS(&ch[i], y), src[i] + (size_t) y * (size_t) w, sizeof(int16_t) * (size_t) w);
	}
	m.channel = ch; m.num_channels = 3;
	tr.rct.tr = J40__TR_RCT; tr.rct.begin_c = 0; tr.rct.type = type;
	j40__inverse_rct16(&m, &tr);
	for (i = 0; i < 3; ++i) {
		for (y = 0; y < h; ++y) memcpy(src[i] + (size_t) y * (size_t) w, J40__I16_PIXELS(&ch[i], y), sizeof(int16_t) * (size_t) w);
		j40__free_plane(&ch[i]);
	}
}

/* ------------------------------------------------------------------------------------------ */
/* restoration filters: the reference's own routines (j40__gaborish j40.h:7271, j40__epf j40.h:7578 with j40__epf_distance 7338,
 * j40__epf_recip_sigmas 7374, j40__epf_step 7427). The reference declares and defines them and never calls them (its decode
 * ignores the frame header's `gab` / `epf` fields, j40.h:5339-5366); here they run on caller-supplied planes so that the HIP
 * kernels and the restatement in hotpath_oracle.c can be held against them. */

/* frame header's RestorationFilter as the reference parsed it (needs a staged decode): out[0] gab.enabled, out[1..6] gab.weights[c][j],
 * out[7] epf.iters, out[8..15] sharp_lut, out[16..18] channel_scale, out[19] quant_mul, out[20] pass0_sigma_scale, out[21]
 * pass2_sigma_scale, out[22] border_sad_mul, out[23] sigma_for_modular */
REF_API void ref_stage_restoration(ref_stage *s, float *out24) {
	j40__frame_st *f = &s->inner->frame;
	int i, j, k = 0;
	out24[k++] = (float) f->gab.enabled;
	for (i = 0; i < 3; ++i) for (j = 0; j < 2; ++j) out24[k++] = f->gab.weights[i][j];
	out24[k++] = (float) f->epf.iters;
	for (i = 0; i < 8; ++i) out24[k++] = f->epf.sharp_lut[i];
	for (i = 0; i < 3; ++i) out24[k++] = f->epf.channel_scale[i];
	out24[k++] = f->epf.quant_mul; out24[k++] = f->epf.pass0_sigma_scale; out24[k++] = f->epf.pass2_sigma_scale;
	out24[k++] = f->epf.border_sad_mul; out24[k++] = f->epf.sigma_for_modular;
}

static int ref_planes_in(j40__st *st, float *const src[3], int32_t w, int32_t h, j40__plane ch[3]) {
	int c, y;
	memset(ch, 0, sizeof(j40__plane) * 3);
	for (c = 0; c < 3; ++c) {
		if (j40__init_plane(st, J40__PLANE_F32, w, h, 0, &ch[c])) return -1;
		for (y = 0; y < h; ++y) memcpy(J40__F32_PIXELS(&ch[c], y), src[c] + (size_t) y * (size_t) w, sizeof(float) * (size_t) w);
	}
	return 0;
}
static void ref_planes_out(float *const dst[3], int32_t w, int32_t h, j40__plane ch[3]) {
	int c, y;
	for (c = 0; c < 3; ++c) {
		if (!ch[c].pixels) continue;
		for (y = 0; y < h; ++y) memcpy(dst[c] + (size_t) y * (size_t) w, J40__F32_PIXELS(&ch[c], y), sizeof(float) * (size_t) w);
		j40__free_plane(&ch[c]);
	}
}

/* j40__gaborish on three tightly packed w*h planes, in place; weights6 = gab.weights[c][j]. Returns the reference's error code. */
REF_API uint32_t ref_kat_gaborish(float *x, float *y, float *b, int32_t w, int32_t h, const float *weights6) {
	j40__st stbuf, *st = &stbuf;
	j40__frame_st *f = (j40__frame_st *) calloc(1, sizeof(j40__frame_st));
	j40__plane ch[3];
	float *p[3];
	int i, j;
	memset(st, 0, sizeof *st);
	st->frame = f;
	f->gab.enabled = 1;
	for (i = 0; i < 3; ++i) for (j = 0; j < 2; ++j) f->gab.weights[i][j] = weights6[i * 2 + j];
	p[0] = x; p[1] = y; p[2] = b;
	if (ref_planes_in(st, p, w, h, ch) == 0) j40__gaborish(st, ch);
	ref_planes_out(p, w, h, ch);
	free(f);
	return st->err;
}

/* one j40__epf_step (j40.h:7427) by itself, as j40__epf would call it (j40.h:7606-7616): step 0 = twelve taps, cross-shaped distances,
 * pass0_sigma_scale; 1 = four taps, cross, scale 1; 2 = four taps, plain distances, pass2_sigma_scale. recip_sigmas: the w8*h8 plane
 * of j40__epf_recip_sigmas. params15 as below. ONLY in the REF_ZEROED_ALLOC build (see the top of this file). */
REF_API uint32_t ref_kat_epf_step(float *x, float *y, float *b, int32_t w, int32_t h, const float *recip_sigmas, int32_t step, const float *params15) {
	static const int32_t K12[][2] = {{0,-2}, {-1,-1}, {-1,0}, {-1,1}, {0,-2}, {0,-1}, {0,1}, {0,2}, {-1,1}, {-1,0}, {-1,1}, {0,2}};  /* as j40.h:7579-7581 */
	static const int32_t K4[][2] = {{0,-1}, {-1,0}, {1,0}, {0,1}};
	j40__st stbuf, *st = &stbuf;
	j40__frame_st *f = (j40__frame_st *) calloc(1, sizeof(j40__frame_st));
	j40__lf_group_st gg;
	j40__plane ch[3], rs = J40__INIT, distances[12][3] = J40__INIT;
	float *p[3];
	int32_t w8 = (w + 7) / 8, h8 = (h + 7) / 8, i, k, c, y8, nk = step == 0 ? 12 : 4;
#ifndef REF_ZEROED_ALLOC
	(void) x; (void) y; (void) b; (void) recip_sigmas; (void) params15; (void) gg; (void) ch; (void) rs; (void) distances; (void) p; (void) w8; (void) h8; (void) i; (void) k; (void) c; (void) y8; (void) nk; (void) st; (void) K12; (void) K4;
	free(f);
	return J40__4("TODO");
#else
	memset(st, 0, sizeof *st); memset(&gg, 0, sizeof gg); memset(ch, 0, sizeof ch);
	st->frame = f;
	for (i = 0; i < 3; ++i) f->epf.channel_scale[i] = params15[8 + i];
	f->epf.border_sad_mul = params15[14];
	gg.width = w; gg.height = h; gg.width8 = w8; gg.height8 = h8;
	p[0] = x; p[1] = y; p[2] = b;
	if (j40__init_plane(st, J40__PLANE_F32, w8, h8, J40__PLANE_FORCE_PAD, &rs)) goto done;
	for (y8 = 0; y8 < h8; ++y8) memcpy(J40__F32_PIXELS(&rs, y8), recip_sigmas + (size_t) y8 * (size_t) w8, sizeof(float) * (size_t) w8);
	for (k = 0; k < nk; ++k) for (c = 0; c < 3; ++c) if (j40__init_plane(st, J40__PLANE_F32, w + 2, h + 2, 0, &distances[k][c])) goto done;
	if (ref_planes_in(st, p, w, h, ch)) goto done;
	j40__epf_step(st, ch, step == 0 ? params15[12] : step == 1 ? 1.0f : params15[13], &rs, nk, step == 0 ? K12 : K4, distances, step != 2, &gg);
done:
	ref_planes_out(p, w, h, ch);
	j40__free_plane(&rs);
	for (k = 0; k < 12; ++k) for (c = 0; c < 3; ++c) j40__free_plane(&distances[k][c]);
	free(f);
	return st->err;
#endif
}

/* j40__epf on three tightly packed w*h planes, in place, with ONE LfGroup-shaped record spanning the whole picture (the reference
 * notes that the filters run over the entire image, j40.h:7268): `sharpness` is the w8*h8 map (int16, as decoded), `hfmul_inv` the
 * HfMul reciprocal of the varblock covering each 8x8 cell (w8*h8; the harness gives every cell a varblock record of its own, which
 * needs w8*h8 <= 2^20 -- the width of the reference's varblock index). params15 = sharp_lut[8], channel_scale[3], quant_mul,
 * pass0_sigma_scale, pass2_sigma_scale, border_sad_mul. sigma_out (optional): j40__epf_recip_sigmas' plane (w8*h8). */
REF_API uint32_t ref_kat_epf(float *x, float *y, float *b, int32_t w, int32_t h, const int16_t *sharpness, const float *hfmul_inv,
		int32_t iters, const float *params15, float *sigma_out) {
	j40__st stbuf, *st = &stbuf;
	j40__frame_st *f = (j40__frame_st *) calloc(1, sizeof(j40__frame_st));
	j40__lf_group_st gg;
	j40__plane ch[3];
	float *p[3];
	int32_t w8 = (w + 7) / 8, h8 = (h + 7) / 8, i, y8, x8;
	memset(st, 0, sizeof *st);
	memset(&gg, 0, sizeof gg);
	memset(ch, 0, sizeof ch);
	st->frame = f;
	if ((int64_t) w8 * h8 > (1 << 20)) { free(f); return J40__4("rnge"); }
	f->is_modular = 0;
	f->epf.iters = iters;
	for (i = 0; i < 8; ++i) f->epf.sharp_lut[i] = params15[i];
	for (i = 0; i < 3; ++i) f->epf.channel_scale[i] = params15[8 + i];
	f->epf.quant_mul = params15[11]; f->epf.pass0_sigma_scale = params15[12]; f->epf.pass2_sigma_scale = params15[13]; f->epf.border_sad_mul = params15[14];
	f->epf.sigma_for_modular = 1.0f;
	gg.width = w; gg.height = h; gg.width8 = w8; gg.height8 = h8; gg.width64 = (w + 63) / 64; gg.height64 = (h + 63) / 64;
	gg.nb_varblocks = w8 * h8;
	gg.varblocks = (j40__varblock *) calloc((size_t) (w8 * h8), sizeof(j40__varblock));
	p[0] = x; p[1] = y; p[2] = b;
	if (!gg.varblocks || j40__init_plane(st, J40__PLANE_I16, w8, h8, 0, &gg.sharpness) || j40__init_plane(st, J40__PLANE_I32, w8, h8, 0, &gg.blocks)) goto done;
	for (y8 = 0; y8 < h8; ++y8) for (x8 = 0; x8 < w8; ++x8) {
		int32_t cell = y8 * w8 + x8;
		J40__I16_PIXELS(&gg.sharpness, y8)[x8] = sharpness[cell];
		J40__I32_PIXELS(&gg.blocks, y8)[x8] = cell | (2 << 20);
		gg.varblocks[cell].hfmul.inv = hfmul_inv[cell];
	}
	if (sigma_out) {
		j40__plane rs = J40__INIT;
		if (!j40__epf_recip_sigmas(st, &gg, &rs)) { ref_copy_plane(&rs, sigma_out, 4); j40__free_plane(&rs); }
		if (st->err) goto done;
	}
	if (ref_planes_in(st, p, w, h, ch) == 0) j40__epf(st, ch, &gg);
done:
	ref_planes_out(p, w, h, ch);
	j40__free_plane(&gg.sharpness); j40__free_plane(&gg.blocks);
	free(gg.varblocks); free(f);
	return st->err;
}
