/*
 * oracle/hotpath_oracle.c -- CPU restatement of the reference's algorithm for the hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker. The product (build/libj40hip.so) never links
 * or calls it and has no CPU path of its own.
 *
 * Plain scalar C99 that follows the structure of lifthrasiir/j40 (single-threaded, one pass over the
 * sections, then LF group by LF group), driven behind the same seam as the HIP kernels through the
 * plan views of include/j40hip.h. Every routine cites the reference lines it restates
 * (/root/reference/j40.h).
 *
 * Pinning: tests/test_oracle.py checks this restatement against the unmodified reference
 * (oracle/_ref/libj40ref.so, compiled from /root/reference by oracle/Makefile) on the whole stream
 * matrix and against the committed golden fixtures (tests/golden/manifest.json): bit-exact RGBA for
 * Modular and -- built, like the reference, without FMA contraction and with the same libm -- for
 * VarDCT as well. Squeeze is not implemented by the reference (j40.h:3812, 4518): the inverse step here restates ISO 18181-1
 * instead -- PARITY UNPINNED for that one transform (no libjxl in this image); tests pin it by lossless round trips.
 *
 * Build: gcc -O2 -ffp-contract=off (see oracle/Makefile); no -march=native (j40.h:5834).
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/j40hip.h"

#define ORACLE_API __attribute__((visibility("default")))
#define E4(a, b, c, d) (((uint32_t) (a) << 24) | ((uint32_t) (b) << 16) | ((uint32_t) (c) << 8) | (uint32_t) (d))

/* ---------------------------------------------------------------------------------------------- */
/* bit reader: LSB-first, bytes enter at the top of the accumulator (j40.h:1847-1922)              */

typedef struct { const uint8_t *p, *end; uint64_t bits; int nbits; uint32_t err; } obits;

static void obits_init(obits *b, const uint8_t *base, uint32_t byte_off, uint32_t size, uint32_t bit_off) {
	b->p = base + byte_off + (bit_off >> 3); b->end = base + byte_off + size; b->bits = 0; b->nbits = 0; b->err = 0;
	if (bit_off & 7) {
		if (b->p < b->end) { b->bits = (uint64_t) *b->p++ >> (bit_off & 7); b->nbits = 8 - (int) (bit_off & 7); }
		else b->err = E4('s', 'h', 'r', 't');
	}
}
static void obits_fill(obits *b) { while (b->nbits <= 56 && b->p < b->end) { b->bits |= (uint64_t) *b->p++ << b->nbits; b->nbits += 8; } }
static uint32_t obits_u(obits *b, int n) {
	uint32_t v;
	if (b->nbits < n) { obits_fill(b); if (b->nbits < n) { if (!b->err) b->err = E4('s', 'h', 'r', 't'); b->bits = 0; b->nbits = 0; return 0; } }
	v = (uint32_t) (b->bits & (((uint64_t) 1 << n) - 1));
	b->bits >>= n; b->nbits -= n;
	return v;
}
static void obits_finish(obits *b, const uint8_t *base, uint32_t declared_end) {  /* single-section frames: j40.h:8203, 7796-7803 */
	int n = b->nbits & 7;
	if (b->bits & (((uint64_t) 1 << n) - 1)) { if (!b->err) b->err = E4('p', 'a', 'd', '0'); }
	b->bits >>= n; b->nbits -= n;
	{ const uint32_t at = (uint32_t) (b->p - base) - (uint32_t) (b->nbits >> 3);
	  if (!b->err && at < declared_end) b->err = E4('s', 'h', 'r', 't');
	  else if (!b->err && at > declared_end) b->err = E4('e', 'x', 'c', 's'); }
}

/* ---------------------------------------------------------------------------------------------- */
/* entropy code: alias tables (j40.h:2362), rANS step (2441), prefix codes (2256), hybrid ints (2313), */
/* LZ77 (2804-2876)                                                                                */

typedef struct { int16_t cutoff, offset_or_next, symbol; } oalias;
typedef struct {
	const j40hip_codespec_view *spec;
	oalias **alias;          /* per cluster, ANS */
	uint32_t ans_state;
	int32_t num_to_copy, copy_pos, num_decoded;
	int32_t *window;
} ocode;

static oalias *build_alias(const int16_t *D, int log_alpha) {
	int16_t log_bucket = (int16_t) (12 - log_alpha), bucket = (int16_t) (1 << log_bucket), size = (int16_t) (1 << log_alpha);
	oalias *t = (oalias *) calloc((size_t) size, sizeof(oalias));
	int16_t u = -1, o = -1, i, j;
	for (i = 0; i < size && !D[i]; ++i);
	for (j = (int16_t) (i + 1); j < size && !D[j]; ++j);
	if (i < size && j >= size) {
		for (j = 0; j < size; ++j) { t[j].symbol = i; t[j].offset_or_next = (int16_t) (j << log_bucket); t[j].cutoff = 0; }
		return t;
	}
	for (i = 0; i < size; ++i) {
		t[i].cutoff = D[i];
		if (D[i] > bucket) { t[i].offset_or_next = o; o = i; }
		else if (D[i] < bucket) { t[i].offset_or_next = u; u = i; }
		else { t[i].symbol = i; t[i].offset_or_next = 0; }
	}
	while (o >= 0 && u >= 0) {
		int16_t by = (int16_t) (bucket - t[u].cutoff), next_u = t[u].offset_or_next, tmp;
		t[o].cutoff = (int16_t) (t[o].cutoff - by);
		t[u].symbol = o;
		t[u].offset_or_next = (int16_t) (t[o].cutoff - t[u].cutoff);
		u = next_u;
		if (t[o].cutoff < bucket) { tmp = t[o].offset_or_next; t[o].offset_or_next = u; u = o; o = tmp; }
		else if (t[o].cutoff == bucket) { tmp = t[o].offset_or_next; t[o].offset_or_next = 0; o = tmp; }
	}
	return t;
}

static void ocode_init(ocode *c, const j40hip_codespec_view *spec) {
	int i;
	memset(c, 0, sizeof *c);
	c->spec = spec;
	if (!spec->use_prefix_code) {
		c->alias = (oalias **) calloc((size_t) spec->num_clusters, sizeof(oalias *));
		for (i = 0; i < spec->num_clusters; ++i) c->alias[i] = build_alias(spec->clusters[i].D, spec->log_alpha_size);
	}
}
static void ocode_free(ocode *c) {
	int i;
	if (c->alias) { for (i = 0; i < c->spec->num_clusters; ++i) free(c->alias[i]); free(c->alias); }
	free(c->window);
}
static void ocode_restart(ocode *c) { c->ans_state = 0; c->num_to_copy = c->copy_pos = c->num_decoded = 0; }

static int32_t hybrid(obits *b, int32_t token, int split_exp, int msb, int lsb) {
	int32_t split = 1 << split_exp, max_token = split + ((30 - split_exp) << (lsb + msb)) - 1, in_token, midbits, mid, top, lo, hi;
	if (token < split) return token;
	if (token > max_token) { token = max_token; if (!b->err) b->err = E4('i', 'o', 'v', 'f'); }
	in_token = msb + lsb;
	midbits = split_exp - in_token + ((token - split) >> in_token);
	mid = (int32_t) obits_u(b, midbits);
	top = 1 << msb; lo = token & ((1 << lsb) - 1); hi = (token >> lsb) & (top - 1);
	return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
}

/* canonical prefix code decoded one bit at a time from the code lengths (RFC 7932 section 3.2) */
static int32_t prefix_symbol(obits *b, const j40hip_cluster_view *cl) {
	int32_t count[16] = {0}, first_code = 0, first_index = 0, code = 0, len, s, only = -1, marked = 0;
	for (s = 0; s < cl->alphabet_size; ++s) { if (cl->lengths[s] == 255) only = s; else if (cl->lengths[s] == (2 | 128)) ++marked; else ++count[cl->lengths[s]]; }
	if (only >= 0) return only;
	if (marked == 4) {
		/* simple code, NSYM = 4, tree-select 0: the reference's table holds the sorted symbols at the index made of the two bits in
		 * read order (template {2,2,2,2} with symref {0,1,2,3}, j40.h:2090, 2112) -- not RFC 7932's canonical assignment */
		int32_t k = (int32_t) obits_u(b, 1);
		k |= (int32_t) obits_u(b, 1) << 1;
		if (b->err) return 0;
		for (s = 0; s < cl->alphabet_size; ++s) if (cl->lengths[s] == (2 | 128) && k-- == 0) return s;
	}
	if (cl->alphabet_size <= 1) return 0;
	count[0] = 0;
	for (len = 1; len <= 15; ++len) {
		int32_t n = count[len];
		code = (code << 1) | (int32_t) obits_u(b, 1);
		if (b->err) return 0;
		if (code - first_code < n) {  /* the (code - first_code)-th symbol of this length, in symbol order */
			int32_t k = code - first_code;
			for (s = 0; s < cl->alphabet_size; ++s) if (cl->lengths[s] == len && k-- == 0) return s;
		}
		first_index += n; first_code = (first_code + n) << 1;
	}
	if (!b->err) b->err = E4('h', 'u', 'f', 'd');
	return 0;
}

static int32_t ans_symbol(obits *b, ocode *c, int cluster) {
	const j40hip_codespec_view *spec = c->spec;
	int log_bucket = 12 - spec->log_alpha_size;
	int32_t index, i, pos, symbol, offset;
	const oalias *bk;
	if (c->ans_state == 0) { c->ans_state = obits_u(b, 16); c->ans_state |= obits_u(b, 16) << 16; }
	index = (int32_t) (c->ans_state & 0xfff); i = index >> log_bucket; pos = index & ((1 << log_bucket) - 1);
	bk = &c->alias[cluster][i];
	symbol = pos < bk->cutoff ? i : bk->symbol;
	offset = pos < bk->cutoff ? 0 : bk->offset_or_next;
	c->ans_state = (uint32_t) spec->clusters[cluster].D[symbol] * (c->ans_state >> 12) + (uint32_t) offset + (uint32_t) pos;
	if (c->ans_state < (1u << 16)) c->ans_state = (c->ans_state << 16) | obits_u(b, 16);
	return symbol;
}

static int32_t cluster_token(obits *b, ocode *c, int cluster) {
	return c->spec->use_prefix_code ? prefix_symbol(b, &c->spec->clusters[cluster]) : ans_symbol(b, c, cluster);
}

static int32_t ocode_symbol(obits *b, ocode *c, int32_t ctx, int32_t dist_mult) {
	static const uint8_t SPECIAL[120] = {  /* LZ77 special distances (dx + 7) * 16 + dy, spec table (cf. j40.h:2834) */
		0x71, 0x80, 0x81, 0x61, 0x72, 0x90, 0x82, 0x62, 0x91, 0x51, 0x92, 0x52, 0x73, 0xa0, 0x83, 0x63, 0xa1, 0x41, 0x93, 0x53, 0xa2, 0x42, 0x74, 0xb0,
		0x84, 0x64, 0xb1, 0x31, 0xa3, 0x43, 0x94, 0x54, 0xb2, 0x32, 0x75, 0xa4, 0x44, 0xb3, 0x33, 0xc0, 0x85, 0x65, 0xc1, 0x21, 0x95, 0x55, 0xc2, 0x22,
		0xb4, 0x34, 0xa5, 0x45, 0xc3, 0x23, 0x76, 0xd0, 0x86, 0x66, 0xd1, 0x11, 0x96, 0x56, 0xd2, 0x12, 0xb5, 0x35, 0xc4, 0x24, 0xa6, 0x46, 0xd3, 0x13,
		0x77, 0xe0, 0x87, 0x67, 0xc5, 0x25, 0xe1, 0x01, 0xb6, 0x36, 0xd4, 0x14, 0x97, 0x57, 0xe2, 0x02, 0xa7, 0x47, 0xe3, 0x03, 0xc6, 0x26, 0xd5, 0x15,
		0xf0, 0xb7, 0x37, 0xe4, 0x04, 0xf1, 0xf2, 0xd6, 0x16, 0xf3, 0xc7, 0x27, 0xe5, 0x05, 0xf4, 0xd7, 0x17, 0xe6, 0x06, 0xf5, 0xe7, 0x07, 0xf6, 0xf7};
	const j40hip_codespec_view *spec = c->spec;
	const int32_t MASK = 0xfffff;
	int cluster, token;
	if (c->num_to_copy > 0) {
		--c->num_to_copy;
		return c->window[c->num_decoded++ & MASK] = c->window[c->copy_pos++ & MASK];
	}
	cluster = spec->cluster_map[ctx];
	token = cluster_token(b, c, cluster);
	if (spec->lz77_enabled && token >= spec->min_symbol) {
		int lz = spec->cluster_map[spec->num_dist - 1];
		int32_t num_to_copy = hybrid(b, token - spec->min_symbol, spec->lz_len_split_exp, spec->lz_len_msb, spec->lz_len_lsb) + spec->min_length;
		int32_t distance;
		token = cluster_token(b, c, lz);
		distance = hybrid(b, token, spec->clusters[lz].split_exp, spec->clusters[lz].msb_in_token, spec->clusters[lz].lsb_in_token);
		if (b->err) return 0;
		if (!dist_mult) ++distance;
		else if (distance >= 120) distance -= 119;
		else { int32_t sp = SPECIAL[distance]; distance = ((sp >> 4) - 7) + dist_mult * (sp & 7); if (distance < 1) distance = 1; }
		if (distance > c->num_decoded) distance = c->num_decoded;
		if (distance > (1 << 20)) distance = 1 << 20;
		c->copy_pos = c->num_decoded - distance;
		if (!c->window) c->window = (int32_t *) calloc((size_t) 1 << 20, sizeof(int32_t));
		c->num_to_copy = num_to_copy - 1;
		return c->window[c->num_decoded++ & MASK] = c->window[c->copy_pos++ & MASK];
	}
	token = hybrid(b, token, spec->clusters[cluster].split_exp, spec->clusters[cluster].msb_in_token, spec->clusters[cluster].lsb_in_token);
	if (spec->lz77_enabled) {
		if (!c->window) c->window = (int32_t *) calloc((size_t) 1 << 20, sizeof(int32_t));
		c->window[c->num_decoded++ & MASK] = token;
	}
	return token;
}

static void ocode_finish(obits *b, ocode *c) {  /* j40.h:2884 */
	if (c->spec->use_prefix_code) return;
	if (c->ans_state) { if (c->ans_state != 0x130000 && !b->err) b->err = E4('a', 'n', 's', '?'); }
	else { uint32_t lo = obits_u(b, 16), hi = obits_u(b, 16); if ((lo != 0 || hi != 0x13) && !b->err) b->err = E4('a', 'n', 's', '?'); }
}

static int32_t unpack_signed(int32_t x) { return (x & 1) ? -(x / 2 + 1) : x / 2; }

/* ---------------------------------------------------------------------------------------------- */
/* VarDCT                                                                                         */

/* DctSelect -> log rows, log columns, dequant parameter set, order (spec table; cf. j40.h:4591) */
static const int8_t DCTSEL[27][4] = {
	{3, 3, 0, 0}, {3, 3, 1, 1}, {3, 3, 2, 1}, {3, 3, 3, 1}, {4, 4, 4, 2}, {5, 5, 5, 3}, {4, 3, 6, 4}, {3, 4, 6, 4}, {5, 3, 7, 5}, {3, 5, 7, 5}, {5, 4, 8, 6}, {4, 5, 8, 6},
	{3, 3, 9, 1}, {3, 3, 9, 1}, {3, 3, 10, 1}, {3, 3, 10, 1}, {3, 3, 10, 1}, {3, 3, 10, 1}, {6, 6, 11, 7}, {6, 5, 12, 8}, {5, 6, 12, 8}, {7, 7, 13, 9}, {7, 6, 14, 10},
	{6, 7, 14, 10}, {8, 8, 15, 11}, {8, 7, 16, 12}, {7, 8, 16, 12}};

/* j40__hf_coeffs (j40.h:6888-7005) for one section; coeffs[c] = the LF group's coefficient arrays */
static uint32_t hf_coeffs(const j40hip_vardct_view *v, int pass, const j40hip_section_view *sec, ocode *code, float *const coeffs[3]) {
	static const int8_t FREQ2[64] = {-1, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 30, 32, 32, 34, 34, 36, 36, 38, 38, 40, 40, 42, 42, 44, 44,
		46, 46, 46, 46, 48, 48, 48, 48, 50, 50, 50, 50, 52, 52, 52, 52, 54, 54, 54, 54, 56, 56, 56, 56, 58, 58, 58, 58, 60, 60, 60, 60};
	static const int16_t NNZ2[8] = {0, 62, 124, 186, 246, 304, 360, 412}, NNZ_LIMIT[8] = {2, 3, 5, 9, 13, 21, 33, 64};
	const j40hip_lf_group_view *gg = &v->lf_groups[sec->ggidx];
	int32_t gw8 = (sec->gw + 7) / 8, gh8 = (sec->gh + 7) / 8;
	int32_t lfidx_size = (v->nb_lf_thr[0] + 1) * (v->nb_lf_thr[1] + 1) * (v->nb_lf_thr[2] + 1);
	int8_t (*nonzeros)[3] = (int8_t (*)[3]) malloc((size_t) (gw8 * gh8) * 3);
	int32_t x8, y8, i, j, c_yxb, preset_bits = 0, ctxoff;
	obits b;
	obits_init(&b, v->codestream, sec->byte_off, sec->size, sec->bit_off);
	while ((1 << preset_bits) < v->num_hf_presets) ++preset_bits;
	ctxoff = 495 * v->nb_block_ctx * (int32_t) obits_u(&b, preset_bits);  /* j40.h:7020 */
	ocode_restart(code);
	for (y8 = 0; y8 < gh8 && !b.err; ++y8) for (x8 = 0; x8 < gw8 && !b.err; ++x8) {
		int32_t ggx8 = x8 + sec->gx_in_gg / 8, ggy8 = y8 + sec->gy_in_gg / 8, nzpos = y8 * gw8 + x8;
		int32_t voff = gg->blocks[ggy8 * gg->width8 + ggx8], dctsel = voff >> 20;
		int32_t log_rows, log_columns, log_size, coeffoff, qfidx, lfidx, bctx0, bctxc;
		if (dctsel < 2) continue;
		dctsel -= 2; voff &= 0xfffff;
		log_rows = DCTSEL[dctsel][0]; log_columns = DCTSEL[dctsel][1]; log_size = log_rows + log_columns;
		coeffoff = gg->coeffoff_qfidx[voff] & ~15; qfidx = gg->coeffoff_qfidx[voff] & 15;
		lfidx = gg->lfindices[ggy8 * gg->width8 + ggx8];
		bctx0 = (DCTSEL[dctsel][3] * (v->nb_qf_thr + 1) + qfidx) * lfidx_size + lfidx;
		bctxc = 13 * (v->nb_qf_thr + 1) * lfidx_size;
		for (c_yxb = 0; c_yxb < 3 && !b.err; ++c_yxb) {
			int32_t c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			const int32_t *order = v->orders[(pass * 13 + DCTSEL[dctsel][3]) * 3 + c];
			int32_t bctx = v->block_ctx_map[bctx0 + bctxc * c_yxb], nz, nzctx, cctx, qnz, prev;
			nz = x8 > 0 ? (y8 > 0 ? (nonzeros[nzpos - 1][c] + nonzeros[nzpos - gw8][c] + 1) >> 1 : nonzeros[nzpos - 1][c]) : (y8 > 0 ? nonzeros[nzpos - gw8][c] : 32);
			nzctx = ctxoff + bctx + (nz < 8 ? nz : 4 + nz / 2) * v->nb_block_ctx;
			nz = ocode_symbol(&b, code, nzctx, 0);
			if (nz > (63 << (log_size - 6))) { b.err = E4('c', 'o', 'e', 'f'); break; }
			qnz = (nz + (1 << (log_size - 6)) - 1) >> (log_size - 6);
			for (i = 0; i < (1 << (log_rows - 3)); ++i) for (j = 0; j < (1 << (log_columns - 3)); ++j) nonzeros[nzpos + i * gw8 + j][c] = (int8_t) qnz;
			cctx = ctxoff + 458 * bctx + 37 * v->nb_block_ctx;
			prev = nz <= (1 << (log_size - 4));
			for (i = 1 << (log_size - 6); nz > 0 && i < (1 << log_size); ++i) {
				int32_t q = (nz + (1 << (log_size - 6)) - 1) >> (log_size - 6), k = 0, ucoeff;
				while (q >= NNZ_LIMIT[k]) ++k;
				ucoeff = ocode_symbol(&b, code, cctx + NNZ2[k] + FREQ2[i >> (log_size - 6)] + prev, 0);
				coeffs[c][coeffoff + order[i]] += (float) unpack_signed(ucoeff);
				nz -= prev = (ucoeff != 0);
				if (b.err) break;
			}
			if (nz != 0 && !b.err) b.err = E4('c', 'o', 'e', 'f');
		}
	}
	if (!b.err) ocode_finish(&b, code);
	/* extra channels: the group's Modular sub-image follows (j40.h:7024-7034); the reference decodes and then drops it
	 * (j40.h:7868-7870), this restatement of the pixel path stops at the coefficients */
	if (!b.err && v->check_section_end) obits_finish(&b, v->codestream, v->single_declared_end);   /* never in frames with several sections: j40.h:7778-7795 drops that error */
	free(nonzeros);
	return b.err;
}

/* inverse DCT family; half-secant table regenerated exactly like the reference's literals
 * (formula printed with 8 resp. 7 decimals, j40.h:5689-5730) */
static float HS[256];
static void init_hs(void) {
	static int done = 0;
	int n, k;
	char buf[64];
	if (done) return;
	for (n = 1; n <= 7; ++n) for (k = 0; k < (1 << n); ++k) {
		double x = 1.0 / (2.0 * cos(((double) k + 0.5) * 3.14159265358979323846 / (double) (1 << (n + 1))));
		snprintf(buf, sizeof buf, x < 10.0 ? "%.8f" : "%.7f", x);
		HS[(1 << n) + k] = strtof(buf, NULL);
	}
	done = 1;
}
/* 1-D IDCT of length 1 << t over `rep` interleaved columns, out-of-place with both buffers used
 * as scratch, recursion as in j40__inverse_dct_core (j40.h:5802-5841) */
static void idct(float *out, float *in, int t, int rep) {
	int N = 1 << t, i, r;
	if (t == 0) { memcpy(out, in, sizeof(float) * (size_t) rep); return; }
	if (t == 1) { for (r = 0; r < rep; ++r) { float x = in[r], y = in[rep + r]; out[r] = x + y; out[rep + r] = x - y; } return; }
	for (i = 0; i < N / 2; ++i) for (r = 0; r < rep; ++r) out[i * rep + r] = in[i * 2 * rep + r];
	for (r = 0; r < rep; ++r) out[N / 2 * rep + r] = 1.4142135623730951f * in[rep + r];
	for (i = 1; i < N / 2; ++i) for (r = 0; r < rep; ++r) out[(N / 2 + i) * rep + r] = in[(i * 2 - 1) * rep + r] + in[(i * 2 + 1) * rep + r];
	idct(in, out, t - 1, rep);
	idct(in + N / 2 * rep, out + N / 2 * rep, t - 1, rep);
	for (i = 0; i < N / 2; ++i) {
		float mult = HS[N / 2 + i];
		for (r = 0; r < rep; ++r) { float x = in[i * rep + r], y = in[(N / 2 + i) * rep + r]; out[i * rep + r] = x + y * mult; out[(N - i - 1) * rep + r] = x - y * mult; }
	}
}

static void transpose(float *out, const float *in, int rows, int cols) { int y, x; for (y = 0; y < rows; ++y) for (x = 0; x < cols; ++x) out[x * rows + y] = in[y * cols + x]; }

static void idct2d(float *buf, float *scratch, int log_rows, int log_columns) {  /* j40.h:5972 */
	int R = 1 << log_rows, C = 1 << log_columns;
	if (log_columns > log_rows) transpose(scratch, buf, R, C); else memcpy(scratch, buf, sizeof(float) * (size_t) (R * C));
	idct(buf, scratch, log_columns, R);   /* scratch is [C][R] */
	transpose(scratch, buf, C, R);        /* -> [R][C] */
	idct(buf, scratch, log_rows, C);
}

static void aux2x2(float *out, const float *in, int x, int y, int S2) {  /* j40.h:5993 */
	int p = y * 8 + x, q = (y * 2) * 8 + (x * 2);
	float c00 = in[p], c01 = in[p + S2], c10 = in[p + S2 * 8], c11 = in[p + S2 * 9];
	out[q] = c00 + c01 + c10 + c11; out[q + 1] = c00 + c01 - c10 - c11; out[q + 8] = c00 - c01 + c10 - c11; out[q + 9] = c00 - c01 - c10 + c11;
}

static const float AFV_BASIS[256] = {  /* ISO 18181-1 AFVBasis, stored [sample][coefficient] (cf. j40.h:6108) */
	0.25000000f, 0.87690293f, 0, 0, 0, -0.41053776f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0.25000000f, 0.22065181f, 0, 0, -0.70710678f, 0.62354854f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0.25000000f, -0.10140050f, 0.40670076f, -0.21255748f, 0, -0.06435072f, -0.45175566f, -0.30468475f, 0.30179295f, 0.40824829f, 0.17478670f, -0.21105601f, -0.14266085f, -0.13813540f, -0.17437603f, 0.11354987f,
	0.25000000f, -0.10140050f, 0.44444817f, 0.30854971f, 0, -0.06435072f, 0.15854504f, 0.51126161f, 0.25792363f, 0, 0.08126112f, 0.18567181f, -0.34164468f, 0.33022826f, 0.07027907f, -0.07417505f,
	0.25000000f, 0.22065181f, 0, 0, 0.70710678f, 0.62354854f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0.25000000f, -0.10140050f, 0, 0.47067023f, 0, -0.06435072f, -0.04038515f, 0, 0.16272340f, 0, 0, 0, 0.73674975f, 0.08755115f, -0.29210266f, 0.19402893f,
	0.25000000f, -0.10140050f, 0.19574399f, -0.16212052f, 0, -0.06435072f, 0.00741823f, -0.29048013f, 0.09520023f, 0, -0.36753980f, 0.49215859f, 0.24627108f, -0.07946707f, 0.36238173f, -0.43519050f,
	0.25000000f, -0.10140050f, 0.29291001f, 0, 0, -0.06435072f, 0.39351034f, -0.06578702f, 0, -0.40824829f, -0.30788221f, -0.38525014f, -0.08574019f, -0.46133749f, 0, 0.21918685f,
	0.25000000f, -0.10140050f, -0.40670076f, -0.21255748f, 0, -0.06435072f, -0.45175566f, 0.30468475f, 0.30179295f, -0.40824829f, -0.17478670f, 0.21105601f, -0.14266085f, -0.13813540f, -0.17437603f, 0.11354987f,
	0.25000000f, -0.10140050f, -0.19574399f, -0.16212052f, 0, -0.06435072f, 0.00741823f, 0.29048013f, 0.09520023f, 0, 0.36753980f, -0.49215859f, 0.24627108f, -0.07946707f, 0.36238173f, -0.43519050f,
	0.25000000f, -0.10140050f, 0, -0.47067023f, 0, -0.06435072f, 0.11074166f, 0, -0.16272340f, 0, 0, 0, 0.14883399f, 0.49724647f, 0.29210266f, 0.55504438f,
	0.25000000f, -0.10140050f, 0.11379074f, -0.14642919f, 0, -0.06435072f, 0.08298163f, -0.23889774f, -0.35312385f, -0.40824829f, 0.48266891f, 0.17419413f, -0.04768680f, 0.12538059f, -0.43266080f, -0.25468277f,
	0.25000000f, -0.10140050f, -0.44444817f, 0.30854971f, 0, -0.06435072f, 0.15854504f, -0.51126161f, 0.25792363f, 0, -0.08126112f, -0.18567181f, -0.34164468f, 0.33022826f, 0.07027907f, -0.07417505f,
	0.25000000f, -0.10140050f, -0.29291001f, 0, 0, -0.06435072f, 0.39351034f, 0.06578702f, 0, 0.40824829f, 0.30788221f, 0.38525014f, -0.08574019f, -0.46133749f, 0, 0.21918685f,
	0.25000000f, -0.10140050f, -0.11379074f, -0.14642919f, 0, -0.06435072f, 0.08298163f, 0.23889774f, -0.35312385f, 0.40824829f, -0.48266891f, -0.17419413f, -0.04768680f, 0.12538059f, -0.43266080f, -0.25468277f,
	0.25000000f, -0.10140050f, 0, 0.42511496f, 0, -0.06435072f, -0.45175566f, 0, -0.60358590f, 0, 0, 0, -0.14266085f, -0.13813540f, 0.34875205f, 0.11354987f};

static void inverse_8x8_special(int dctsel, float *buf) {  /* j40.h:6002-6246 */
	float s[64];
	int x, y, ix, iy, i, j;
	switch (dctsel) {
	case 2:  /* DCT2x2 pyramid */
		aux2x2(buf, buf, 0, 0, 1);
		memcpy(s, buf, sizeof s);
		for (y = 0; y < 2; ++y) for (x = 0; x < 2; ++x) aux2x2(s, buf, x, y, 2);
		for (y = 0; y < 4; ++y) for (x = 0; x < 4; ++x) aux2x2(buf, s, x, y, 4);
		break;
	case 3:  /* four DCT4x4 */
		aux2x2(buf, buf, 0, 0, 1);
		idct(s, buf, 2, 16);
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[x * 8 + y] = s[y * 8 + x];
		idct(s, buf, 2, 16);
		for (y = 0; y < 4; ++y) for (x = 0; x < 4; ++x) {
			buf[y * 8 + x] = s[(y * 2) * 8 + (x * 2)]; buf[y * 8 + (x + 4)] = s[(y * 2 + 1) * 8 + (x * 2)];
			buf[(y + 4) * 8 + x] = s[(y * 2) * 8 + (x * 2 + 1)]; buf[(y + 4) * 8 + (x + 4)] = s[(y * 2 + 1) * 8 + (x * 2 + 1)];
		}
		break;
	case 1:  /* Hornuss */
		memcpy(s, buf, sizeof s);
		aux2x2(s, buf, 0, 0, 1);
		for (y = 0; y < 2; ++y) for (x = 0; x < 2; ++x) {
			int pos00 = y * 8 + x, pos11 = (y + 2) * 8 + (x + 2);
			float rsum[4] = {0, 0, 0, 0}, sample11;
			for (iy = 0; iy < 4; ++iy) for (ix = 0; ix < 4; ++ix) rsum[ix] += s[(y + iy * 2) * 8 + (x + ix * 2)];
			sample11 = s[pos00] - (rsum[0] + rsum[1] + rsum[2] + rsum[3] - s[pos00]) * 0.0625f;
			s[pos00] = s[pos11]; s[pos11] = 0.0f;
			for (iy = 0; iy < 4; ++iy) for (ix = 0; ix < 4; ++ix) buf[(4 * y + iy) * 8 + (4 * x + ix)] = s[(y + iy * 2) * 8 + (x + ix * 2)] + sample11;
		}
		break;
	case 13: {  /* DCT8x4 */
		float t = buf[0] + buf[8]; buf[8] = buf[0] - buf[8]; buf[0] = t;
		idct(s, buf, 2, 16);
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[x * 8 + y] = s[y * 8 + x];
		idct(s, buf, 3, 8);
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[y * 8 + (((x & 1) << 2) | (x >> 1))] = s[y * 8 + x];
		break;
	}
	case 12:  /* DCT4x8 */
		memcpy(s, buf, sizeof s);
		s[0] = buf[0] + buf[8]; s[8] = buf[0] - buf[8];
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[x * 8 + y] = s[y * 8 + x];
		idct(s, buf, 3, 8);
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[x * 8 + y] = s[y * 8 + x];
		idct(s, buf, 2, 16);
		for (y = 0; y < 8; ++y) for (x = 0; x < 8; ++x) buf[(((y & 1) << 2) | (y >> 1)) * 8 + x] = s[y * 8 + x];
		break;
	default: {  /* AFV0..3 */
		int flipx = (dctsel - 14) & 1, flipy = (dctsel - 14) >> 1;
		float *bufafv = buf, *buf22 = buf + 16, *buf32 = buf + 32, *safv = s, *s22 = s + 16, *s32 = s + 32;
		for (y = 0; y < 8; y += 2) for (x = 0; x < 8; ++x) s[(x % 2) * 16 + (y / 2) * 4 + (x / 2)] = buf[y * 8 + x];
		for (y = 1; y < 8; y += 2) for (x = 0; x < 8; ++x) s32[x * 4 + (y / 2)] = buf[y * 8 + x];
		safv[0] = (buf[0] + buf[1] + buf[8]) * 4.0f; s22[0] = buf[0] - buf[1] + buf[8]; s32[0] = buf[0] - buf[8];
		for (i = 0; i < 16; ++i) { float sum = 0.0f; for (j = 0; j < 16; ++j) sum += safv[j] * AFV_BASIS[i * 16 + j]; bufafv[i] = sum; }
		idct(buf22, s22, 2, 4);
		idct(buf32, s32, 3, 4);
		for (y = 0; y < 4; ++y) { for (x = 0; x < 4; ++x) safv[y * 4 + x] = bufafv[y * 4 + x]; for (x = 0; x < 4; ++x) s22[x * 4 + y] = buf22[y * 4 + x]; }
		for (y = 0; y < 8; ++y) for (x = 0; x < 4; ++x) s32[x * 8 + y] = buf32[y * 4 + x];
		idct(buf22, s22, 2, 4);
		idct(buf32, s32, 2, 8);
		memcpy(s + 16, buf + 16, sizeof(float) * 48);
		for (y = 0; y < 4; ++y) {
			int ay = flipy ? 7 - y : y, p22 = (flipy * 4 + y) * 8 + (!flipx * 4), p23 = (!flipy * 4 + y) * 8;
			for (x = 0; x < 4; ++x) buf[ay * 8 + (flipx ? 7 - x : x)] = safv[y * 4 + x];
			for (x = 0; x < 4; ++x) buf[p22 + x] = s22[y * 4 + x];
			for (x = 0; x < 8; ++x) buf[p23 + x] = s32[y * 8 + x];
		}
	} }
}

static uint8_t render_u8(int16_t px, int bpp) {  /* j40.h:7950-7951 */
	int32_t maxpixel = (1 << bpp) - 1, p = px < 0 ? 0 : px > maxpixel ? maxpixel : px;
	return (uint8_t) ((p * 255 + (1 << (bpp - 1))) / maxpixel);
}

/* the float -> int16 conversion of j40.h:7235 as the reference's x86-64 build performs it */
static int16_t to_i16(float t) {
	int32_t i;
	if (!(t > -2147483904.0f && t < 2147483648.0f)) i = (int32_t) 0x80000000u; else i = (int32_t) t;
	return (int16_t) (uint16_t) (uint32_t) i;
}

/* j40__dequant_hf (j40.h:7053) + j40__combine_vardct_from_lf_group (j40.h:7099) + render (j40.h:7910) */
/* xyb_out (optional): the samples as the inverse transforms leave them, before the colour conversion, into three frame-wide planes of
 * width * height floats (X, Y, B) -- what the restoration filters work on. */
static void colour_samples(const j40hip_vardct_view *v, float *const samples[3], int32_t ggw, int32_t ggh, int32_t left, int32_t top, uint8_t *rgba);
static void combine_lf_group(const j40hip_vardct_view *v, const j40hip_lf_group_view *gg, float *const coeffs[3], uint8_t *rgba, float *xyb_out) {
	static const float QM_SCALE[8] = {1.5625f, 1.25f, 1.0f, 0.8f, 0.64f, 0.512f, 0.4096f, 0.32768f};
	int32_t ggw8 = gg->width8, ggh8 = gg->height8, ggw = gg->width, ggh = gg->height, x8, y8, x, y, i, c;
	float x_qm = QM_SCALE[v->x_qm_scale], b_qm = QM_SCALE[v->b_qm_scale];
	float kx_lf = v->base_corr_x + (float) v->x_factor_lf * v->inv_colour_factor, kb_lf = v->base_corr_b + (float) v->b_factor_lf * v->inv_colour_factor;
	float *samples[3], *scratch = (float *) malloc(sizeof(float) * 2 * 65536), *scratch2 = scratch + 65536, cbrt_bias[3], itscale = 255.0f / v->intensity_target;
	for (c = 0; c < 3; ++c) samples[c] = (float *) malloc(sizeof(float) * (size_t) (ggw * ggh));
	/* dequantisation over whole blocks, LLF slots included (they are overwritten below) */
	for (y8 = 0; y8 < ggh8; ++y8) for (x8 = 0; x8 < ggw8; ++x8) {
		int32_t voff = gg->blocks[y8 * ggw8 + x8], dctsel = voff >> 20, size;
		float mult[3];
		const float *dq;
		if (dctsel < 2) continue;
		voff &= 0xfffff; dctsel -= 2;
		size = 1 << (DCTSEL[dctsel][0] + DCTSEL[dctsel][1]);
		mult[1] = 65536.0f / (float) v->global_scale * gg->hfmul_inv[voff];
		mult[0] = mult[1] * x_qm; mult[2] = mult[1] * b_qm;
		dq = v->dq_matrix[DCTSEL[dctsel][2]];
		for (c = 0; c < 3; ++c) {
			float *co = coeffs[c] + (gg->coeffoff_qfidx[voff] & ~15);
			for (i = 0; i < size; ++i) {
				if (-1.0f <= co[i] && co[i] <= 1.0f) co[i] *= v->quant_bias[c]; else co[i] -= v->quant_bias_num / co[i];
				co[i] *= mult[c] / dq[i * 3 + c];
			}
		}
	}
	for (y8 = 0; y8 < ggh8; ++y8) for (x8 = 0; x8 < ggw8; ++x8) {
		int32_t voff = gg->blocks[y8 * ggw8 + x8], dctsel = voff >> 20, log_rows, log_columns, size, effvw, effvh, vw8, vh8, coeffoff;
		float kx_hf, kb_hf;
		if (dctsel < 2) continue;
		dctsel -= 2; voff &= 0xfffff;
		log_rows = DCTSEL[dctsel][0]; log_columns = DCTSEL[dctsel][1]; size = 1 << (log_rows + log_columns);
		coeffoff = gg->coeffoff_qfidx[voff] & ~15;
		kx_hf = v->base_corr_x + v->inv_colour_factor * (float) gg->xfromy[(y8 / 8) * gg->width64 + x8 / 8];
		kb_hf = v->base_corr_b + v->inv_colour_factor * (float) gg->bfromy[(y8 / 8) * gg->width64 + x8 / 8];
		effvh = ggh - y8 * 8 < (1 << log_rows) ? ggh - y8 * 8 : 1 << log_rows;
		effvw = ggw - x8 * 8 < (1 << log_columns) ? ggw - x8 * 8 : 1 << log_columns;
		vh8 = 1 << ((log_rows < log_columns ? log_rows : log_columns) - 3); vw8 = 1 << ((log_rows > log_columns ? log_rows : log_columns) - 3);
		for (c = 0; c < 3; ++c) {
			const float *cx = coeffs[c] + coeffoff, *cy = coeffs[1] + coeffoff, *lc = gg->llfcoeffs[c] + (coeffoff >> 6), *ly = gg->llfcoeffs[1] + (coeffoff >> 6);
			float k_hf = c == 0 ? kx_hf : kb_hf, k_lf = c == 0 ? kx_lf : kb_lf;
			if (c == 1) { for (i = 0; i < size; ++i) scratch[i] = cy[i]; for (y = 0; y < vh8; ++y) for (x = 0; x < vw8; ++x) scratch[y * vw8 * 8 + x] = ly[y * vw8 + x]; }
			else { for (i = 0; i < size; ++i) scratch[i] = cx[i] + cy[i] * k_hf; for (y = 0; y < vh8; ++y) for (x = 0; x < vw8; ++x) scratch[y * vw8 * 8 + x] = lc[y * vw8 + x] + ly[y * vw8 + x] * k_lf; }
			if ((dctsel >= 1 && dctsel <= 3) || (dctsel >= 12 && dctsel <= 17)) inverse_8x8_special(dctsel, scratch);
			else idct2d(scratch, scratch2, log_rows, log_columns);
			for (y = 0; y < effvh; ++y) for (x = 0; x < effvw; ++x) samples[c][(y8 * 8 + y) * ggw + (x8 * 8 + x)] = scratch[y << log_columns | x];
		}
	}
	if (xyb_out) for (c = 0; c < 3; ++c) for (y = 0; y < ggh; ++y)
		memcpy(xyb_out + (size_t) c * (size_t) v->width * (size_t) v->height + (size_t) (gg->top + y) * (size_t) v->width + (size_t) gg->left, samples[c] + (size_t) y * (size_t) ggw, sizeof(float) * (size_t) ggw);
	if (rgba) colour_samples(v, samples, ggw, ggh, gg->left, gg->top, rgba);
	for (c = 0; c < 3; ++c) free(samples[c]);
	free(scratch);
	(void) cbrt_bias; (void) itscale;
}

/* XYB -> linear -> sRGB -> u8 (j40.h:7204-7240) + render (j40.h:7910): samples[c] = ggw * ggh floats placed at (left, top) of the frame */
static void colour_samples(const j40hip_vardct_view *v, float *const samples[3], int32_t ggw, int32_t ggh, int32_t left, int32_t top, uint8_t *rgba) {
	float cbrt_bias[3], itscale = 255.0f / v->intensity_target;
	int32_t x, y, c;
	for (c = 0; c < 3; ++c) cbrt_bias[c] = cbrtf(v->opsin_bias[c]);
	for (y = 0; y < ggh; ++y) for (x = 0; x < ggw; ++x) {
		int32_t pos = y * ggw + x;
		float p[3], s[3];
		uint8_t *out = rgba + ((size_t) (top + y) * (size_t) v->width + (size_t) (left + x)) * 4;
		p[0] = samples[1][pos] + samples[0][pos]; p[1] = samples[1][pos] - samples[0][pos]; p[2] = samples[2][pos];
		for (c = 0; c < 3; ++c) { float pp = p[c] - cbrt_bias[c]; s[c] = (pp * pp * pp + v->opsin_bias[c]) * itscale; }
		for (c = 0; c < 3; ++c) {
			float val = s[0] * v->opsin_inv_mat[c * 3] + s[1] * v->opsin_inv_mat[c * 3 + 1] + s[2] * v->opsin_inv_mat[c * 3 + 2];
			val = (val <= 0.0031308f ? 12.92f * val : 1.055f * powf(val, 1.0f / 2.4f) - 0.055f);
			out[c] = render_u8(to_i16((float) ((1 << v->bpp) - 1) * val + 0.5f), v->bpp);
		}
		out[3] = 255;
	}
}

/* the colour conversion alone, on three frame-wide planes of width * height floats (the restoration filters' output) */
ORACLE_API void oracle_xyb_to_rgba(const j40hip_vardct_view *v, const float *xyb, uint8_t *rgba) {
	float *planes[3];
	int c;
	for (c = 0; c < 3; ++c) planes[c] = (float *) xyb + (size_t) c * (size_t) v->width * (size_t) v->height;
	colour_samples(v, planes, v->width, v->height, 0, 0, rgba);
}

/* Decodes a VarDCT frame described by `v` into tightly packed RGBA. coeffs_out (optional): per LF
 * group and channel the quantised coefficients as the reference holds them before dequantisation,
 * concatenated [lf group][channel][width8 * height8 * 64]. Returns 0 or the first error. */
ORACLE_API uint32_t oracle_decode_vardct_xyb(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out, float *xyb_out);
ORACLE_API uint32_t oracle_decode_vardct(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out) { return oracle_decode_vardct_xyb(v, rgba, coeffs_out, NULL); }
/* ... xyb_out (optional): three planes of width * height floats, the samples before the colour conversion */
ORACLE_API uint32_t oracle_decode_vardct_xyb(const j40hip_vardct_view *v, uint8_t *rgba, float *coeffs_out, float *xyb_out) {
	float ***coeffs = (float ***) calloc((size_t) v->num_lf_groups, sizeof(float **));
	ocode *codes = (ocode *) calloc((size_t) v->num_passes, sizeof(ocode));
	uint32_t err = 0;
	int32_t g, c, pass;
	size_t off = 0;
	init_hs();
	for (g = 0; g < v->num_lf_groups; ++g) {
		coeffs[g] = (float **) calloc(3, sizeof(float *));
		for (c = 0; c < 3; ++c) coeffs[g][c] = (float *) calloc((size_t) (v->lf_groups[g].width8 * v->lf_groups[g].height8) * 64, sizeof(float));
	}
	for (pass = 0; pass < v->num_passes; ++pass) ocode_init(&codes[pass], &v->coeff_specs[pass]);
	for (pass = 0; pass < v->num_passes && !err; ++pass) for (g = 0; g < v->num_groups && !err; ++g) {
		const j40hip_section_view *sec = &v->sections[pass * v->num_groups + g];
		err = hf_coeffs(v, pass, sec, &codes[pass], coeffs[sec->ggidx]);
	}
	if (coeffs_out) for (g = 0; g < v->num_lf_groups; ++g) for (c = 0; c < 3; ++c) {
		size_t n = (size_t) (v->lf_groups[g].width8 * v->lf_groups[g].height8) * 64;
		memcpy(coeffs_out + off, coeffs[g][c], sizeof(float) * n); off += n;
	}
	if (!err && (rgba || xyb_out)) for (g = 0; g < v->num_lf_groups; ++g) combine_lf_group(v, &v->lf_groups[g], coeffs[g], rgba, xyb_out);
	for (pass = 0; pass < v->num_passes; ++pass) ocode_free(&codes[pass]);
	for (g = 0; g < v->num_lf_groups; ++g) { for (c = 0; c < 3; ++c) free(coeffs[g][c]); free(coeffs[g]); }
	free(coeffs); free(codes);
	return err;
}

/* ---------------------------------------------------------------------------------------------- */
/* Modular (j40.h:3965-4240, 4318-4490, 7910-7962)                                                 */

typedef struct { int16_t *px; int32_t w, h, meta; } oplane;
typedef struct { int32_t w, n, nw, ne, nn, nee, ww, nww; } oneigh;
typedef struct { int on; int32_t width, p1, p2, p3[5], w[4]; int32_t (*errors)[5]; int32_t pred[5], trueerrw, trueerrn, trueerrnw, trueerrne; } owp;

static int32_t iabs32(int32_t v) { return v < 0 ? -v : v; }
static int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }
static int32_t imax32(int32_t a, int32_t b) { return a > b ? a : b; }
static int32_t gradient(int32_t w, int32_t n, int32_t nw) { int32_t lo = imin32(w, n), hi = imax32(w, n); return imin32(imax32(lo, w + n - nw), hi); }
static int floor_lg(uint32_t x) { int n = 0; while (x >>= 1) ++n; return n; }
static int32_t div24(int32_t i) { return (int32_t) (((int64_t) 1 << 24) / (i + 1)); }

static oneigh neighbours(const int16_t *px, int32_t stride, int32_t width, int32_t x, int32_t y) {  /* j40.h:3965 */
	oneigh p;
	p.w = x > 0 ? px[x - 1] : y > 0 ? px[x - stride] : 0;
	p.n = y > 0 ? px[x - stride] : p.w;
	p.nw = x > 0 && y > 0 ? px[(x - 1) - stride] : p.w;
	p.ne = x + 1 < width && y > 0 ? px[(x + 1) - stride] : p.n;
	p.nn = y > 1 ? px[x - 2 * stride] : p.n;
	p.nee = x + 2 < width && y > 0 ? px[(x + 2) - stride] : p.ne;
	p.ww = x > 1 ? px[x - 2] : p.w;
	p.nww = x > 1 && y > 0 ? px[(x - 2) - stride] : p.ww;
	return p;
}

static void wp_reset(owp *s) { int i; if (s->on) memset(s->errors, 0, sizeof(int32_t[5]) * (size_t) s->width * 2); for (i = 0; i < 5; ++i) s->pred[i] = 0; s->trueerrw = s->trueerrn = s->trueerrnw = s->trueerrne = 0; }

static void wp_before(owp *s, int32_t x, int32_t y, const oneigh *p) {  /* j40.h:4011 */
	static const int32_t ZERO[5] = {0, 0, 0, 0, 0};
	int32_t (*err)[5], (*nerr)[5], w[4], wsum = 0, sum = 0, logw, i;
	const int32_t *errw, *errn, *errnw, *errne, *errww, *errw2;
	if (!s->on) return;
	err = s->errors + ((y & 1) ? s->width : 0); nerr = s->errors + ((y & 1) ? 0 : s->width);
	errw = x > 0 ? err[x - 1] : ZERO; errn = y > 0 ? nerr[x] : ZERO;
	errnw = x > 0 && y > 0 ? nerr[x - 1] : errn; errne = x + 1 < s->width && y > 0 ? nerr[x + 1] : errn;
	errww = x > 1 ? err[x - 2] : ZERO; errw2 = x + 1 < s->width ? ZERO : errw;
	s->trueerrw = x > 0 ? err[x - 1][4] : 0; s->trueerrn = y > 0 ? nerr[x][4] : 0;
	s->trueerrnw = x > 0 && y > 0 ? nerr[x - 1][4] : s->trueerrn; s->trueerrne = x + 1 < s->width && y > 0 ? nerr[x + 1][4] : s->trueerrn;
	s->pred[0] = (p->w + p->ne - p->n) * 8;
	s->pred[1] = p->n * 8 - (((s->trueerrw + s->trueerrn + s->trueerrne) * s->p1) >> 5);
	s->pred[2] = p->w * 8 - (((s->trueerrw + s->trueerrn + s->trueerrnw) * s->p2) >> 5);
	s->pred[3] = p->n * 8 - ((s->trueerrnw * s->p3[0] + s->trueerrn * s->p3[1] + s->trueerrne * s->p3[2] + (p->nn - p->n) * 8 * s->p3[3] + (p->nw - p->w) * 8 * s->p3[4]) >> 5);
	for (i = 0; i < 4; ++i) {
		int32_t errsum = errn[i] + errw[i] + errnw[i] + errww[i] + errne[i] + errw2[i], shift = imax32(floor_lg((uint32_t) errsum + 1) - 5, 0);
		w[i] = (int32_t) (4 + ((int64_t) s->w[i] * div24(errsum >> shift) >> shift));
	}
	logw = floor_lg((uint32_t) (w[0] + w[1] + w[2] + w[3])) - 4;
	for (i = 0; i < 4; ++i) { w[i] >>= logw; wsum += w[i]; sum += s->pred[i] * w[i]; }
	s->pred[4] = (int32_t) (((int64_t) sum + (wsum >> 1) - 1) * div24(wsum - 1) >> 24);
	if (((s->trueerrn ^ s->trueerrw) | (s->trueerrn ^ s->trueerrnw)) <= 0) {
		int32_t lo = imin32(p->w, imin32(p->n, p->ne)) * 8, hi = imax32(p->w, imax32(p->n, p->ne)) * 8;
		s->pred[4] = imin32(imax32(lo, s->pred[4]), hi);
	}
}
static void wp_after(owp *s, int32_t x, int32_t y, int32_t val) {
	int32_t *e, i;
	if (!s->on) return;
	e = s->errors[((y & 1) ? s->width : 0) + x];
	for (i = 0; i < 4; ++i) e[i] = (iabs32(s->pred[i] - val * 8) + 3) >> 3;
	e[4] = s->pred[4] - val * 8;
}

static int32_t predict(int pred, const owp *wp, const oneigh *p, uint32_t *err) {  /* j40.h:4080 */
	switch (pred) {
	case 0: return 0; case 1: return p->w; case 2: return p->n; case 3: return (p->w + p->n) / 2;
	case 4: return iabs32(p->n - p->nw) < iabs32(p->w - p->nw) ? p->w : p->n;
	case 5: return gradient(p->w, p->n, p->nw);
	case 6: return (wp->pred[4] + 3) >> 3;
	case 7: return p->ne; case 8: return p->nw; case 9: return p->ww;
	case 10: return (p->w + p->nw) / 2; case 11: return (p->n + p->nw) / 2; case 12: return (p->n + p->ne) / 2;
	case 13: return (6 * p->n - 2 * p->nn + 7 * p->w + p->ww + p->nee + 3 * p->ne + 8) / 16;
	default: if (!*err) *err = E4('p', 'r', 'e', 'd'); return 0;
	}
}

static void set_wp(owp *wp, const int8_t *params) { int i; wp->p1 = params[0]; wp->p2 = params[1]; for (i = 0; i < 5; ++i) wp->p3[i] = params[2 + i]; for (i = 0; i < 4; ++i) wp->w[i] = params[7 + i]; }

/* where channel `cidx` of a section lives: its plane and the rectangle of it the section codes. Frames whose channels differ in
 * size (after a Squeeze) list explicit rectangles; otherwise the section's rectangle (the whole plane for meta channels) */
typedef struct { oplane *pl; int32_t gx, gy, gw, gh, shifts; } ochan;
static ochan section_channel(const j40hip_modular_view *v, const j40hip_modular_section_view *sec, oplane *planes, int32_t cidx) {
	ochan c;
	if (sec->chan_off >= 0) {
		const int32_t *r = v->chan_rects + 6 * (sec->chan_off + cidx);
		c.pl = &planes[r[0]]; c.gx = r[1]; c.gy = r[2]; c.gw = r[3]; c.gh = r[4]; c.shifts = r[5];
		return c;
	}
	c.pl = &planes[sec->first_channel + cidx]; c.shifts = 0;
	c.gx = c.pl->meta ? 0 : sec->gx; c.gy = c.pl->meta ? 0 : sec->gy; c.gw = c.pl->meta ? c.pl->w : sec->gw; c.gh = c.pl->meta ? c.pl->h : sec->gh;
	return c;
}

/* one section: the listed channels of the global image, restricted to the section's rectangle
 * (whole plane for meta channels) -- j40__modular_channel16, j40.h:4127 */
static uint32_t modular_section(const j40hip_modular_view *v, const j40hip_modular_section_view *sec, oplane *planes, ocode *code) {
	const j40hip_tree_node *tree = v->tree + sec->tree_off;   /* the global tree or the section's own (j40.h:3740-3746) */
	int32_t uses_wp = 0, ti;
	obits b;
	owp wp;
	uint32_t err = 0;
	int32_t cidx, dist_mult = 0, k;
	for (ti = 0; ti < sec->tree_nodes; ++ti) if (tree[ti].prop == 15 || tree[ti].prop == -1 - 6) uses_wp = 1;
	obits_init(&b, v->codestream, sec->byte_off, sec->size, sec->bit_off);
	ocode_restart(code);
	/* j40.h:3840-3844: the widest non-meta channel of the Modular image the stream belongs to -- for LfGlobal's section that is the
	 * frame-wide image, including channels the section itself does not code (dist_mult_p1); for a pass group its sub-image */
	if (sec->dist_mult_p1) dist_mult = sec->dist_mult_p1 - 1;
	else for (cidx = 0; cidx < sec->num_channels; ++cidx) {
		ochan oc = section_channel(v, sec, planes, cidx);
		if (!oc.pl->meta) dist_mult = imax32(dist_mult, oc.gw);
	}
	dist_mult = imin32(dist_mult, 1 << 21);
	memset(&wp, 0, sizeof wp);
	set_wp(&wp, sec->wp);
	for (cidx = 0; cidx < sec->num_channels && !b.err && !err; ++cidx) {
		ochan oc = section_channel(v, sec, planes, cidx);
		oplane *pl = oc.pl;
		int32_t gx = oc.gx, gy = oc.gy, gw = oc.gw, gh = oc.gh, x, y;
		if (gw <= 0 || gh <= 0) continue;
		wp.on = uses_wp; wp.width = gw;
		wp.errors = uses_wp ? (int32_t (*)[5]) calloc((size_t) gw * 2, sizeof(int32_t[5])) : NULL;
		wp_reset(&wp);
		for (y = 0; y < gh && !b.err && !err; ++y) {
			int16_t *row = pl->px + (size_t) (gy + y) * (size_t) pl->w + (size_t) gx;
			for (x = 0; x < gw; ++x) {
				const j40hip_tree_node *n = tree;
				oneigh p = neighbours(row, pl->w, gw, x, y);
				int32_t val;
				wp_before(&wp, x, y, &p);
				while (n->prop >= 0) {
					switch (n->prop) {
					case 0: val = cidx; break; case 1: val = sec->sidx; break; case 2: val = y; break; case 3: val = x; break;
					case 4: val = iabs32(p.n); break; case 5: val = iabs32(p.w); break; case 6: val = p.n; break; case 7: val = p.w; break;
					case 8: val = x > 0 ? p.w - (p.ww + p.nw - p.nww) : p.w; break;
					case 9: val = p.w + p.n - p.nw; break; case 10: val = p.w - p.nw; break; case 11: val = p.nw - p.n; break;
					case 12: val = p.n - p.ne; break; case 13: val = p.n - p.nn; break; case 14: val = p.w - p.ww; break;
					case 15:
						val = wp.trueerrw;
						if (iabs32(val) < iabs32(wp.trueerrn)) val = wp.trueerrn;
						if (iabs32(val) < iabs32(wp.trueerrnw)) val = wp.trueerrnw;
						if (iabs32(val) < iabs32(wp.trueerrne)) val = wp.trueerrne;
						break;
					default: {
						int32_t r = (n->prop - 16) / 4, rgx = 0, rgy = 0;
						const oplane *rc = NULL;
						const int16_t *rrow;
						for (k = cidx - 1; k >= 0; --k) {  /* earlier channels of equal geometry, nearest first */
							ochan oc2 = section_channel(v, sec, planes, k);
							const oplane *cand = oc2.pl;
							if (cand->meta != pl->meta || oc2.gw != gw || oc2.gh != gh || oc2.shifts != oc.shifts) continue;
							if (r-- == 0) { rc = cand; rgx = oc2.gx; rgy = oc2.gy; break; }
						}
						if (!rc) { err = E4('t', 'r', 'e', 'c'); val = 0; break; }
						rrow = rc->px + (size_t) (rgy + y) * (size_t) rc->w + (size_t) rgx;
						val = rrow[x];
						if (n->prop & 2) {
							int32_t rw = x > 0 ? rrow[x - 1] : 0, rn = y > 0 ? rrow[x - rc->w] : rw, rnw = x > 0 && y > 0 ? rrow[x - 1 - rc->w] : rw;
							val -= gradient(rw, rn, rnw);
						}
						if (n->prop & 1) val = iabs32(val);
					} }
					if (err) break;
					n += val > n->value ? n->a : n->b;
				}
				if (err) break;
				val = ocode_symbol(&b, code, n->value, dist_mult);
				val = unpack_signed(val) * n->b + n->a;
				val += predict(-1 - n->prop, &wp, &p, &err);
				if (val < -32768 || val > 32767) { err = E4('p', 'o', 'v', 'f'); break; }
				row[x] = (int16_t) val;
				wp_after(&wp, x, y, val);
				if (b.err) break;
			}
		}
		free(wp.errors); wp.errors = NULL;
	}
	if (!b.err && !err) ocode_finish(&b, code);
	if (!b.err && !err && v->check_section_end) obits_finish(&b, v->codestream, v->single_declared_end);
	return b.err ? b.err : err;
}

static const int16_t PALETTE_DELTAS[72][3] = {  /* spec table; entry 2k = triple k, 2k + 1 = its negation (cf. j40.h:4275) */
	{0, 0, 0}, {4, 4, 4}, {11, 0, 0}, {0, 0, -13}, {0, -12, 0}, {-10, -10, -10}, {-18, -18, -18}, {-27, -27, -27}, {-18, -18, 0}, {0, 0, -32}, {-32, 0, 0}, {-37, -37, -37},
	{0, -32, -32}, {24, 24, 45}, {50, 50, 50}, {-45, -24, -24}, {-24, -45, -45}, {0, -24, -24}, {-34, -34, 0}, {-24, 0, -24}, {-45, -45, -24}, {64, 64, 64}, {-32, 0, -32}, {0, -32, 0},
	{-32, 0, 32}, {-24, -45, -24}, {45, 24, 45}, {24, -24, -45}, {-45, -24, 24}, {80, 80, 80}, {64, 0, 0}, {0, 0, -64}, {0, -64, -64}, {-24, -24, 45}, {96, 96, 96}, {64, 64, 0},
	{45, -24, -24}, {34, -34, 0}, {112, 112, 112}, {24, -45, -45}, {45, 45, -24}, {0, -32, 32}, {24, -24, 45}, {0, 96, 96}, {45, -24, 24}, {24, -45, -24}, {-24, -45, 24}, {0, -64, 0},
	{96, 0, 0}, {128, 128, 128}, {64, 0, 64}, {144, 144, 144}, {96, 96, 0}, {-36, -36, 36}, {45, -24, -45}, {45, -45, -24}, {0, 0, -96}, {0, 128, 128}, {0, 96, 0}, {45, 24, -45},
	{-128, 0, 0}, {24, -45, 24}, {-45, 24, -45}, {64, 0, -64}, {64, -64, -64}, {96, 0, 96}, {45, -45, 24}, {24, 45, -45}, {64, 64, -64}, {128, 128, 0}, {0, 0, -128}, {-24, 45, -45}};

/* one pixel of the inverse RCT, types 0..6 (j40.h:4341-4393); int16 results wrap like the reference's */
static void rct_pixel(int32_t type7, int16_t *q0, int16_t *q1, int16_t *q2) {
	int16_t a = *q0, b = *q1, d = *q2;
	switch (type7) {
	case 0: break;
	case 1: *q2 = (int16_t) (d + a); break;
	case 2: *q2 = (int16_t) (b + a); break;
	case 3: *q1 = (int16_t) (b + a); *q2 = (int16_t) (d + a); break;
	case 4: *q1 = (int16_t) (b + (int16_t) (a / 2 + d / 2 + (a & d & 1))); break;
	case 5: *q1 = (int16_t) ((int32_t) b + a + (d >> 1)); *q2 = (int16_t) (d + a); break;
	default: { int32_t tmp = (int32_t) a - ((int32_t) d >> 1), r1 = (int32_t) d + tmp, r2 = tmp - ((int32_t) b >> 1); *q0 = (int16_t) (r2 + b); *q1 = (int16_t) r1; *q2 = (int16_t) r2; }
	}
}

/* undoes `trs` last to first on the image planes[0 .. *np) (j40__inverse_transform, j40.h:4506): the frame, or the sub-image of a
 * section with a palette of its own. wpb: the weighted predictor parameters of that image's header. */
/* the Squeeze transform's "tendency" (ISO 18181-1): from the previous output sample B, the current average a and the next average n;
 * non-zero on monotone runs only, clamped so that both samples of the pair stay between their neighbours. Division truncates. */
static int32_t osq_tendency(int32_t B, int32_t a, int32_t n) {
	int32_t diff = 0;
	if (B >= a && a >= n) {
		diff = (4 * B - 3 * n - a + 6) / 12;
		if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
		if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
	} else if (B <= a && a <= n) {
		diff = (4 * B - 3 * n - a - 6) / 12;
		if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
		if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
	}
	return diff;
}
/* known-answer hooks (tests/test_squeeze.py): the tendency, and one line of n_avg averages and n_res residuals joined */
ORACLE_API int32_t oracle_kat_squeeze_tendency(int32_t B, int32_t a, int32_t n) { return osq_tendency(B, a, n); }
ORACLE_API void oracle_kat_unsqueeze_line(const int16_t *avg, int32_t n_avg, const int16_t *res, int32_t n_res, int16_t *out) {
	int32_t k, left = 0;
	for (k = 0; k < n_res; ++k) {
		int32_t a = avg[k], next = k + 1 < n_avg ? avg[k + 1] : a, B = k > 0 ? left : a, diff = osq_tendency(B, a, next) + res[k], A = a + diff / 2;
		out[2 * k] = (int16_t) A; out[2 * k + 1] = (int16_t) (A - diff);
		left = (int16_t) (A - diff);
	}
	if (n_avg > n_res) out[2 * n_res] = avg[n_res];
}

static uint32_t undo_transforms(oplane *planes, int32_t *np, const j40hip_transform_view *trs, int32_t ntr, const int8_t *wpb, int32_t bpp) {
	static const uint8_t PERM[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
	uint32_t err = 0;
	int32_t t, i;
	size_t k;
	for (t = ntr - 1; t >= 0 && !err; --t) {
		const j40hip_transform_view *tr = &trs[t];
		if (tr->kind == 0) {  /* inverse RCT, j40.h:4318 */
			oplane ch[3];
			int16_t *p0, *p1, *p2;
			size_t n;
			for (i = 0; i < 3; ++i) ch[i] = planes[tr->begin_c + i];
			p0 = ch[0].px; p1 = ch[1].px; p2 = ch[2].px; n = (size_t) ch[0].w * (size_t) ch[0].h;
			for (k = 0; k < n; ++k) rct_pixel(tr->rct_type % 7, &p0[k], &p1[k], &p2[k]);
			for (i = 0; i < 3; ++i) planes[tr->begin_c + PERM[tr->rct_type / 7][i]] = ch[i];
		} else if (tr->kind == 1) {  /* inverse palette, j40.h:4402 */
			int32_t first = tr->begin_c + 1, last = tr->begin_c + tr->num_c, width = planes[first].w, height = planes[first].h, x, y, j;
			int use_pred = tr->nb_deltas > 0;
			owp wp;
			memmove(planes + last, planes + first, sizeof(oplane) * (size_t) ((*np) - first));
			(*np) += last - first;
			for (i = first; i < last; ++i) { planes[i].w = width; planes[i].h = height; planes[i].meta = 0; planes[i].px = (int16_t *) calloc((size_t) width * (size_t) height + 1, sizeof(int16_t)); }
			memset(&wp, 0, sizeof wp);
			set_wp(&wp, wpb);
			wp.on = use_pred && tr->d_pred == 6; wp.width = width;
			wp.errors = wp.on ? (int32_t (*)[5]) calloc((size_t) width * 2, sizeof(int32_t[5])) : NULL;
			for (i = 0; i < tr->num_c; ++i) {
				const int16_t *palp = tr->nb_colours > 0 ? planes[0].px + (size_t) i * (size_t) planes[0].w : NULL;
				oplane *dst = &planes[first + i], *idxc = &planes[last];
				wp_reset(&wp);
				for (y = 0; y < height; ++y) for (x = 0; x < width; ++x) {
					int16_t idx = idxc->px[(size_t) y * (size_t) width + (size_t) x], val;
					int is_delta = idx < tr->nb_deltas;
					if (idx < 0) {
						if (i < 3) { int32_t e; idx = (int16_t) (~idx % 143); e = idx + 1; val = PALETTE_DELTAS[e >> 1][i]; if (e & 1) val = (int16_t) -val; if (bpp > 8) val = (int16_t) (val << (imin32(bpp, 24) - 8)); }
						else val = 0;
					} else if (idx < tr->nb_colours) val = palp[idx];
					else {
						idx = (int16_t) (idx - tr->nb_colours);
						if (idx < 64) val = (int16_t) ((i < 3 ? idx >> (2 * i) : 0) * (((int32_t) 1 << bpp) - 1) / 4 + ((int32_t) 1 << imax32(0, bpp - 3)));
						else { val = (int16_t) (idx - 64); for (j = 0; j < i; ++j) val = (int16_t) (val / 5); val = (int16_t) ((val % 5) * ((1 << bpp) - 1) / 4); }
					}
					if (use_pred) {
						oneigh p = neighbours(dst->px + (size_t) y * (size_t) width, width, width, x, y);
						wp_before(&wp, x, y, &p);
						if (is_delta) val = (int16_t) (val + predict(tr->d_pred, &wp, &p, &err));
						wp_after(&wp, x, y, val);
					}
					dst->px[(size_t) y * (size_t) width + (size_t) x] = val;
				}
			}
			free(wp.errors);
			free(planes[0].px);
			memmove(planes, planes + 1, sizeof(oplane) * (size_t) --(*np));
		} else if (tr->kind == 2) {
			/* one inverse Squeeze step (ISO 18181-1; the reference stops at "TODO", j40.h:4518): every squeezed channel is joined with
			 * its residual channel along rows (horizontal) or columns, then the residual channels leave the list.
			 * out[2k] = A, out[2k + 1] = A - diff, diff = residual[k] + tendency(out[2k - 1], avg[k], avg[k + 1]), A = avg[k] + diff / 2 */
			int32_t end_c = tr->begin_c + tr->num_c, offset = tr->in_place ? end_c : *np - tr->num_c, c;
			for (c = tr->begin_c; c < end_c; ++c) {
				oplane *avg = &planes[c], *res = &planes[offset + c - tr->begin_c], out;
				int32_t line, nlines, n_avg, n_res, kk;
				out.meta = avg->meta;
				out.w = tr->horizontal ? avg->w + res->w : avg->w; out.h = tr->horizontal ? avg->h : avg->h + res->h;
				out.px = (int16_t *) calloc((size_t) imax32(out.w, 0) * (size_t) imax32(out.h, 0) + 1, sizeof(int16_t));
				nlines = tr->horizontal ? out.h : out.w; n_avg = tr->horizontal ? avg->w : avg->h; n_res = tr->horizontal ? res->w : res->h;
				for (line = 0; line < nlines; ++line) {
					/* element (line, k) of a plane: row `line`, column k for the horizontal step; row k, column `line` for the vertical one */
					#define SQ_AT(pl, k) ((pl)->px[tr->horizontal ? (size_t) line * (size_t) (pl)->w + (size_t) (k) : (size_t) (k) * (size_t) (pl)->w + (size_t) line])
					int32_t left = 0;
					for (kk = 0; kk < n_res; ++kk) {
						int32_t a = SQ_AT(avg, kk), next = kk + 1 < n_avg ? SQ_AT(avg, kk + 1) : a, B = kk > 0 ? left : a, diff, A;
						diff = osq_tendency(B, a, next) + SQ_AT(res, kk);
						A = a + diff / 2;
						SQ_AT(&out, 2 * kk) = (int16_t) A; SQ_AT(&out, 2 * kk + 1) = (int16_t) (A - diff);
						left = (int16_t) (A - diff);
					}
					if (n_avg > n_res) SQ_AT(&out, 2 * n_res) = SQ_AT(avg, n_res);
					#undef SQ_AT
				}
				free(avg->px);
				*avg = out;
			}
			for (c = 0; c < tr->num_c; ++c) free(planes[offset + c].px);
			memmove(planes + offset, planes + offset + tr->num_c, sizeof(oplane) * (size_t) (*np - offset - tr->num_c));
			*np -= tr->num_c;
		} else err = E4('T', 'O', 'D', 'O');
	}
	return err;
}

/* Decodes a Modular frame described by `v` into tightly packed RGBA; returns 0 or the first error */
ORACLE_API uint32_t oracle_decode_modular(const j40hip_modular_view *v, uint8_t *rgba) {
	static const uint8_t PERM[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
	oplane planes[320];
	int32_t nplanes = v->num_channels, c, s, t, i;
	ocode *codes;
	uint32_t err = 0;
	size_t k, npx = (size_t) v->width * (size_t) v->height;
	for (c = 0; c < nplanes; ++c) {
		planes[c].w = v->channel_w[c]; planes[c].h = v->channel_h[c]; planes[c].meta = v->channel_meta[c];
		planes[c].px = (int16_t *) calloc((size_t) imax32(planes[c].w, 0) * (size_t) imax32(planes[c].h, 0) + 1, sizeof(int16_t));
	}
	codes = (ocode *) calloc((size_t) v->num_codespecs, sizeof(ocode));
	for (i = 0; i < v->num_codespecs; ++i) ocode_init(&codes[i], &v->codespec[i]);
	for (s = 0; s < v->num_sections && !err; ++s) {
		const j40hip_modular_section_view *sec = &v->sections[s];
		if (sec->preset_status) { err = sec->preset_status; break; }
		if (sec->sub_off < 0) { err = modular_section(v, sec, planes, &codes[sec->spec_idx]); continue; }
		{   /* the section's own sub-image: decode, undo its transforms, paste */
			oplane sp[64];
			j40hip_modular_section_view whole = *sec;
			int32_t nsp = sec->num_channels, y;
			for (c = 0; c < nsp; ++c) {
				sp[c].w = v->sub_w[sec->sub_off + c]; sp[c].h = v->sub_h[sec->sub_off + c]; sp[c].meta = v->sub_meta[sec->sub_off + c];
				sp[c].px = (int16_t *) calloc((size_t) sp[c].w * (size_t) sp[c].h + 1, sizeof(int16_t));
			}
			whole.gx = whole.gy = 0; whole.first_channel = 0;
			err = modular_section(v, &whole, sp, &codes[sec->spec_idx]);
			if (!err && sec->sub_paste) {
				err = undo_transforms(sp, &nsp, v->sub_transforms + sec->sub_tr_off, sec->sub_tr_count, sec->wp, v->bpp);
				for (c = 0; c < nsp && !err; ++c) for (y = 0; y < sp[c].h; ++y)
					memcpy(planes[sec->first_channel + c].px + (size_t) (sec->gy + y) * (size_t) planes[sec->first_channel + c].w + (size_t) sec->gx, sp[c].px + (size_t) y * (size_t) sp[c].w, sizeof(int16_t) * (size_t) sp[c].w);
			}
			for (c = 0; c < nsp; ++c) free(sp[c].px);
		}
	}
	for (i = 0; i < v->num_codespecs; ++i) ocode_free(&codes[i]);
	free(codes);
	/* transforms of a group's own header act on the group's sub-image before it is pasted (j40.h:7030-7032):
	 * same thing as undoing them over the group's rectangle of the frame planes, last to first */
	for (s = 0; s < v->num_sections && !err; ++s) {
		const j40hip_modular_section_view *sec = &v->sections[s];
		for (t = sec->local_count - 1; t >= 0; --t) {
			int32_t begin = sec->first_channel + v->local_rct[2 * (sec->local_off + t)], type = v->local_rct[2 * (sec->local_off + t) + 1], x, y;
			for (y = 0; y < sec->gh; ++y) for (x = 0; x < sec->gw; ++x) {
				size_t at = (size_t) (sec->gy + y) * (size_t) planes[begin].w + (size_t) (sec->gx + x);
				int16_t q[3];
				for (i = 0; i < 3; ++i) q[i] = planes[begin + i].px[at];
				rct_pixel(type % 7, &q[0], &q[1], &q[2]);
				for (i = 0; i < 3; ++i) planes[begin + PERM[type / 7][i]].px[at] = q[i];
			}
		}
	}
	/* sections with a palette of their own: transforms of the sub-image, then the paste (j40.h:7030-7032) happened in the loop
	 * above; now the frame's transforms */
	if (!err) err = undo_transforms(planes, &nplanes, v->transforms, v->num_transforms, v->global_wp, v->bpp);
	if (!err && nplanes >= 3) for (k = 0; k < npx; ++k) {  /* j40__render_to_u8x4_rgba, j40.h:7910 */
		for (c = 0; c < 3; ++c) rgba[k * 4 + (size_t) c] = render_u8(planes[c].px[k], v->bpp);
		rgba[k * 4 + 3] = v->alpha_channel >= 0 ? render_u8(planes[v->alpha_channel].px[k], v->bpp) : 255;
	}
	for (c = 0; c < nplanes; ++c) free(planes[c].px);
	return err;
}

/* ---------------------------------------------------------------------------------------------- */
/* Restoration filters: Gaborish (j40.h:7271-7325) and the edge-preserving filter (j40.h:7338-7625), restated over
 * the whole picture (the reference notes that they apply to the entire image, j40.h:7268), out of place: every output sample is a
 * function of the step's INPUT planes (the reference works in place behind line buffers that hold the input rows, which is the
 * same thing). The reference defines these routines and never calls them; tests/test_oracle.py pins this restatement against the
 * routines themselves (oracle/ref_harness.c: ref_kat_gaborish, ref_kat_epf), bit for bit. What they compute is kept as it stands
 * there, also where it departs from ISO 18181-1: the filter taps are fetched at (x + k[1], y + k[0]) while the distances are taken
 * towards (x + k[0], y + k[1]) (j40.h:7338, 7490, 7551); the border weight applies where BOTH coordinates are at a block edge
 * (j40.h:7529); twelve-tap kernel with repeated entries (j40.h:7579); the sign of the weight slope (j40.h:7466). */

static int32_t omirror(int32_t c, int32_t size) {  /* j40.h:7327 */
	for (;;) { if (c < 0) c = -c - 1; else if (c >= size) c = size * 2 - 1 - c; else return c; }
}
static int osurely_nonzero(float x) { return isfinite(x) && fabs(x) >= 1e-8f; }  /* j40.h:625 */

/* three planes of w*h floats, in place; weights6 = gab.weights[c][j]. 0, "gab0" (j40.h:7289) or "!mem". */
ORACLE_API uint32_t oracle_gaborish(float *px, float *py, float *pb, int32_t w, int32_t h, const float *weights6) {
	float *planes[3], *in = (float *) malloc(sizeof(float) * (size_t) w * (size_t) h);
	int32_t c, x, y;
	planes[0] = px; planes[1] = py; planes[2] = pb;
	if (!in) return E4('!', 'm', 'e', 'm');
	for (c = 0; c < 3; ++c) {
		float w0 = 1.0f, w1 = weights6[c * 2], w2 = weights6[c * 2 + 1], wsum = w0 + w1 * 4 + w2 * 4;
		if (!osurely_nonzero(wsum)) { free(in); return E4('g', 'a', 'b', '0'); }
		w0 /= wsum; w1 /= wsum; w2 /= wsum;
		memcpy(in, planes[c], sizeof(float) * (size_t) w * (size_t) h);
		for (y = 0; y < h; ++y) {
			const float *n = in + (size_t) (y > 0 ? y - 1 : 0) * (size_t) w, *l = in + (size_t) y * (size_t) w, *s = in + (size_t) (y + 1 < h ? y + 1 : y) * (size_t) w;
			float *o = planes[c] + (size_t) y * (size_t) w;
			if (w < 2) continue;  /* (the reference reads index 1 of a one-sample row: not restated) */
			o[0] = n[0] * (w2 + w1) + n[1] * w2 + l[0] * (w1 + w0) + l[1] * w1 + s[0] * (w2 + w1) + s[1] * w2;
			for (x = 1; x < w - 1; ++x) o[x] = n[x - 1] * w2 + n[x] * w1 + n[x + 1] * w2 + l[x - 1] * w1 + l[x] * w0 + l[x + 1] * w1 + s[x - 1] * w2 + s[x] * w1 + s[x + 1] * w2;
			o[w - 1] = n[w - 2] * w2 + n[w - 1] * (w1 + w2) + l[w - 2] * w1 + l[w - 1] * (w0 + w1) + s[w - 2] * w2 + s[w - 1] * (w1 + w2);
		}
	}
	free(in);
	return 0;
}

/* |in(x, y) - in(x + dx, y + dy)| with both positions mirrored into the picture: one entry of j40__epf_distance's plane (j40.h:7338-7369) */
static float oepf_dist(const float *in, int32_t w, int32_t h, int32_t x, int32_t y, int32_t dx, int32_t dy) {
	return fabsf(in[(size_t) omirror(y, h) * (size_t) w + (size_t) omirror(x, w)] - in[(size_t) omirror(y + dy, h) * (size_t) w + (size_t) omirror(x + dx, w)]);
}

/* A tap of the weighted sum: the step's input at (x + k1, y + k0), mirrored (the reference's `lines[2 + k0][c][x + k1]`, j40.h:7551;
 * k0 = k1 = 0 is the centre sample, j40.h:7536).
 * quirk = 0: exactly that -- what the routine's steady-state loop sets out to do (j40.h:7505-7512), and what the HIP kernels compute by
 * default. quirk = 1: what j40__epf_step ACTUALLY reads, so that this file can be held against the routine bit for bit, all channels.
 * Its line buffer gives each channel three row slots (`cstride = stride * 3`, j40.h:7437) for FOUR buffered rows (j40.h:7482), so the
 * fourth row slot of channel c is the first of channel c + 1, and a channel's rows are copied in after its predecessor's
 * (j40.h:7499-7511): for c < 2 the row "y" read at y % 4 == 0 is channel c + 1's row y + 1, and the row "y - 1" read at y % 4 == 1 is
 * channel c + 1's row y. And the mirrored borders of the three rows buffered before the loop are written with `c * cstride` added to
 * row pointers that already include it (j40.h:7484-7488): right for channel 0; channel 1's land in channel 2's slots (computed from
 * rows not yet copied in), channel 2's past the end of the buffer -- so where rows 0 and 1 read picture row 0 from those first buffers,
 * channels 1 and 2 find border slots nobody wrote: 0.0f under the zero-filling allocator of oracle/ref_harness.c (REF_ZEROED_ALLOC),
 * heap corruption under malloc. */
static float oepf_tap(float *const in[3], int32_t w, int32_t h, int32_t x, int32_t y, int32_t k0, int32_t k1, int c, int quirk) {
	int32_t xx = x + k1, yy = y + k0;
	if (quirk) {
		if (c < 2 && ((k0 == 0 && (y & 3) == 0) || (k0 == -1 && (y & 3) == 1))) { ++c; ++yy; }   /* the slot holds the next channel's row */
		else if (c > 0 && (xx < 0 || xx >= w) && ((y == 0 && k0 <= 0) || (y == 1 && k0 < 0))) return 0.0f;   /* a border slot of the first buffers */
	}
	return in[c][(size_t) omirror(yy, h) * (size_t) w + (size_t) omirror(xx, w)];
}

/* one j40__epf_step (j40.h:7427-7576) */
static void oepf_step(float *planes[3], float *in[3], int32_t w, int32_t h, int32_t w8, const float *recip_sigmas, float sigma_scale, float border_sad_mul,
		const float *channel_scale, int32_t nk, const int32_t (*k)[2], int cross, int quirk) {
	int32_t x, y, c, i;
	float border_sigma_scale;
	sigma_scale *= 1.9330952441687859f;
	border_sigma_scale = sigma_scale * border_sad_mul;
	for (c = 0; c < 3; ++c) memcpy(in[c], planes[c], sizeof(float) * (size_t) w * (size_t) h);
	for (y = 0; y < h; ++y) for (x = 0; x < w; ++x) {
		float rs = recip_sigmas[(size_t) (y / 8) * (size_t) w8 + (size_t) (x / 8)], ism, sum_w = 1.0f, sum[3];
		if (rs < 0.0f) continue;  /* the whole cell keeps its samples (j40.h:7521-7524) */
		ism = rs * (((((x + 1) | (y + 1)) & 7) < 2) ? border_sigma_scale : sigma_scale);
		for (c = 0; c < 3; ++c) sum[c] = oepf_tap(in, w, h, x, y, 0, 0, c, quirk);
		for (i = 0; i < nk; ++i) {
			float dist = 0.0f, weight;
			for (c = 0; c < 3; ++c) {
				if (cross) dist += channel_scale[c] * (oepf_dist(in[c], w, h, x, y, k[i][0], k[i][1]) + oepf_dist(in[c], w, h, x - 1, y, k[i][0], k[i][1]) +
					oepf_dist(in[c], w, h, x, y - 1, k[i][0], k[i][1]) + oepf_dist(in[c], w, h, x, y + 1, k[i][0], k[i][1]) + oepf_dist(in[c], w, h, x + 1, y, k[i][0], k[i][1]));
				else dist += channel_scale[c] * oepf_dist(in[c], w, h, x, y, k[i][0], k[i][1]);
			}
			weight = 1.0f + dist * ism;
			weight = 0.0f > weight ? 0.0f : weight;
			sum_w += weight;
			for (c = 0; c < 3; ++c) sum[c] += oepf_tap(in, w, h, x, y, k[i][0], k[i][1], c, quirk) * weight;
		}
		for (c = 0; c < 3; ++c) planes[c][(size_t) y * (size_t) w + (size_t) x] = sum[c] / sum_w;
	}
}

static const int32_t OEPF_K12[12][2] = {{0, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {-1, 1}, {-1, 0}, {-1, 1}, {0, 2}};  /* j40.h:7579 */
static const int32_t OEPF_K4[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};

/* one step by itself (0, 1, 2 as j40__epf runs them, j40.h:7606-7616) on a given plane of reciprocal sigmas */
ORACLE_API uint32_t oracle_epf_step(float *px, float *py, float *pb, int32_t w, int32_t h, const float *recip_sigmas, int32_t step, const float *params15, int32_t quirk) {
	float *planes[3], *in[3];
	int c;
	planes[0] = px; planes[1] = py; planes[2] = pb;
	for (c = 0; c < 3; ++c) in[c] = (float *) malloc(sizeof(float) * (size_t) w * (size_t) h);
	if (in[0] && in[1] && in[2]) oepf_step(planes, in, w, h, (w + 7) / 8, recip_sigmas, step == 0 ? params15[12] : step == 1 ? 1.0f : params15[13], params15[14], params15 + 8,
		step == 0 ? 12 : 4, step == 0 ? OEPF_K12 : OEPF_K4, step != 2, quirk);
	for (c = 0; c < 3; ++c) free(in[c]);
	return in[0] && in[1] && in[2] ? 0 : E4('!', 'm', 'e', 'm');
}

/* three planes of w*h floats, in place. sharpness: int16[w8*h8] as decoded; hfmul_inv: float[w8*h8], the HfMul reciprocal of the
 * varblock covering each cell; params15 = sharp_lut[8], channel_scale[3], quant_mul, pass0_sigma_scale, pass2_sigma_scale,
 * border_sad_mul; sigma_out (optional): the plane of j40__epf_recip_sigmas (j40.h:7374-7425); quirk: see oepf_tap. 0, "epf0", "shrp" or "!mem". */
ORACLE_API uint32_t oracle_epf(float *px, float *py, float *pb, int32_t w, int32_t h, const int16_t *sharpness, const float *hfmul_inv, int32_t iters,
		const float *params15, float *sigma_out, int32_t quirk) {
	int32_t w8 = (w + 7) / 8, h8 = (h + 7) / 8, i, c;
	float lut[8], *rs, *planes[3], *in[3] = {NULL, NULL, NULL};
	uint16_t ub = 0;
	uint32_t err = 0;
	if (iters <= 0) return 0;
	planes[0] = px; planes[1] = py; planes[2] = pb;
	for (i = 0; i < 8; ++i) {
		float q = params15[11] * params15[i];
		if (!osurely_nonzero(q)) return E4('e', 'p', 'f', '0');
		lut[i] = 1.0f / q;
	}
	rs = (float *) malloc(sizeof(float) * (size_t) w8 * (size_t) h8);
	for (c = 0; c < 3; ++c) in[c] = (float *) malloc(sizeof(float) * (size_t) w * (size_t) h);
	if (!rs || !in[0] || !in[1] || !in[2]) { err = E4('!', 'm', 'e', 'm'); goto done; }
	for (i = 0; i < w8 * h8; ++i) { ub |= (uint16_t) sharpness[i]; rs[i] = lut[sharpness[i] & 7]; }
	if (!(ub < 8)) { err = E4('s', 'h', 'r', 'p'); goto done; }
	for (i = 0; i < w8 * h8; ++i) { rs[i] *= hfmul_inv[i]; if (rs[i] > 1.0f / 0.3f) rs[i] = -1.0f; }
	if (sigma_out) memcpy(sigma_out, rs, sizeof(float) * (size_t) w8 * (size_t) h8);
	if (iters >= 3) oepf_step(planes, in, w, h, w8, rs, params15[12], params15[14], params15 + 8, 12, OEPF_K12, 1, quirk);
	if (iters >= 1) oepf_step(planes, in, w, h, w8, rs, 1.0f, params15[14], params15 + 8, 4, OEPF_K4, 1, quirk);
	if (iters >= 2) oepf_step(planes, in, w, h, w8, rs, params15[13], params15[14], params15 + 8, 4, OEPF_K4, 0, quirk);
done:
	free(rs); for (c = 0; c < 3; ++c) free(in[c]);
	return err;
}
