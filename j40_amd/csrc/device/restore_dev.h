// j40_amd/csrc/device/restore_dev.h -- the restoration filters as device / host functions: Gaborish
// (j40__gaborish, j40.h:7271-7325) and the edge-preserving filter (j40__epf and its parts, j40.h:7338-7625) per output sample.
//
// The reference defines these routines and never calls them (its decode ignores the frame header's `gab` / `epf` fields,
// j40.h:5339-5366); they are still the only statement of the filters this image holds, so the arithmetic here is theirs, operation
// for operation (IEEE single precision, no contraction: the library is built with -ffp-contract=off), stated per sample over the
// WHOLE picture (the reference notes that the filters apply to the entire image, j40.h:7268) and out of place: a step's output is a
// function of its input planes, which is what the reference's in-place loops behind their line buffers come to. Kept as they stand
// there, also where they depart from ISO 18181-1: the taps of the weighted sum are fetched at (x + k[1], y + k[0]) while the
// distances are taken towards (x + k[0], y + k[1]) (j40.h:7338, 7490, 7551); the border weight applies where BOTH coordinates sit at
// a block edge (j40.h:7529); the twelve-tap kernel lists some taps more than once (j40.h:7579); the weight's slope is positive
// (j40.h:7466).
//
// EPF_AS_J40 (`quirk`): j40__epf_step's line buffer gives each channel three row slots for four buffered rows and sets up the first
// rows' mirrored borders at the wrong offsets for channels 1 and 2 (j40.h:7437, 7482-7488) -- the routine writes outside its buffer
// (it can only be run under an allocator that leaves slack: oracle/ref_harness.c, REF_ZEROED_ALLOC) and half of the rows of X and
// Y read the NEXT channel's samples. epf_tap() restates what is read in that case, so that the kernels can be held against the
// routine itself bit for bit on every channel (tests/test_restoration.py); the default is the filter the routine's steady-state
// loop sets out to compute -- every channel its own rows, mirrored at the picture's edges.
#pragma once
#include <stdint.h>
#ifndef J40_DEV
#ifdef __HIPCC__
#define J40_DEV __device__ __forceinline__
#define J40_DEVM __device__ __forceinline__
#else
#define J40_DEV static inline
#define J40_DEVM inline
#endif
#endif

namespace j40hip {

// what the filter kernels are given (host-built from the frame header's RestorationFilter bundle, j40.h:5088-5100)
struct RestoreParams {
	int32_t width, height, w8, h8;
	float gab_w[3][3];          // per channel: w0, w1, w2 already divided by their weighted sum (j40.h:7287-7290)
	float channel_scale[3];     // epf.channel_scale
	float sigma_scale[3];       // per step 0, 1, 2: pass0_sigma_scale / 1 / pass2_sigma_scale, times 1.9330952441687859f (j40.h:7466)
	float border_scale[3];      // ... times border_sad_mul (j40.h:7467)
	float inv_quant_sharp_lut[8];   // 1 / (quant_mul * sharp_lut[i]) (j40.h:7382-7386)
	int32_t quirk;              // EPF_AS_J40
};

J40_DEV int32_t restore_mirror(int32_t c, int32_t size) {   // j40.h:7327
	for (;;) { if (c < 0) c = -c - 1; else if (c >= size) c = size * 2 - 1 - c; else return c; }
}
J40_DEV float restore_fabs(float v) { return v < 0.0f ? -v : (v == 0.0f ? 0.0f : v); }   // fabsf: -0 -> +0 as well

// One Gaborish output sample of a plane of `w` (>= 2) x `h` samples. n / l / s: the rows above (row 0 for y = 0), at and below
// (the last row for y = h - 1) the sample; w0, w1, w2: centre, edge and corner weights, normalised (j40.h:7304-7318).
template <typename ROW>
J40_DEV float gaborish_sample(ROW n, ROW l, ROW s, int32_t x, int32_t w, float w0, float w1, float w2) {
	if (x == 0) return n[0] * (w2 + w1) + n[1] * w2 + l[0] * (w1 + w0) + l[1] * w1 + s[0] * (w2 + w1) + s[1] * w2;
	if (x == w - 1) return n[w - 2] * w2 + n[w - 1] * (w1 + w2) + l[w - 2] * w1 + l[w - 1] * (w0 + w1) + s[w - 2] * w2 + s[w - 1] * (w1 + w2);
	return n[x - 1] * w2 + n[x] * w1 + n[x + 1] * w2 + l[x - 1] * w1 + l[x] * w0 + l[x + 1] * w1 + s[x - 1] * w2 + s[x] * w1 + s[x + 1] * w2;
}

// the reciprocal sigma of one 8x8 cell (j40__epf_recip_sigmas, j40.h:7392-7419): < 0 = the cell keeps its samples
J40_DEV float epf_recip_sigma(const RestoreParams &p, int32_t sharpness, float hfmul_inv) {
	float rs = p.inv_quant_sharp_lut[sharpness & 7];
	rs *= hfmul_inv;
	if (rs > 1.0f / 0.3f) rs = -1.0f;
	return rs;
}

// taps: (k0, k1) of the twelve-tap kernel of step 0 and the four-tap kernel of steps 1 and 2 (j40.h:7579-7583)
#define J40_EPF_K12 {{0, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 1}, {0, 2}, {-1, 1}, {-1, 0}, {-1, 1}, {0, 2}}
#define J40_EPF_K4 {{0, -1}, {-1, 0}, {1, 0}, {0, 1}}

// ACC(c, x, y): the step's input sample of channel c at (x, y) MIRRORED INTO THE PICTURE -- the coordinates may lie up to three
// samples outside it (an accessor over the planes mirrors them itself, EpfMirrored below; the kernels' LDS tiles were filled mirrored).
// |in(x, y) - in(x + dx, y + dy)|: an entry of j40__epf_distance's plane (j40.h:7338-7369)
template <typename ACC>
J40_DEV float epf_distance(const ACC &in, int32_t c, int32_t w, int32_t h, int32_t x, int32_t y, int32_t dx, int32_t dy) {
	(void) w; (void) h;
	return restore_fabs(in(c, x, y) - in(c, x + dx, y + dy));
}
// an accessor over planes that takes inside coordinates only, made into one that mirrors
template <typename INSIDE> struct EpfMirrored {
	INSIDE in; int32_t w, h;
	J40_DEVM float operator()(int32_t c, int32_t x, int32_t y) const { return in(c, restore_mirror(x, w), restore_mirror(y, h)); }
};

// a tap of the weighted sum (the reference's lines[2 + k0][c][x + k1], j40.h:7536, 7551); quirk: the header of this file
template <typename ACC>
J40_DEV float epf_tap(const ACC &in, int32_t c, int32_t w, int32_t h, int32_t x, int32_t y, int32_t k0, int32_t k1, int32_t quirk) {
	(void) h;
	const int32_t xx = x + k1;
	int32_t yy = y + k0;
	if (quirk) {
		if (c < 2 && ((k0 == 0 && (y & 3) == 0) || (k0 == -1 && (y & 3) == 1))) { ++c; ++yy; }   // the row slot holds the next channel's row
		else if (c > 0 && (xx < 0 || xx >= w) && ((y == 0 && k0 <= 0) || (y == 1 && k0 < 0))) return 0.0f;   // a border slot nobody wrote
	}
	return in(c, xx, yy);
}

// One output sample triple of step STEP (0: twelve taps, cross-shaped distances; 1: four taps, cross; 2: four taps, plain distances;
// j40.h:7606-7616) at (x, y), whose cell has the reciprocal sigma `rs` >= 0 (j40.h:7517-7567).
template <int STEP, typename ACC>
J40_DEV void epf_sample(const ACC &in, const RestoreParams &p, int32_t x, int32_t y, float rs, float out[3]) {
	constexpr int NK = STEP == 0 ? 12 : 4;
	const int32_t k12[12][2] = J40_EPF_K12, k4[4][2] = J40_EPF_K4;
	const int32_t w = p.width, h = p.height;
	const float ism = rs * (((((x + 1) | (y + 1)) & 7) < 2) ? p.border_scale[STEP] : p.sigma_scale[STEP]);
	float sum_w = 1.0f, sum[3];
	for (int c = 0; c < 3; ++c) sum[c] = epf_tap(in, c, w, h, x, y, 0, 0, p.quirk);
#pragma unroll
	for (int i = 0; i < NK; ++i) {
		const int32_t k0 = STEP == 0 ? k12[i][0] : k4[i][0], k1 = STEP == 0 ? k12[i][1] : k4[i][1];
		float dist = 0.0f;
		for (int c = 0; c < 3; ++c) {
			if (STEP != 2) dist += p.channel_scale[c] * (epf_distance(in, c, w, h, x, y, k0, k1) + epf_distance(in, c, w, h, x - 1, y, k0, k1) + epf_distance(in, c, w, h, x, y - 1, k0, k1) +
				epf_distance(in, c, w, h, x, y + 1, k0, k1) + epf_distance(in, c, w, h, x + 1, y, k0, k1));
			else dist += p.channel_scale[c] * epf_distance(in, c, w, h, x, y, k0, k1);
		}
		float weight = 1.0f + dist * ism;
		weight = 0.0f > weight ? 0.0f : weight;
		sum_w += weight;
		for (int c = 0; c < 3; ++c) sum[c] += epf_tap(in, c, w, h, x, y, k0, k1, p.quirk) * weight;
	}
	for (int c = 0; c < 3; ++c) out[c] = sum[c] / sum_w;
}

}  // namespace j40hip
