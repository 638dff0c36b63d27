// j40_amd/csrc/device/idct_dev.h -- inverse transforms and colour conversion as per-lane device
// functions: the variable-block inverse DCT family (Perera-Liu radix-2 butterflies), dequantisation,
// XYB -> sRGB and the u8 pack. (The 8x8 "special" transforms live in special8_dev.h.)
//
// Bit parity with the reference's float path needs its exact operation order and no FMA
// contraction (compiled with -ffp-contract=off; see the note at j40.h:5834). Each function cites the
// reference routine whose arithmetic it reproduces.
#pragma once
#include "plan.h"
#ifndef J40_DEV
#ifdef __HIPCC__
#define J40_DEV __device__ __forceinline__
#define J40_DEVM __device__ __forceinline__
#else
#define J40_DEV static inline
#define J40_DEVM inline
#endif
#endif
#ifndef __HIPCC__
#include <math.h>
#endif

namespace j40hip {

#define J40_SQRT2F 1.4142135623730951f

// 1-D inverse DCT of length N on a register-resident vector (j40__inverse_dct_core recursion,
// j40.h:5802-5841, with j40__dct2 / j40__inverse_dct4 as the tails); hs = half-secant table
template <int N> struct Idct1D {
	static J40_DEVM void run(float *x, const float *hs) {
		float a[N / 2], b[N / 2];
#pragma unroll
		for (int i = 0; i < N / 2; ++i) a[i] = x[2 * i];
		b[0] = J40_SQRT2F * x[1];
#pragma unroll
		for (int i = 1; i < N / 2; ++i) b[i] = x[2 * i - 1] + x[2 * i + 1];
		Idct1D<N / 2>::run(a, hs);
		Idct1D<N / 2>::run(b, hs);
#pragma unroll
		for (int i = 0; i < N / 2; ++i) {
			const float m = hs[N / 2 + i];
			const float t = b[i] * m;
			x[i] = a[i] + t;
			x[N - 1 - i] = a[i] - t;
		}
	}
};
template <> struct Idct1D<2> {
	static J40_DEVM void run(float *x, const float *) { const float p = x[0], q = x[1]; x[0] = p + q; x[1] = p - q; }
};
template <> struct Idct1D<1> { static J40_DEVM void run(float *, const float *) {} };

// ---- dequantisation (j40__dequant_hf, j40.h:7086-7094) ----
J40_DEV float dequant_coeff(float q, float quant_bias_c, float quant_bias_num, float mult_c, float dq) {
	if (-1.0f <= q && q <= 1.0f) q *= quant_bias_c; else q -= quant_bias_num / q;
	return q * (mult_c / dq);
}
// ---- XYB -> sRGB -> clamped u8 (j40.h:7208-7237, 7947-7953) ----

// float -> int16 the way the reference's x86 build does it: cvttss2si (truncate, "integer
// indefinite" 0x80000000 when out of int32 range or NaN), then keep the low 16 bits
J40_DEV int32_t f32_to_i16_x86(float t) {
	int32_t i;
	if (!(t > -2147483904.0f && t < 2147483648.0f)) i = (int32_t) 0x80000000u;
	else i = (int32_t) t;
	return (int32_t) (int16_t) (uint16_t) (uint32_t) i;
}

// x^(1.0f / 2.4f) rounded to float, for x > 0. The reference calls powf(v, 1.0f / 2.4f) (j40.h:7217); a
// correctly rounded powf is what glibc delivers up to a few results per 10^7, and the u8 output only
// changes when 255 * (1.055 p - 0.055) + 0.5 sits within that rounding of an integer. A general fp64
// pow() costs ~250 double-rate instructions per sample and made the pixel kernels compute bound, so:
//   12 * (1.0f / 2.4f) = 5 - 2^-23 exactly, hence r = x^(1/2.4f) solves r^12 = x^5 * x^(-2^-23).
//   Seed r0 from the hardware log2/exp2 (relative error ~1e-6), then one correction step in fp64:
//   E = x^5 * (1 + d) / r0^12 - 1 with d = expm1(-2^-23 ln x), r = r0 * (1 + E)^(1/12) (3rd-order series).
// Relative error ~1e-14 before the final rounding (tests/hostsim sweeps every float in [2^-9, 4) and samples
// of the rest against (float) pow((double) x, (double) (1.0f / 2.4f)): identical).
J40_DEV float pow_1_over_2p4(float xf) {
#ifdef __HIPCC__
	const float l2 = __builtin_amdgcn_logf(xf);                       // v_log_f32 (log2)
	const float r0f = __builtin_amdgcn_exp2f(l2 * 0.416666657f);      // v_exp_f32
	const float rho0f = __builtin_amdgcn_rcpf(r0f);
#else
	const float l2 = log2f(xf);
	const float r0f = exp2f(l2 * 0.416666657f);
	const float rho0f = 1.0f / r0f;
#endif
	const double x = (double) xf, r0 = (double) r0f, rho0 = (double) rho0f;
	const double x2 = x * x, x4 = x2 * x2, x5 = x4 * x;
	const double rho = rho0 * fma(-r0, rho0, 2.0);                    // 1 / r0 to ~1e-14
	const double rho2 = rho * rho, rho4 = rho2 * rho2, rho8 = rho4 * rho4, rho12 = rho8 * rho4;
	const double t = (double) l2 * -8.262958294867817e-08;           // -2^-23 * ln 2 * log2 x
	const double d = fma(t * 0.5, t, t);
	double q = x5 * rho12;
	q = fma(q, d, q);
	const double E = q - 1.0;
	// (1 + E)^(1/12) = 1 + E/12 - 11 E^2/288 + 253 E^3/10368 - ...
	const double poly = fma(fma(fma(253.0 / 10368.0, E, -11.0 / 288.0), E, 1.0 / 12.0), E, 1.0);
	return (float) (r0 * poly);
}

J40_DEV float srgb_transfer(float v) {
	if (v <= 0.0031308f) return 12.92f * v;
	return 1.055f * pow_1_over_2p4(v) - 0.055f;
}

// The 8-bit sample as a function of the linear value v is a non-decreasing step function (every stage --
// transfer curve, scale, + 0.5, truncation, clamp -- is monotone) as long as the int16 wrap-around of the
// reference's conversion stays out of reach, i.e. for -9 < v < 50000. So the sample is the number of thresholds
// thr[1..255] that are <= v, where thr[k] = smallest float with sample >= k (built on the host from the exact
// formula by bisection over the floats, tables.cpp). The floats sharing their top 16 bits form a bucket narrower
// than one step; a byte table behind the thresholds gives the sample at the bucket's low end and one comparison
// with the next threshold settles it: same bytes as the long way in a dozen instructions and no transcendental
// (the transfer function was ~40 % of the pixel kernels' issue slots). thr[0] = -inf, thr[256] = thr[257] = +inf.
J40_DEV int32_t srgb_u8_from_thresholds(float v, const J40_LDS float *thr) {
	int32_t bits;
#ifdef __HIPCC__
	bits = __float_as_int(v);
#else
	memcpy(&bits, &v, 4);
#endif
	int32_t b = bits >> 16;   // negative values land below the first bucket
	b = (b < SRGB_BUCKET_LO ? SRGB_BUCKET_LO : b > SRGB_BUCKET_HI ? SRGB_BUCKET_HI : b) - SRGB_BUCKET_LO;
	const int32_t k = ((const J40_LDS uint8_t *) (thr + SRGB_THRESHOLDS))[b];
	return k + (int32_t) (v >= thr[k + 1]);
}

// the long way for one linear value: transfer curve, conversion with the reference's int16 quirk, clamp, scaling to 8 bits
// (j40.h:7213-7240, 7925-7935). Kept out of line: the pixel kernels almost never need it (8-bit frames, values inside
// (-9, 50000)), and inlined it bloats their inner loops.
#ifdef __HIPCC__
__device__ __attribute__((noinline))
#else
static inline
#endif
int32_t srgb_sample_slow(float v, int32_t bpp) {
	const int32_t maxpixel = (1 << bpp) - 1, maxpixel2 = 1 << (bpp - 1);
	int32_t px = f32_to_i16_x86((float) maxpixel * srgb_transfer(v) + 0.5f);
	px = px < 0 ? 0 : px > maxpixel ? maxpixel : px;
	return (px * 255 + maxpixel2) / maxpixel;
}

// the frame constants of the colour conversion, copied out of DevFrame once per kernel so that they live in (scalar)
// registers instead of being re-read from memory after every pixel store
struct ColourConsts {
	float cbrt_opsin_bias[3], opsin_bias[3], itscale, m[9];
	int32_t bpp;
};
J40_DEV ColourConsts load_colour_consts(const DevFrame &f) {
	ColourConsts c;
	for (int i = 0; i < 3; ++i) { c.cbrt_opsin_bias[i] = f.cbrt_opsin_bias[i]; c.opsin_bias[i] = f.opsin_bias[i]; }
	for (int i = 0; i < 9; ++i) c.m[i] = f.opsin_inv_mat[i];
	c.itscale = f.itscale; c.bpp = f.bpp;
	return c;
}

// returns RGBA packed little-endian (R in the low byte), alpha = 255. thr: the threshold table (see above), used for 8-bit
// frames; other depths go the long way. (The choice is made on f.bpp, not on the pointer: a null test on an LDS pointer was
// folded to "always null" by the compiler and every pixel took the long way.) XYB -> linear RGB: j40.h:7208-7212.
J40_DEV uint32_t xyb_to_rgba8(float sx, float sy, float sb, const ColourConsts &f, const J40_LDS float *thr) {
	float p[3] = {sy + sx, sy - sx, sb};
	float s[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float pp = p[c] - f.cbrt_opsin_bias[c];
		s[c] = (pp * pp * pp + f.opsin_bias[c]) * f.itscale;
	}
	float v[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) v[c] = s[0] * f.m[c * 3] + s[1] * f.m[c * 3 + 1] + s[2] * f.m[c * 3 + 2];
	const float vmin = fminf(fminf(v[0], v[1]), v[2]), vmax = fmaxf(fmaxf(v[0], v[1]), v[2]);
	uint32_t out = 0xff000000u;
	if (f.bpp == 8 && vmin > -9.0f && vmax < 50000.0f && v[0] == v[0] && v[1] == v[1] && v[2] == v[2]) {   // (fmin / fmax drop NaNs)
#pragma unroll
		for (int c = 0; c < 3; ++c) out |= (uint32_t) srgb_u8_from_thresholds(v[c], thr) << (8 * c);
	} else {
		for (int c = 0; c < 3; ++c) out |= (uint32_t) srgb_sample_slow(v[c], f.bpp) << (8 * c);
	}
	return out;
}
J40_DEV uint32_t xyb_to_rgba8(float sx, float sy, float sb, const DevFrame &f, const J40_LDS float *thr) {
	return xyb_to_rgba8(sx, sy, sb, load_colour_consts(f), thr);
}

// host: the sample the long way, for one linear value of an 8-bit frame (table construction and tests)
J40_DEV int32_t srgb_u8_exact8(float v) {
	int32_t px = f32_to_i16_x86(255.0f * srgb_transfer(v) + 0.5f);
	return px < 0 ? 0 : px > 255 ? 255 : px;
}

} // namespace j40hip
