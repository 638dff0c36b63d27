// j40_amd/csrc/device/idct_dev.h -- inverse transforms and colour conversion as per-lane device
// functions: the variable-block inverse DCT family (Perera-Liu radix-2 butterflies), the ten 8x8
// "special" transforms, dequantisation, XYB -> sRGB and the u8 pack.
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

// strided variant working in memory (LDS or global), for the 8x8 specials: length N, elements at
// v[i * stride]
template <int N> J40_DEV void idct1d_strided(float *v, int stride, const float *hs) {
	float x[N];
#pragma unroll
	for (int i = 0; i < N; ++i) x[i] = v[i * stride];
	Idct1D<N>::run(x, hs);
#pragma unroll
	for (int i = 0; i < N; ++i) v[i * stride] = x[i];
}

// (the loops below have constant bounds and are fully unrolled on the device, so that a caller holding `buf` and `scratch` in
// local arrays gets them in registers: k_vardct_special)
#ifdef __HIPCC__
#define J40_UNROLL _Pragma("unroll")
#else
#define J40_UNROLL
#endif
// ---- the 8x8 special transforms; `buf` holds 64 coefficients in, 64 samples out (row-major 8x8),
//      `scratch` is 64 floats of private workspace ----

J40_DEV void aux_idct2x2(float *out, const float *in, int x, int y, int S2) {  // j40.h:5993
	const int p = y * 8 + x, q = (y * 2) * 8 + (x * 2);
	const float c00 = in[p], c01 = in[p + S2], c10 = in[p + S2 * 8], c11 = in[p + S2 * 9];
	out[q] = c00 + c01 + c10 + c11;
	out[q + 1] = c00 + c01 - c10 - c11;
	out[q + 8] = c00 - c01 + c10 - c11;
	out[q + 9] = c00 - c01 - c10 + c11;
}

J40_DEV void inverse_dct2x2_pyramid(float *buf, float *scratch) {  // DctSelect 2, j40.h:6002
	{ float t[4]; { const float c00 = buf[0], c01 = buf[1], c10 = buf[8], c11 = buf[9];
		t[0] = c00 + c01 + c10 + c11; t[1] = c00 + c01 - c10 - c11; t[2] = c00 - c01 + c10 - c11; t[3] = c00 - c01 - c10 + c11; }
	  buf[0] = t[0]; buf[1] = t[1]; buf[8] = t[2]; buf[9] = t[3]; }
	J40_UNROLL
	for (int i = 0; i < 64; ++i)
		scratch[i] = buf[i];
	J40_UNROLL
	for (int y = 0; y < 2; ++y)
		J40_UNROLL
		for (int x = 0; x < 2; ++x)
			aux_idct2x2(scratch, buf, x, y, 2);
	J40_UNROLL
	for (int y = 0; y < 4; ++y)
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			aux_idct2x2(buf, scratch, x, y, 4);
}

// columnar 4-point IDCTs over `rep` interleaved columns: out/in layout [4][rep]
J40_DEV void idct4_columns(float *out, const float *in, int rep, const float *hs) {
	J40_UNROLL
	for (int r = 0; r < rep; ++r)
		{
		float x[4] = {in[r], in[rep + r], in[2 * rep + r], in[3 * rep + r]};
		Idct1D<4>::run(x, hs);
		out[r] = x[0]; out[rep + r] = x[1]; out[2 * rep + r] = x[2]; out[3 * rep + r] = x[3];
	}
}
J40_DEV void idct8_columns(float *out, const float *in, int rep, const float *hs) {
	J40_UNROLL
	for (int r = 0; r < rep; ++r)
		{
		float x[8];
		J40_UNROLL
		for (int i = 0; i < 8; ++i)
			x[i] = in[i * rep + r];
		Idct1D<8>::run(x, hs);
		J40_UNROLL
		for (int i = 0; i < 8; ++i)
			out[i * rep + r] = x[i];
	}
}

J40_DEV void inverse_dct4x4_quad(float *buf, float *scratch, const float *hs) {  // DctSelect 3, j40.h:6015
	{ const float c00 = buf[0], c01 = buf[1], c10 = buf[8], c11 = buf[9];
	  buf[0] = c00 + c01 + c10 + c11; buf[1] = c00 + c01 - c10 - c11; buf[8] = c00 - c01 + c10 - c11; buf[9] = c00 - c01 - c10 + c11; }
	idct4_columns(scratch, buf, 16, hs);
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[x * 8 + y] = scratch[y * 8 + x];
	idct4_columns(scratch, buf, 16, hs);
	J40_UNROLL
	for (int y = 0; y < 4; ++y)
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			{
		buf[y * 8 + x] = scratch[(y * 2) * 8 + (x * 2)];
		buf[y * 8 + (x + 4)] = scratch[(y * 2 + 1) * 8 + (x * 2)];
		buf[(y + 4) * 8 + x] = scratch[(y * 2) * 8 + (x * 2 + 1)];
		buf[(y + 4) * 8 + (x + 4)] = scratch[(y * 2 + 1) * 8 + (x * 2 + 1)];
	}
}

J40_DEV void inverse_hornuss(float *buf, float *scratch) {  // DctSelect 1, j40.h:6046
	J40_UNROLL
	for (int i = 0; i < 64; ++i)
		scratch[i] = buf[i];
	aux_idct2x2(scratch, buf, 0, 0, 1);
	J40_UNROLL
	for (int y = 0; y < 2; ++y)
		J40_UNROLL
		for (int x = 0; x < 2; ++x)
			{
		const int pos00 = y * 8 + x, pos11 = (y + 2) * 8 + (x + 2);
		float rsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
		J40_UNROLL
		for (int iy = 0; iy < 4; ++iy)
			J40_UNROLL
			for (int ix = 0; ix < 4; ++ix)
				rsum[ix] += scratch[(y + iy * 2) * 8 + (x + ix * 2)];
		const float sample11 = scratch[pos00] - (rsum[0] + rsum[1] + rsum[2] + rsum[3] - scratch[pos00]) * 0.0625f;
		scratch[pos00] = scratch[pos11];
		scratch[pos11] = 0.0f;
		J40_UNROLL
		for (int iy = 0; iy < 4; ++iy)
			J40_UNROLL
			for (int ix = 0; ix < 4; ++ix)
				buf[(4 * y + iy) * 8 + (4 * x + ix)] = scratch[(y + iy * 2) * 8 + (x + ix * 2)] + sample11;
	}
}

J40_DEV void inverse_dct8x4(float *buf, float *scratch, const float *hs) {  // DctSelect 13 ("DCT32"), j40.h:6067
	{ const float t = buf[0] + buf[8]; buf[8] = buf[0] - buf[8]; buf[0] = t; }
	// two 4x8 coefficient matrices from even / odd rows: view buf as [4][16], IDCT4 down the columns
	idct4_columns(scratch, buf, 16, hs);
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[x * 8 + y] = scratch[y * 8 + x];
	idct8_columns(scratch, buf, 8, hs);
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[y * 8 + (((x & 1) << 2) | (x >> 1))] = scratch[y * 8 + x];
}

J40_DEV void inverse_dct4x8(float *buf, float *scratch, const float *hs) {  // DctSelect 12 ("DCT23"), j40.h:6087
	J40_UNROLL
	for (int i = 0; i < 64; ++i)
		scratch[i] = buf[i];
	scratch[0] = buf[0] + buf[8];
	scratch[8] = buf[0] - buf[8];
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[x * 8 + y] = scratch[y * 8 + x];
	idct8_columns(scratch, buf, 8, hs);
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[x * 8 + y] = scratch[y * 8 + x];
	idct4_columns(scratch, buf, 16, hs);
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		{ const int oy = ((y & 1) << 2) | (y >> 1); for (int x = 0; x < 8; ++x) buf[oy * 8 + x] = scratch[y * 8 + x]; }
}

J40_DEV void inverse_afv(float *buf, float *scratch, int flipx, int flipy, const float *hs, const float *afv_basis) {  // DctSelect 14-17, j40.h:6183
	float *bufafv = buf, *buf22 = buf + 16, *buf32 = buf + 32;
	float *safv = scratch, *s22 = scratch + 16, *s32 = scratch + 32;
	J40_UNROLL
	for (int y = 0; y < 8; y += 2)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			scratch[(x % 2) * 16 + (y / 2) * 4 + (x / 2)] = buf[y * 8 + x];
	J40_UNROLL
	for (int y = 1; y < 8; y += 2)
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			s32[x * 4 + (y / 2)] = buf[y * 8 + x];
	safv[0] = (buf[0] + buf[1] + buf[8]) * 4.0f;
	s22[0] = buf[0] - buf[1] + buf[8];
	s32[0] = buf[0] - buf[8];
	J40_UNROLL
	for (int i = 0; i < 16; ++i)
		{  // AFV 4x4: dense 16x16 basis product (j40.h:6176-6180)
		float sum = 0.0f;
		J40_UNROLL
		for (int j = 0; j < 16; ++j)
			sum += safv[j] * afv_basis[i * 16 + j];
		bufafv[i] = sum;
	}
	idct4_columns(buf22, s22, 4, hs);
	idct8_columns(buf32, s32, 4, hs);
	J40_UNROLL
	for (int y = 0; y < 4; ++y)
		{
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			safv[y * 4 + x] = bufafv[y * 4 + x];
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			s22[x * 4 + y] = buf22[y * 4 + x];
	}
	J40_UNROLL
	for (int y = 0; y < 8; ++y)
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			s32[x * 8 + y] = buf32[y * 4 + x];
	idct4_columns(buf22, s22, 4, hs);
	idct4_columns(buf32, s32, 8, hs);
	J40_UNROLL
	for (int i = 16; i < 64; ++i)
		scratch[i] = buf[i];
	J40_UNROLL
	for (int y = 0; y < 4; ++y)
		{
		const int ay = flipy ? 7 - y : y;
		const int dct22pos = (flipy * 4 + y) * 8 + (!flipx * 4);
		const int dct23pos = (!flipy * 4 + y) * 8;
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			buf[ay * 8 + (flipx ? 7 - x : x)] = safv[y * 4 + x];
		J40_UNROLL
		for (int x = 0; x < 4; ++x)
			buf[dct22pos + x] = s22[y * 4 + x];
		J40_UNROLL
		for (int x = 0; x < 8; ++x)
			buf[dct23pos + x] = s32[y * 8 + x];
	}
}

// dispatch for the 8x8 class other than plain DCT8x8 (j40.h:7178-7187)
J40_DEV void inverse_special8x8(int dctsel, float *buf, float *scratch, const float *hs, const float *afv_basis) {
	switch (dctsel) {
	case 1: inverse_hornuss(buf, scratch); break;
	case 2: inverse_dct2x2_pyramid(buf, scratch); break;
	case 3: inverse_dct4x4_quad(buf, scratch, hs); break;
	case 12: inverse_dct4x8(buf, scratch, hs); break;
	case 13: inverse_dct8x4(buf, scratch, hs); break;
	case 14: inverse_afv(buf, scratch, 0, 0, hs, afv_basis); break;
	case 15: inverse_afv(buf, scratch, 1, 0, hs, afv_basis); break;
	case 16: inverse_afv(buf, scratch, 0, 1, hs, afv_basis); break;
	default: inverse_afv(buf, scratch, 1, 1, hs, afv_basis); break;
	}
}

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
