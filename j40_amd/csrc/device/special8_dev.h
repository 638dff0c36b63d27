// j40_amd/csrc/device/special8_dev.h -- the nine 8x8 "special" transforms of VarDCT (Hornuss, DCT2x2, DCT4x4, DCT4x8, DCT8x4
// and the four AFV orientations; DctSelect 1-3 and 12-17) as COOPERATIVE device functions: eight lanes share one 8x8 tile.
//
// Each transform is stated here as two phases of independent 8-lane work over the tile, derived from what the transform
// computes (which coefficients feed which 1-D transform, and where each result lands), not from the reference's buffer
// shuffles: the reference (j40.h:5993-6246) interleaves, transposes and re-packs through scratch arrays between its passes;
// here every lane gathers its inputs straight from where they lie and scatters its results straight to where they belong.
//   phase 0 reads the coefficient tile `src` and writes the work tile `mid` (any layout the transform likes);
//   phase 1 reads `mid` and writes the samples into `src`, row-major (sample (y, x) at src[8 y + x]).
// Both phases are out of place, so the lanes of a phase may run in any order (a wavefront runs them in lockstep; tests/hostsim
// one after the other); the caller separates the phases with a barrier.
// Only the ORDER OF THE FLOATING-POINT OPERATIONS is the reference's, because the results have to round the same way:
// the 4- and 8-point inverse DCTs are Idct1D<4> / Idct1D<8> (idct_dev.h), the 2x2 sums add left to right, the AFV basis
// product accumulates from zero in index order (j40.h:6176-6180).
#pragma once
#include "idct_dev.h"

namespace j40hip {

// the four sums / differences of a 2x2 group, each evaluated left to right
struct Quad { float pp, pm, mp, mm; };
J40_DEV Quad quad_sums(float a, float b, float c, float d) {
	Quad q;
	q.pp = a + b + c + d; q.pm = a + b - c - d; q.mp = a - b + c - d; q.mm = a - b - c + d;
	return q;
}

J40_DEV void idct4_pair(float *lo, float *hi, const float *hs) { Idct1D<4>::run(lo, hs); Idct1D<4>::run(hi, hs); }

// where row / column k of an interleaved pair of 4-point transforms ends up: even members first, then the odd ones
J40_DEV int deinterleave8(int k) { return ((k & 1) << 2) | (k >> 1); }

// ---- DctSelect 13: two 4-sample-wide halves side by side. Columns carry two interleaved 4-point transforms (even / odd
// coefficient rows), rows one 8-point transform; the result is written transposed with the halves de-interleaved. ----
J40_DEV void tall_halves_phase0(int lane, const float *src, float *mid, const float *hs) {
	float even[4], odd[4];
	for (int i = 0; i < 4; ++i) { even[i] = src[16 * i + lane]; odd[i] = src[16 * i + 8 + lane]; }
	if (lane == 0) { even[0] = src[0] + src[8]; odd[0] = src[0] - src[8]; }   // the two lowest frequencies are a sum / difference pair
	idct4_pair(even, odd, hs);
	for (int i = 0; i < 4; ++i) { mid[16 * i + lane] = even[i]; mid[16 * i + 8 + lane] = odd[i]; }
}
J40_DEV void tall_halves_phase1(int lane, const float *mid, float *dst, const float *hs) {
	float v[8];
	for (int k = 0; k < 8; ++k) v[k] = mid[8 * lane + k];
	Idct1D<8>::run(v, hs);
	const int column = deinterleave8(lane);
	for (int k = 0; k < 8; ++k) dst[8 * k + column] = v[k];
}

// ---- DctSelect 12: two 4-sample-high halves on top of each other: rows first (8-point), then the interleaved columns ----
J40_DEV void wide_halves_phase0(int lane, const float *src, float *mid, const float *hs) {
	float v[8];
	for (int k = 0; k < 8; ++k) v[k] = src[8 * lane + k];
	if (lane == 0) v[0] = src[0] + src[8];
	if (lane == 1) v[0] = src[0] - src[8];
	Idct1D<8>::run(v, hs);
	for (int k = 0; k < 8; ++k) mid[8 * lane + k] = v[k];
}
J40_DEV void wide_halves_phase1(int lane, const float *mid, float *dst, const float *hs) {
	float even[4], odd[4];
	for (int i = 0; i < 4; ++i) { even[i] = mid[16 * i + lane]; odd[i] = mid[16 * i + 8 + lane]; }
	idct4_pair(even, odd, hs);
	for (int i = 0; i < 4; ++i) { dst[8 * i + lane] = even[i]; dst[8 * (4 + i) + lane] = odd[i]; }
}

// ---- DctSelect 3: four 4x4 quadrants, their coefficients interleaved in both directions; the four lowest frequencies are
// the 2x2 sums of the quadrants' own ----
J40_DEV void quadrants_phase0(int lane, const float *src, float *mid, const float *hs) {
	float even[4], odd[4];
	for (int i = 0; i < 4; ++i) { even[i] = src[16 * i + lane]; odd[i] = src[16 * i + 8 + lane]; }
	if (lane < 2) {
		const Quad q = quad_sums(src[0], src[1], src[8], src[9]);
		even[0] = lane == 0 ? q.pp : q.pm; odd[0] = lane == 0 ? q.mp : q.mm;
	}
	idct4_pair(even, odd, hs);
	for (int i = 0; i < 4; ++i) { mid[16 * i + lane] = even[i]; mid[16 * i + 8 + lane] = odd[i]; }
}
J40_DEV void quadrants_phase1(int lane, const float *mid, float *dst, const float *hs) {
	float even[4], odd[4];
	for (int i = 0; i < 4; ++i) { even[i] = mid[8 * lane + 2 * i]; odd[i] = mid[8 * lane + 2 * i + 1]; }
	idct4_pair(even, odd, hs);
	// coefficient row `lane` belongs to quadrant row (lane & 1) and becomes sample COLUMN lane >> 1 there; the even / odd
	// coefficient columns are the left / right quadrant
	const int top = 4 * (lane & 1), column = lane >> 1;
	for (int i = 0; i < 4; ++i) { dst[8 * (top + i) + column] = even[i]; dst[8 * (top + i) + column + 4] = odd[i]; }
}

// ---- DctSelect 2: a three-level pyramid of 2x2 sums: 1 group, then 4, then 16, each level doubling the resolved area ----
J40_DEV void pyramid_phase0(int lane, const float *src, float *mid) {
	if (lane == 0) {
		const Quad top = quad_sums(src[0], src[1], src[8], src[9]);
		const float base[4] = {top.pp, top.pm, top.mp, top.mm};   // the 2x2 the next level starts from
		for (int g = 0; g < 4; ++g) {
			const int gy = g >> 1, gx = g & 1;
			const Quad q = quad_sums(base[g], src[8 * gy + gx + 2], src[8 * (gy + 2) + gx], src[8 * (gy + 2) + gx + 2]);
			float *out = mid + 8 * (2 * gy) + 2 * gx;
			out[0] = q.pp; out[1] = q.pm; out[8] = q.mp; out[9] = q.mm;
		}
	}
	// everything outside the top-left 4x4 goes to the last level unchanged
	for (int k = lane < 4 ? 4 : 0; k < 8; ++k) mid[8 * lane + k] = src[8 * lane + k];
}
J40_DEV void pyramid_phase1(int lane, const float *mid, float *dst) {
	for (int g = 2 * lane; g < 2 * lane + 2; ++g) {
		const int gy = g >> 2, gx = g & 3;
		const Quad q = quad_sums(mid[8 * gy + gx], mid[8 * gy + gx + 4], mid[8 * (gy + 4) + gx], mid[8 * (gy + 4) + gx + 4]);
		float *out = dst + 8 * (2 * gy) + 2 * gx;
		out[0] = q.pp; out[1] = q.pm; out[8] = q.mp; out[9] = q.mm;
	}
}

// ---- DctSelect 1 (Hornuss): four 4x4 quadrants, each an average-plus-details form: the quadrant's 16 interleaved
// coefficients minus their mean term, with two of them exchanged. Two lanes per quadrant, each computes the quadrant's common
// term (cheap) and writes two of its four rows. ----
J40_DEV void hornuss_phase0(int lane, const float *src, float *mid) {
	const int quadrant = lane >> 1, qy = quadrant >> 1, qx = quadrant & 1, half = lane & 1;
	float e[4][4];
	for (int iy = 0; iy < 4; ++iy) for (int ix = 0; ix < 4; ++ix) e[iy][ix] = src[8 * (qy + 2 * iy) + qx + 2 * ix];
	{   // the quadrants' lowest coefficients are the 2x2 sums of the tile's four lowest
		const Quad q = quad_sums(src[0], src[1], src[8], src[9]);
		e[0][0] = quadrant == 0 ? q.pp : quadrant == 1 ? q.pm : quadrant == 2 ? q.mp : q.mm;
	}
	float column_sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (int iy = 0; iy < 4; ++iy) for (int ix = 0; ix < 4; ++ix) column_sum[ix] += e[iy][ix];
	const float common = e[0][0] - (column_sum[0] + column_sum[1] + column_sum[2] + column_sum[3] - e[0][0]) * 0.0625f;
	e[0][0] = e[1][1]; e[1][1] = 0.0f;
	for (int iy = 2 * half; iy < 2 * half + 2; ++iy) for (int ix = 0; ix < 4; ++ix) mid[8 * (4 * qy + iy) + 4 * qx + ix] = e[iy][ix] + common;
}
J40_DEV void copy_row_phase1(int lane, const float *mid, float *dst) { for (int k = 0; k < 8; ++k) dst[8 * lane + k] = mid[8 * lane + k]; }

// ---- DctSelect 14-17 (AFV): one 4x4 corner from a dense 16x16 basis product, a 4x4 DCT block beside it and a 4x8 DCT block
// over the other half; `mirror_x` / `mirror_y` pick the corner. Coefficients: even rows + even columns feed the corner, even
// rows + odd columns the 4x4 block, odd rows the 4x8 block; the three lowest frequencies are mixed.
// Work tile: [0, 16) the corner's samples, [16, 32) the 4x4 block after its first pass, [32, 64) the 4x8 block after its first ----
J40_DEV void afv_phase0(int lane, const float *src, float *mid, const float *hs, const float *basis) {
	{   // the corner: two of the sixteen outputs per lane
		float in[16];
		for (int k = 0; k < 16; ++k) in[k] = src[16 * (k >> 2) + 2 * (k & 3)];
		in[0] = (src[0] + src[1] + src[8]) * 4.0f;
		for (int o = 2 * lane; o < 2 * lane + 2; ++o) {
			float acc = 0.0f;
			for (int k = 0; k < 16; ++k) acc += in[k] * basis[16 * o + k];
			mid[o] = acc;
		}
	}
	if (lane < 4) {   // the 4x4 block, along its coefficient rows: one column per lane
		float v[4];
		for (int i = 0; i < 4; ++i) v[i] = src[16 * i + 2 * lane + 1];
		if (lane == 0) v[0] = src[0] - src[1] + src[8];
		Idct1D<4>::run(v, hs);
		for (int i = 0; i < 4; ++i) mid[16 + 4 * i + lane] = v[i];
	} else {          // the 4x8 block, along its eight columns: one coefficient row per lane
		const int row = lane - 4;
		float v[8];
		for (int k = 0; k < 8; ++k) v[k] = src[8 * (2 * row + 1) + k];
		if (row == 0) v[0] = src[0] - src[8];
		Idct1D<8>::run(v, hs);
		for (int k = 0; k < 8; ++k) mid[32 + 4 * k + row] = v[k];
	}
}
J40_DEV void afv_phase1(int lane, const float *mid, float *dst, const float *hs, int mirror_x, int mirror_y) {
	for (int o = 2 * lane; o < 2 * lane + 2; ++o) {   // the corner's samples go to their (possibly mirrored) places
		const int y = o >> 2, x = o & 3;
		dst[8 * (mirror_y ? 7 - y : y) + (mirror_x ? 7 - x : x)] = mid[o];
	}
	{   // the 4x8 block across its four rows: one column per lane; it fills the half of the tile the corner is not in
		float v[4];
		for (int i = 0; i < 4; ++i) v[i] = mid[32 + 4 * lane + i];
		Idct1D<4>::run(v, hs);
		const int top = mirror_y ? 0 : 4;
		for (int i = 0; i < 4; ++i) dst[8 * (top + i) + lane] = v[i];
	}
	if (lane < 4) {   // the 4x4 block's second pass; it sits beside the corner
		float v[4];
		for (int i = 0; i < 4; ++i) v[i] = mid[16 + 4 * lane + i];
		Idct1D<4>::run(v, hs);
		const int top = mirror_y ? 4 : 0, left = mirror_x ? 0 : 4;
		for (int i = 0; i < 4; ++i) dst[8 * (top + i) + left + lane] = v[i];
	}
}

// the two phases of DctSelect `sel` for one lane (0..7) of a tile
J40_DEV void special8_phase0(int sel, int lane, const float *src, float *mid, const float *hs, const float *afv_basis) {
	switch (sel) {
	case 1: hornuss_phase0(lane, src, mid); break;
	case 2: pyramid_phase0(lane, src, mid); break;
	case 3: quadrants_phase0(lane, src, mid, hs); break;
	case 12: wide_halves_phase0(lane, src, mid, hs); break;
	case 13: tall_halves_phase0(lane, src, mid, hs); break;
	default: afv_phase0(lane, src, mid, hs, afv_basis); break;
	}
}
J40_DEV void special8_phase1(int sel, int lane, const float *mid, float *dst, const float *hs) {
	switch (sel) {
	case 1: copy_row_phase1(lane, mid, dst); break;
	case 2: pyramid_phase1(lane, mid, dst); break;
	case 3: quadrants_phase1(lane, mid, dst, hs); break;
	case 12: wide_halves_phase1(lane, mid, dst, hs); break;
	case 13: tall_halves_phase1(lane, mid, dst, hs); break;
	default: afv_phase1(lane, mid, dst, hs, (sel - 14) & 1, (sel - 14) >> 1); break;
	}
}

} // namespace j40hip
