// j40_amd/csrc/device/special8_dev.h -- the nine 8x8 "special" transforms of VarDCT (Hornuss, DCT2x2, DCT4x4, DCT4x8, DCT8x4
// and the four AFV orientations; DctSelect 1-3 and 12-17) as COOPERATIVE device functions: eight lanes share one 8x8 tile.
//
// Each transform is stated here as two phases of independent 8-lane work over the tile, derived from what the transform
// computes (which coefficients feed which 1-D transform, and where each result lands), not from the reference's buffer
// shuffles: the reference (j40.h:5993-6246) interleaves, transposes and re-packs through scratch arrays between its passes;
// here every lane gathers its inputs straight from where they lie and scatters its results straight to where they belong.
//   phase 0 reads the coefficient tile and writes the intermediate values (any layout the transform likes);
//   phase 1 reads them and writes the samples, row-major (sample (y, x) at canonical index 8 y + x).
// Every phase loads everything it needs before it stores anything, so a phase may work IN PLACE on one tile when its eight lanes
// run in lockstep (one wavefront: the kernels), and out of place with the lanes in any order (tests/hostsim, one after the other).
// Only the ORDER OF THE FLOATING-POINT OPERATIONS is the reference's, because the results have to round the same way:
// the 4- and 8-point inverse DCTs are Idct1D<4> / Idct1D<8> (idct_dev.h), the 2x2 sums add left to right, the AFV basis
// product accumulates from zero in index order (j40.h:6176-6180).
#pragma once
#include "idct_dev.h"

namespace j40hip {

// A tile's 64 values lie in eight rows SP8_PITCH floats apart (72 floats per tile): with rows 9 apart and tiles 72 apart the eight
// lanes of a tile -- and the eight tiles of a wavefront -- hit distinct LDS banks whether the lanes walk along a row or down a
// column (tiles 65 floats apart with rows of 8 cost 1.9 conflict cycles per LDS instruction). SP8(i): where canonical index i lives.
enum { SP8_PITCH = 9, SP8_TILE = 72 };
#define SP8(i) ((((i) >> 3) * 9) + ((i) & 7))   /* SP8_PITCH */

// Between a phase's loads and its stores: the eight lanes of a tile belong to one wavefront, which executes them in lockstep and
// its LDS operations in order, so "every load of the phase is issued before any store" is all an in-place phase needs -- as long
// as the compiler keeps that order (it reasons per lane and would otherwise be free to move a store to one address above a load
// from another).
#ifdef __HIPCC__
#define SP8_LOADS_DONE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define SP8_LOADS_DONE() do { } while (0)
#endif

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
template <typename SRC, typename DST> J40_DEV void tall_halves_phase0(int lane, SRC src, DST dst, const float *hs) {
	float even[4], odd[4];
	#pragma unroll
	for (int i = 0; i < 4; ++i) { even[i] = src[SP8(16 * i + lane)]; odd[i] = src[SP8(16 * i + 8 + lane)]; }
	if (lane == 0) { const float a = even[0], b = odd[0]; even[0] = a + b; odd[0] = a - b; }   // the two lowest frequencies (positions 0 and 8) are a sum / difference pair
	idct4_pair(even, odd, hs);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int i = 0; i < 4; ++i) { dst[SP8(16 * i + lane)] = even[i]; dst[SP8(16 * i + 8 + lane)] = odd[i]; }
}
template <typename SRC, typename DST> J40_DEV void tall_halves_phase1(int lane, SRC mid, DST dst, const float *hs) {
	float v[8];
	#pragma unroll
	for (int k = 0; k < 8; ++k) v[k] = mid[SP8(8 * lane + k)];
	Idct1D<8>::run(v, hs);
	const int column = deinterleave8(lane);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int k = 0; k < 8; ++k) dst[SP8(8 * k + column)] = v[k];
}

// ---- DctSelect 12: two 4-sample-high halves on top of each other: rows first (8-point), then the interleaved columns ----
template <typename SRC, typename DST> J40_DEV void wide_halves_phase0(int lane, SRC src, DST dst, const float *hs) {
	float v[8];
	#pragma unroll
	for (int k = 0; k < 8; ++k) v[k] = src[SP8(8 * lane + k)];
	const float a = src[SP8(0)], b = src[SP8(8)];
	if (lane == 0) v[0] = a + b;
	if (lane == 1) v[0] = a - b;
	Idct1D<8>::run(v, hs);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int k = 0; k < 8; ++k) dst[SP8(8 * lane + k)] = v[k];
}
template <typename SRC, typename DST> J40_DEV void wide_halves_phase1(int lane, SRC mid, DST dst, const float *hs) {
	float even[4], odd[4];
	#pragma unroll
	for (int i = 0; i < 4; ++i) { even[i] = mid[SP8(16 * i + lane)]; odd[i] = mid[SP8(16 * i + 8 + lane)]; }
	idct4_pair(even, odd, hs);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int i = 0; i < 4; ++i) { dst[SP8(8 * i + lane)] = even[i]; dst[SP8(8 * (4 + i) + lane)] = odd[i]; }
}

// ---- DctSelect 3: four 4x4 quadrants, their coefficients interleaved in both directions; the four lowest frequencies are
// the 2x2 sums of the quadrants' own ----
template <typename SRC, typename DST> J40_DEV void quadrants_phase0(int lane, SRC src, DST dst, const float *hs) {
	float even[4], odd[4];
	#pragma unroll
	for (int i = 0; i < 4; ++i) { even[i] = src[SP8(16 * i + lane)]; odd[i] = src[SP8(16 * i + 8 + lane)]; }
	const float c00 = src[SP8(0)], c01 = src[SP8(1)], c10 = src[SP8(8)], c11 = src[SP8(9)];
	if (lane < 2) {
		const Quad q = quad_sums(c00, c01, c10, c11);
		even[0] = lane == 0 ? q.pp : q.pm; odd[0] = lane == 0 ? q.mp : q.mm;
	}
	idct4_pair(even, odd, hs);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int i = 0; i < 4; ++i) { dst[SP8(16 * i + lane)] = even[i]; dst[SP8(16 * i + 8 + lane)] = odd[i]; }
}
template <typename SRC, typename DST> J40_DEV void quadrants_phase1(int lane, SRC mid, DST dst, const float *hs) {
	float even[4], odd[4];
	#pragma unroll
	for (int i = 0; i < 4; ++i) { even[i] = mid[SP8(8 * lane + 2 * i)]; odd[i] = mid[SP8(8 * lane + 2 * i + 1)]; }
	idct4_pair(even, odd, hs);
	// coefficient row `lane` belongs to quadrant row (lane & 1) and becomes sample COLUMN lane >> 1 there; the even / odd
	// coefficient columns are the left / right quadrant
	const int top = 4 * (lane & 1), column = lane >> 1;
	SP8_LOADS_DONE();
	#pragma unroll
	for (int i = 0; i < 4; ++i) { dst[SP8(8 * (top + i) + column)] = even[i]; dst[SP8(8 * (top + i) + column + 4)] = odd[i]; }
}

// ---- DctSelect 2: a three-level pyramid of 2x2 sums: 1 group, then 4, then 16, each level doubling the resolved area ----
template <typename SRC, typename DST> J40_DEV void pyramid_phase0(int lane, SRC src, DST dst, bool in_place) {
	float row[8];   // this lane's row: passes through (out of place), apart from what lane 0 rewrites below
	#pragma unroll
	for (int k = 0; k < 8; ++k) row[k] = src[SP8(8 * lane + k)];
	float out[4][4];
	if (lane == 0) {   // the top-left 4x4 holds the first two levels
		float a[4][4];
		#pragma unroll
		for (int r = 0; r < 4; ++r) {
#pragma unroll
			for (int c = 0; c < 4; ++c) a[r][c] = src[SP8(8 * r + c)];
		}
		const Quad top = quad_sums(a[0][0], a[0][1], a[1][0], a[1][1]);
		const float base[4] = {top.pp, top.pm, top.mp, top.mm};   // the 2x2 the next level starts from
		#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int gy = g >> 1, gx = g & 1;
			const Quad q = quad_sums(base[g], a[gy][gx + 2], a[gy + 2][gx], a[gy + 2][gx + 2]);
			out[2 * gy][2 * gx] = q.pp; out[2 * gy][2 * gx + 1] = q.pm; out[2 * gy + 1][2 * gx] = q.mp; out[2 * gy + 1][2 * gx + 1] = q.mm;
		}
	}
	SP8_LOADS_DONE();
	if (lane == 0) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
#pragma unroll
			for (int c = 0; c < 4; ++c) dst[SP8(8 * r + c)] = out[r][c];
		}
	}
	// everything outside the top-left 4x4 goes to the last level unchanged
	if (!in_place) {
#pragma unroll
		for (int k = 0; k < 8; ++k) if (k >= 4 || lane >= 4) dst[SP8(8 * lane + k)] = row[k];
	}
}
template <typename SRC, typename DST> J40_DEV void pyramid_phase1(int lane, SRC mid, DST dst) {
	float in[2][4], out[2][4];
	#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int g = 2 * lane + h, gy = g >> 2, gx = g & 3;
		in[h][0] = mid[SP8(8 * gy + gx)]; in[h][1] = mid[SP8(8 * gy + gx + 4)]; in[h][2] = mid[SP8(8 * (gy + 4) + gx)]; in[h][3] = mid[SP8(8 * (gy + 4) + gx + 4)];
	}
	#pragma unroll
	for (int h = 0; h < 2; ++h) { const Quad q = quad_sums(in[h][0], in[h][1], in[h][2], in[h][3]); out[h][0] = q.pp; out[h][1] = q.pm; out[h][2] = q.mp; out[h][3] = q.mm; }
	SP8_LOADS_DONE();
	#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int g = 2 * lane + h, gy = g >> 2, gx = g & 3, at = 8 * (2 * gy) + 2 * gx;
		dst[SP8(at)] = out[h][0]; dst[SP8(at + 1)] = out[h][1]; dst[SP8(at + 8)] = out[h][2]; dst[SP8(at + 9)] = out[h][3];
	}
}

// ---- DctSelect 1 (Hornuss): four 4x4 quadrants, each an average-plus-details form: the quadrant's 16 interleaved
// coefficients minus their mean term, with two of them exchanged. Two lanes per quadrant, each computes the quadrant's common
// term (cheap) and writes two of its four rows. The samples are complete after phase 0. ----
template <typename SRC, typename DST> J40_DEV void hornuss_phase0(int lane, SRC src, DST dst) {
	const int quadrant = lane >> 1, qy = quadrant >> 1, qx = quadrant & 1, half = lane & 1;
	float e[4][4];
	#pragma unroll
	for (int iy = 0; iy < 4; ++iy) for (int ix = 0; ix < 4; ++ix) e[iy][ix] = src[SP8(8 * (qy + 2 * iy) + qx + 2 * ix)];
	{   // the quadrants' lowest coefficients are the 2x2 sums of the tile's four lowest
		const Quad q = quad_sums(src[SP8(0)], src[SP8(1)], src[SP8(8)], src[SP8(9)]);
		e[0][0] = quadrant == 0 ? q.pp : quadrant == 1 ? q.pm : quadrant == 2 ? q.mp : q.mm;
	}
	float column_sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	#pragma unroll
	for (int iy = 0; iy < 4; ++iy) for (int ix = 0; ix < 4; ++ix) column_sum[ix] += e[iy][ix];
	const float common = e[0][0] - (column_sum[0] + column_sum[1] + column_sum[2] + column_sum[3] - e[0][0]) * 0.0625f;
	// this lane's two rows of the quadrant, read again with the row as part of the ADDRESS (indexing e[][] by the lane's half made
	// the compiler park the array in scratch memory); e[0][0] and e[1][1] are exchanged, and the latter is the mean's place: 0
	float out[2][4];
#pragma unroll
	for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
		for (int ix = 0; ix < 4; ++ix) out[iy][ix] = src[SP8(8 * (qy + 2 * (2 * half + iy)) + qx + 2 * ix)];
	}
	out[0][0] = half ? out[0][0] : e[1][1];
	out[1][1] = half ? out[1][1] : 0.0f;
#pragma unroll
	for (int iy = 0; iy < 2; ++iy) {
#pragma unroll
		for (int ix = 0; ix < 4; ++ix) out[iy][ix] = out[iy][ix] + common;
	}
	SP8_LOADS_DONE();
	#pragma unroll
	for (int iy = 0; iy < 2; ++iy) for (int ix = 0; ix < 4; ++ix) dst[SP8(8 * (4 * qy + 2 * half + iy) + 4 * qx + ix)] = out[iy][ix];
}
template <typename SRC, typename DST> J40_DEV void copy_row_phase1(int lane, SRC mid, DST dst, bool in_place) {
	if (in_place) return;
	float row[8];
	#pragma unroll
	for (int k = 0; k < 8; ++k) row[k] = mid[SP8(8 * lane + k)];
	#pragma unroll
	for (int k = 0; k < 8; ++k) dst[SP8(8 * lane + k)] = row[k];
}

// ---- DctSelect 14-17 (AFV): one 4x4 corner from a dense 16x16 basis product, a 4x4 DCT block beside it and a 4x8 DCT block
// over the other half; `mirror_x` / `mirror_y` pick the corner. Coefficients: even rows + even columns feed the corner, even
// rows + odd columns the 4x4 block, odd rows the 4x8 block; the three lowest frequencies are mixed.
// Between the phases: [0, 16) the corner's samples, [16, 32) the 4x4 block after its first pass, [32, 64) the 4x8 block after its first ----
template <typename SRC, typename DST> J40_DEV void afv_phase0(int lane, SRC src, DST dst, const float *hs, const float *basis) {
	const float s0 = src[SP8(0)], s1 = src[SP8(1)], s8 = src[SP8(8)];
	float acc[2];
	{   // the corner: two of the sixteen outputs per lane
		float in[16];
		#pragma unroll
		for (int k = 0; k < 16; ++k) in[k] = src[SP8(16 * (k >> 2) + 2 * (k & 3))];
		in[0] = (s0 + s1 + s8) * 4.0f;
		#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int o = 2 * lane + h;
			float a = 0.0f;
			#pragma unroll
			for (int k = 0; k < 16; ++k) a += in[k] * basis[16 * o + k];
			acc[h] = a;
		}
	}
	float v[8];
	if (lane < 4) {   // the 4x4 block, along its coefficient rows: one column per lane
		#pragma unroll
		for (int i = 0; i < 4; ++i) v[i] = src[SP8(16 * i + 2 * lane + 1)];
		if (lane == 0) v[0] = s0 - s1 + s8;
		Idct1D<4>::run(v, hs);
	} else {          // the 4x8 block, along its eight columns: one coefficient row per lane
		const int row = lane - 4;
		#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = src[SP8(8 * (2 * row + 1) + k)];
		if (row == 0) v[0] = s0 - s8;
		Idct1D<8>::run(v, hs);
	}
	SP8_LOADS_DONE();
	dst[SP8(2 * lane)] = acc[0]; dst[SP8(2 * lane + 1)] = acc[1];
	if (lane < 4) { for (int i = 0; i < 4; ++i) dst[SP8(16 + 4 * i + lane)] = v[i]; }
	else { const int row = lane - 4; for (int k = 0; k < 8; ++k) dst[SP8(32 + 4 * k + row)] = v[k]; }
}
template <typename SRC, typename DST> J40_DEV void afv_phase1(int lane, SRC mid, DST dst, const float *hs, int mirror_x, int mirror_y) {
	const float c0 = mid[SP8(2 * lane)], c1 = mid[SP8(2 * lane + 1)];
	float v[4], w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	#pragma unroll
	for (int i = 0; i < 4; ++i) v[i] = mid[SP8(32 + 4 * lane + i)];   // the 4x8 block across its four rows: one column per lane
	if (lane < 4) {   // the 4x4 block's second pass
#pragma unroll
		for (int i = 0; i < 4; ++i) w[i] = mid[SP8(16 + 4 * lane + i)];
	}
	Idct1D<4>::run(v, hs);
	if (lane < 4) Idct1D<4>::run(w, hs);
	SP8_LOADS_DONE();
	#pragma unroll
	for (int h = 0; h < 2; ++h) {   // the corner's samples go to their (possibly mirrored) places
		const int o = 2 * lane + h, y = o >> 2, x = o & 3;
		dst[SP8(8 * (mirror_y ? 7 - y : y) + (mirror_x ? 7 - x : x))] = h ? c1 : c0;
	}
	{   // the 4x8 block fills the half of the tile the corner is not in
		const int top = mirror_y ? 0 : 4;
		#pragma unroll
		for (int i = 0; i < 4; ++i) dst[SP8(8 * (top + i) + lane)] = v[i];
	}
	if (lane < 4) {   // the 4x4 block sits beside the corner
		const int top = mirror_y ? 4 : 0, left = mirror_x ? 0 : 4;
		#pragma unroll
		for (int i = 0; i < 4; ++i) dst[SP8(8 * (top + i) + left + lane)] = w[i];
	}
}

// the two phases of DctSelect `sel` for one lane (0..7) of a tile. `src` and `dst` may be the same tile (the kernels: in place;
// tests/hostsim runs the lanes one after the other and therefore out of place)
template <typename SRC, typename DST> J40_DEV void special8_phase0(int sel, int lane, SRC src, DST dst, const float *hs, const float *afv_basis, bool in_place) {
	switch (sel) {
	case 1: hornuss_phase0(lane, src, dst); break;
	case 2: pyramid_phase0(lane, src, dst, in_place); break;
	case 3: quadrants_phase0(lane, src, dst, hs); break;
	case 12: wide_halves_phase0(lane, src, dst, hs); break;
	case 13: tall_halves_phase0(lane, src, dst, hs); break;
	default: afv_phase0(lane, src, dst, hs, afv_basis); break;
	}
}
template <typename SRC, typename DST> J40_DEV void special8_phase1(int sel, int lane, SRC mid, DST dst, const float *hs, bool in_place) {
	switch (sel) {
	case 1: copy_row_phase1(lane, mid, dst, in_place); break;
	case 2: pyramid_phase1(lane, mid, dst); break;
	case 3: quadrants_phase1(lane, mid, dst, hs); break;
	case 12: wide_halves_phase1(lane, mid, dst, hs); break;
	case 13: tall_halves_phase1(lane, mid, dst, hs); break;
	default: afv_phase1(lane, mid, dst, hs, (sel - 14) & 1, (sel - 14) >> 1); break;
	}
}

} // namespace j40hip
