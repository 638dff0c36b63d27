// j40_amd/csrc/device/large_dev.h -- the inverse DCTs with a 128- or 256-point side (DctSelect 21-26; j40__inverse_dct2d,
// j40.h:5972-5990, over j40__inverse_dct_core's recursion, j40.h:5802-5841), as the workgroup of k_vardct_large runs them (kernels.hip).
//
// A 128- or 256-point vector does not fit a lane's registers, a 64-point one does (Idct1D<64>, what the 64x64 kernel runs). The
// recursion says: split the vector into its even-indexed half and the "odd" half (sqrt2 * x[1], x[2i-1] + x[2i+1]), transform both
// halves, combine them with the half secants. So the top one (128) or two (256) levels of the recursion run as LEVELS over the whole
// tile in LDS -- every lane takes butterflies of many vectors, one barrier per level -- and below them each lane takes ONE 64-point
// sub-vector through Idct1D<64> in its registers. Every value sees the operations of the recursion in the recursion's order, so the
// results are the reference's bit for bit (the levels alone, down to length 2, are what round 3's kernel ran: thirteen to fifteen
// barriers per dimension and an integer division per butterfly; this is three or five and none).
//
// Layout: element k of vector v sits at buf[k * sk + v * sv] (one of the strides is 1, the other the tile's odd pitch: both
// dimensions of a tile walk LDS conflict-free, the lanes of a wavefront taking consecutive vectors). The number of vectors is a
// power of two. The functions take (tid, nthreads) and contain no barrier: the kernel runs one call per lane and a barrier behind
// it, tests/hostsim runs the calls of all lanes one after the other (Exec, below) -- the same code either way.
#pragma once
#include "idct_dev.h"
#include "vardct_dev.h"

namespace j40hip {

#ifdef __HIPCC__
#define J40_MUL24(a, b) __mul24((a), (b))   // (indices stay below 2^24: a full-rate multiply)
#else
#define J40_MUL24(a, b) ((a) * (b))
#endif

struct LargeDim { int32_t t, lvec, sk, sv; };   // vectors of length 1 << t (t = 7, 8), 1 << lvec of them, strides as above

// depth d of the recursion, downwards: every sub-vector of length n = N >> d is split into its halves (j40.h:5812-5823)
template <class Ptr>
J40_DEV void large_split_level(Ptr src, Ptr dst, const LargeDim &D, int32_t d, int32_t tid, int32_t nthreads) {
	const int32_t lhn = D.t - d - 1, hn = 1 << lhn, total = 1 << (D.t - 1 + D.lvec), vmask = (1 << D.lvec) - 1;
	for (int32_t w = tid; w < total; w += nthreads) {
		const int32_t v = w & vmask, j = w >> D.lvec, i = j & (hn - 1), o = (j >> lhn) << (lhn + 1);
		const int32_t base = J40_MUL24(v, D.sv);
		const float even = src[base + J40_MUL24(o + 2 * i, D.sk)];
		const float odd = i == 0 ? J40_SQRT2F * src[base + J40_MUL24(o + 1, D.sk)]
		                         : src[base + J40_MUL24(o + 2 * i - 1, D.sk)] + src[base + J40_MUL24(o + 2 * i + 1, D.sk)];
		dst[base + J40_MUL24(o + i, D.sk)] = even;
		dst[base + J40_MUL24(o + hn + i, D.sk)] = odd;
	}
}

// depth d, upwards: the transformed halves of every sub-vector of length n = N >> d are combined (j40.h:5832-5840)
template <class Ptr>
J40_DEV void large_combine_level(Ptr src, Ptr dst, const LargeDim &D, int32_t d, int32_t tid, int32_t nthreads, const float *hs) {
	const int32_t lhn = D.t - d - 1, hn = 1 << lhn, n = hn << 1, total = 1 << (D.t - 1 + D.lvec), vmask = (1 << D.lvec) - 1;
	for (int32_t w = tid; w < total; w += nthreads) {
		const int32_t v = w & vmask, j = w >> D.lvec, i = j & (hn - 1), o = (j >> lhn) << (lhn + 1);
		const int32_t base = J40_MUL24(v, D.sv);
		const float x = src[base + J40_MUL24(o + i, D.sk)], y = src[base + J40_MUL24(o + hn + i, D.sk)];
		const float ym = y * hs[hn + i];
		dst[base + J40_MUL24(o + i, D.sk)] = x + ym;
		dst[base + J40_MUL24(o + n - 1 - i, D.sk)] = x - ym;
	}
}

// below the levels: every 64-point sub-vector (elements [64 j, 64 j + 64) of its vector) through Idct1D<64>, in place, one per lane
template <class Ptr>
J40_DEV void large_idct64(Ptr buf, const LargeDim &D, int32_t tid, int32_t nthreads, const float *hs) {
	const int32_t total = 1 << (D.t - 6 + D.lvec), vmask = (1 << D.lvec) - 1;
	for (int32_t w = tid; w < total; w += nthreads) {
		const int32_t v = w & vmask, j = w >> D.lvec;
		Ptr p = buf + (J40_MUL24(v, D.sv) + J40_MUL24(64 * j, D.sk));
		float x[64];
#pragma unroll
		for (int k = 0; k < 64; ++k) x[k] = p[k * D.sk];
		Idct1D<64>::run(x, hs);
#pragma unroll
		for (int k = 0; k < 64; ++k) p[k * D.sk] = x[k];
	}
}

// One dimension of the tile in `a` (`b`: a buffer of the same size to split and combine into). The result is in `a` again.
// ex.run(f) calls f(tid, nthreads) for every lane of the workgroup and puts a barrier behind it.
template <class Exec, class Ptr>
J40_DEV void large_dim_pass(Exec &ex, Ptr a, Ptr b, const LargeDim &D, const float *hs) {
	const int32_t levels = D.t - 6;
	Ptr cur = a, other = b;
	for (int32_t d = 0; d < levels; ++d) {
		ex.run([&](int32_t tid, int32_t n) { large_split_level(cur, other, D, d, tid, n); });
		Ptr x = cur; cur = other; other = x;
	}
	ex.run([&](int32_t tid, int32_t n) { large_idct64(cur, D, tid, n, hs); });
	for (int32_t d = levels - 1; d >= 0; --d) {
		ex.run([&](int32_t tid, int32_t n) { large_combine_level(cur, other, D, d, tid, n, hs); });
		Ptr x = cur; cur = other; other = x;
	}
}

// Both dimensions of a tile that sits in LDS row-major with rows P = columns + 1 apart (rows * P floats in `a`; `b`: as many to work
// in): columns first like the reference (IDCT along c for every r, then along r for every x; j40.h:5972-5990). Result in `a`.
constexpr int LARGE_PANEL_FLOATS = 16384 + 256;
template <class Exec, class Ptr>
J40_DEV void large_tile_dims(Exec &ex, Ptr a, Ptr b, int32_t log_rows, int32_t log_columns, const float *hs) {
	const int32_t P = (1 << log_columns) + 1;
	const LargeDim along_c = {log_columns, log_rows, 1, P}, along_r = {log_rows, log_columns, P, 1};
	if (log_columns > 6) large_dim_pass(ex, a, b, along_c, hs);
	else ex.run([&](int32_t tid, int32_t n) { large_idct64(a, along_c, tid, n, hs); });   // (64 columns: no level above the registers)
	if (log_rows > 6) large_dim_pass(ex, a, b, along_r, hs);
	else ex.run([&](int32_t tid, int32_t n) { large_idct64(a, along_r, tid, n, hs); });
}

// A channel whose tile fits one LDS buffer (128x128, 128x64, 64x128) but lives in the workgroup's scratch (src, dst: row-major,
// rows x columns): in, both dimensions, out -- one trip through the scratch. la, lb: two buffers of LARGE_PANEL_FLOATS floats.
template <class Exec, class Ptr>
J40_DEV void large_tile_in_lds(Exec &ex, const float *src, float *dst, int32_t log_rows, int32_t log_columns, Ptr la, Ptr lb, const float *hs) {
	const int32_t C = 1 << log_columns, size = 1 << (log_rows + log_columns), P = C + 1;
	ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < size; i += n) la[J40_MUL24(i >> log_columns, P) + (i & (C - 1))] = src[i]; });
	large_tile_dims(ex, la, lb, log_rows, log_columns, hs);
	ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < size; i += n) dst[i] = la[J40_MUL24(i >> log_columns, P) + (i & (C - 1))]; });
}

// A channel whose tile does not fit (a 256-point side): one dimension at a time, a PANEL of M vectors through LDS at a time
// (M << t <= 16384 floats, the panel's rows M + 1 apart). Element k of vector v of the dimension at src[k * stride_k + v * stride_col]
// (one of the strides is 1); dst has the same layout; src is left alone.
template <class Exec, class Ptr>
J40_DEV void large_panels(Exec &ex, const float *src, float *dst, int32_t t, int32_t log_nvec, int32_t stride_k, int32_t stride_col, Ptr la, Ptr lb, const float *hs) {
	const int32_t lm = log_nvec < 14 - t ? log_nvec : 14 - t, M = 1 << lm, P = M + 1, N = 1 << t;
	const LargeDim D = {t, lm, P, 1};
	for (int32_t v0 = 0; v0 < (1 << log_nvec); v0 += M) {
		ex.run([&](int32_t tid, int32_t n) {
			if (stride_k == 1) for (int32_t w = tid; w < (N << lm); w += n) { const int32_t m = w >> t, k = w & (N - 1); la[J40_MUL24(k, P) + m] = src[(size_t) (v0 + m) * (size_t) stride_col + (size_t) k]; }
			else for (int32_t w = tid; w < (N << lm); w += n) { const int32_t k = w >> lm, m = w & (M - 1); la[J40_MUL24(k, P) + m] = src[(size_t) k * (size_t) stride_k + (size_t) (v0 + m)]; }
		});
		large_dim_pass(ex, la, lb, D, hs);
		ex.run([&](int32_t tid, int32_t n) {
			if (stride_k == 1) for (int32_t w = tid; w < (N << lm); w += n) { const int32_t m = w >> t, k = w & (N - 1); dst[(size_t) (v0 + m) * (size_t) stride_col + (size_t) k] = la[J40_MUL24(k, P) + m]; }
			else for (int32_t w = tid; w < (N << lm); w += n) { const int32_t k = w >> lm, m = w & (M - 1); dst[(size_t) k * (size_t) stride_k + (size_t) (v0 + m)] = la[J40_MUL24(k, P) + m]; }
		});
	}
}

// One varblock with a 128- or 256-point side, coefficients to samples: where the three channels' samples are when it returns
// (sample (y, x) of channel c at p[c][y * pitch[c] + x]). `panels`: 2 * LARGE_PANEL_FLOATS floats of LDS; A, B: the workgroup's
// scratch in HBM, three channels of 65536 floats each. Where the tile lives:
//   128x64, 64x128   single-pass frames: all three channels in LDS from the scatter of the coefficient events to the colour
//                    conversion (three tiles and a work buffer are exactly the 133 KB); the scratch is not touched
//   128x128          single-pass frames: a channel at a time in LDS -- zeroed, its events scattered into it (tile_scatter_channel),
//                    both dimensions -- the first two channels' samples parked in the scratch, the third read from LDS
//   256-sized        a channel is 256 KB (128 KB for 256x128, without room to split into): the tile in the scratch, one dimension at a
//                    time through LDS in panels
//   multi-pass frames (dense coefficient planes; rare with such transforms): the tile in the scratch, tiles up to 128x128 take both
//                    dimensions in one trip through LDS
struct LargeSamples { const float *p[3]; int32_t pitch[3]; };

template <class Exec, class Ptr>
J40_DEV LargeSamples large_block(Exec &ex, const DevPlan &plan, const DevVarblock &vb, const VbGeom &g, Ptr panels, float *A, float *B, const float *hs) {
	const DevFrame &f = *plan.frame;
	const int32_t log_rows = DEV_DCT_SELECT[vb.dctsel][0], log_columns = DEV_DCT_SELECT[vb.dctsel][1];
	const int32_t R = 1 << log_rows, C = 1 << log_columns, size = R * C, P = C + 1;
	const int32_t long_side = R > C ? R : C, vh8 = (R < C ? R : C) / 8, vw8 = long_side / 8;
	const int32_t param_idx = vb.dctsel == 21 ? 13 : vb.dctsel <= 23 ? 14 : vb.dctsel == 24 ? 15 : 16;
	const uint16_t *order = plan.pool_u16 + f.order_off[DEV_DCT_SELECT[vb.dctsel][2] * 3];
	const float *dq_scan = plan.pool_f32 + f.dq_scan_off[param_idx];
	const uint32_t *be = plan.block_events + 4 * (size_t) vb.blk;
	const float qbias[3] = {f.quant_bias[0], f.quant_bias[1], f.quant_bias[2]};
	Ptr la = panels, lb = panels + LARGE_PANEL_FLOATS;
	LargeSamples out;
	if (f.sparse_coeffs && size <= 16384) {
		const TileMapLog map = {log_rows, log_columns, P};
		if (R * P <= LARGE_PANEL_FLOATS / 2) {
			constexpr int TS = LARGE_PANEL_FLOATS / 2;
			Ptr work = panels + 3 * TS;
			ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < 3 * TS; i += n) panels[i] = 0.0f; });
			ex.run([&](int32_t tid, int32_t n) {
				tile_scatter_events(plan, g, be, order, dq_scan, size, map, (float *) panels, TS, qbias, f.quant_bias_num, tid, n);
				tile_fill_llf(plan, g, long_side, vh8, vw8, map, (float *) panels, TS, f.kx_lf, f.kb_lf, tid, n);
			});
			for (int ch = 0; ch < 3; ++ch) {
				large_tile_dims(ex, panels + ch * TS, work, log_rows, log_columns, hs);
				out.p[ch] = (const float *) (panels + ch * TS); out.pitch[ch] = P;
			}
			return out;
		}
		for (int ch = 0; ch < 3; ++ch) {
			ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < R * P; i += n) la[i] = 0.0f; });
			ex.run([&](int32_t tid, int32_t n) {
				tile_scatter_channel(plan, g, be, ch, order, dq_scan, size, map, la, qbias, f.quant_bias_num, tid, n);
				tile_fill_llf_channel(plan, g, ch, long_side, vh8, vw8, map, la, f.kx_lf, f.kb_lf, tid, n);
			});
			large_tile_dims(ex, la, lb, log_rows, log_columns, hs);
			if (ch < 2) {
				float *park = B + ch * 65536;
				ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < size; i += n) park[i] = la[J40_MUL24(i >> log_columns, P) + (i & (C - 1))]; });
				out.p[ch] = park; out.pitch[ch] = C;
			} else { out.p[ch] = (const float *) la; out.pitch[ch] = P; }
		}
		return out;
	}
	// the tile in the scratch
	if (f.sparse_coeffs) {
		const TileMapLog map = {log_rows, log_columns, C};
		ex.run([&](int32_t tid, int32_t n) { for (int32_t i = tid; i < size; i += n) { A[i] = 0.0f; A[65536 + i] = 0.0f; A[2 * 65536 + i] = 0.0f; } });
		ex.run([&](int32_t tid, int32_t n) {
			tile_scatter_events(plan, g, be, order, dq_scan, size, map, A, 65536, qbias, f.quant_bias_num, tid, n);
			tile_fill_llf(plan, g, long_side, vh8, vw8, map, A, 65536, f.kx_lf, f.kb_lf, tid, n);
		});
	} else {
		const float *dq = plan.pool_f32 + f.dq_off[param_idx];
		ex.run([&](int32_t tid, int32_t n) {
			for (int32_t i = tid; i < size; i += n) {
				float v[3];
				load_coeff3(plan, g, dq, size, i, long_side, vh8, vw8, v);
				const int32_t r = C > R ? i >> log_columns : i & (R - 1), c = C > R ? i & (C - 1) : i >> log_rows;
				A[r * C + c] = v[0]; A[65536 + r * C + c] = v[1]; A[2 * 65536 + r * C + c] = v[2];
			}
		});
	}
	const bool fits = size <= 16384;
	for (int ch = 0; ch < 3; ++ch) {
		if (fits) large_tile_in_lds(ex, (const float *) A + ch * 65536, B + ch * 65536, log_rows, log_columns, la, lb, hs);
		else {
			large_panels(ex, (const float *) A + ch * 65536, B + ch * 65536, log_columns, log_rows, 1, C, la, lb, hs);   // along c for every r: A -> B
			large_panels(ex, (const float *) B + ch * 65536, A + ch * 65536, log_rows, log_columns, C, 1, la, lb, hs);   // along r for every x: B -> A
		}
		out.p[ch] = (fits ? B : A) + ch * 65536; out.pitch[ch] = C;
	}
	return out;
}

} // namespace j40hip
