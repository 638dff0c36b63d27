// tools/jxlsynth_forward.hpp -- the ANALYSIS side of VarDCT for the stream generator (test / bench infrastructure): a picture's
// XYB samples -> the quantised coefficients a distance-1 encoder would code ("forward DCT per varblock,
// quantise with the library matrices"), so that bits per pixel, symbols per pixel and the spread of the sections' lengths
// come from picture content instead of a tuned distribution.
//
// No forward transform is written out here. The decoder's SYNTHESIS side is the definition of every transform (27 DctSelect
// values incl. the nine 8x8 specials); its analysis side is obtained numerically: the synthesis is applied to unit vectors
// (1-D inverse DCTs of 8..64 points through Idct1D, the specials through the two cooperative phases of special8_dev.h) and the
// resulting matrix is inverted (Gauss-Jordan in double precision). Separable DCT blocks use one matrix per axis; canonical
// coefficient positions follow the decoder's tile mapping (tests/hostsim/hostsim.cpp, vardct_dev.h). Quantisation weights are the
// decoder's own expansion of the library matrices (tables.cpp, load_dq_matrix); the scan order is natural_order (tables.cpp).
// What a stream decodes to is still defined by the reference (oracle/_ref); tests/test_forward_streams.py checks that it decodes
// to the source picture within the expected distance.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include "../j40_amd/csrc/frame.hpp"
#include "../j40_amd/csrc/tables.hpp"
#include <memory>
#include <stdexcept>
#include "../j40_amd/csrc/device/entropy_dev.h"   // (the address-space macros the device headers use; empty on the CPU)
#include "../j40_amd/csrc/device/special8_dev.h"

namespace synthfwd {

using j40hip::Idct1D;

// in-place inverse of an n x n matrix (row-major); false if singular
inline bool invert(std::vector<double> &a, int n) {
	std::vector<double> inv((size_t) n * (size_t) n, 0.0);
	for (int i = 0; i < n; ++i) inv[(size_t) i * (size_t) n + (size_t) i] = 1.0;
	for (int col = 0; col < n; ++col) {
		int piv = col;
		for (int r = col + 1; r < n; ++r) if (std::fabs(a[(size_t) r * n + col]) > std::fabs(a[(size_t) piv * n + col])) piv = r;
		if (std::fabs(a[(size_t) piv * n + col]) < 1e-12) return false;
		if (piv != col) for (int k = 0; k < n; ++k) { std::swap(a[(size_t) piv * n + k], a[(size_t) col * n + k]); std::swap(inv[(size_t) piv * n + k], inv[(size_t) col * n + k]); }
		const double d = 1.0 / a[(size_t) col * n + col];
		for (int k = 0; k < n; ++k) { a[(size_t) col * n + k] *= d; inv[(size_t) col * n + k] *= d; }
		for (int r = 0; r < n; ++r) if (r != col) {
			const double f = a[(size_t) r * n + col];
			if (f == 0.0) continue;
			for (int k = 0; k < n; ++k) { a[(size_t) r * n + k] -= f * a[(size_t) col * n + k]; inv[(size_t) r * n + k] -= f * inv[(size_t) col * n + k]; }
		}
	}
	a.swap(inv);
	return true;
}

inline void idct_points(int n, float *x, const float *hs) {
	switch (n) {
	case 8: Idct1D<8>::run(x, hs); break;
	case 16: Idct1D<16>::run(x, hs); break;
	case 32: Idct1D<32>::run(x, hs); break;
	default: Idct1D<64>::run(x, hs); break;
	}
}

struct Forward {
	std::vector<double> dct[7];       // [t]: (1 << t)-point analysis matrix, row k = coefficient k (t = 3..6)
	std::vector<double> special[27];  // DctSelect 1-3, 12-17: 64 x 64, row i = coefficient at tile index i, column = sample 8 y + x
	std::vector<float> weight[27][3]; // quantisation weights per canonical index
	std::vector<int32_t> order[27];   // scan position -> canonical index (natural order)
	static bool is_special(int sel) { return (sel >= 1 && sel <= 3) || (sel >= 12 && sel <= 17); }

	Forward() {
		const float *hs = j40hip::half_secants(), *afv = j40hip::afv_basis();
		for (int t = 3; t <= 6; ++t) {
			const int n = 1 << t;
			std::vector<double> m((size_t) n * (size_t) n);
			std::vector<float> x((size_t) n);
			for (int k = 0; k < n; ++k) {   // column k of the synthesis matrix = the inverse DCT of unit vector k
				for (int i = 0; i < n; ++i) x[(size_t) i] = i == k ? 1.0f : 0.0f;
				idct_points(n, x.data(), hs);
				for (int i = 0; i < n; ++i) m[(size_t) i * (size_t) n + (size_t) k] = x[(size_t) i];
			}
			if (!invert(m, n)) throw std::runtime_error("singular DCT synthesis matrix");
			dct[t] = m;
		}
		for (int sel = 0; sel < 27; ++sel) if (is_special(sel)) {
			std::vector<double> m(64 * 64);
			float tile[j40hip::SP8_TILE], mid[j40hip::SP8_TILE], out[j40hip::SP8_TILE];   // (tiles with rows nine floats apart: special8_dev.h)
			for (int k = 0; k < 64; ++k) {
				for (int i = 0; i < j40hip::SP8_TILE; ++i) tile[i] = mid[i] = out[i] = 0.0f;
				tile[SP8(k)] = 1.0f;
				for (int l = 0; l < 8; ++l) j40hip::special8_phase0(sel, l, (const float *) tile, mid, hs, afv, false);
				for (int l = 0; l < 8; ++l) j40hip::special8_phase1(sel, l, (const float *) mid, out, hs, false);
				for (int i = 0; i < 64; ++i) m[(size_t) i * 64 + (size_t) k] = out[SP8(i)];
			}
			if (!invert(m, 64)) throw std::runtime_error("singular special synthesis matrix");
			special[sel] = m;
		}
		static const int8_t PARAM[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
		for (int sel = 0; sel < 27; ++sel) {
			const j40hip::DctSelect &d = j40hip::DCT_SELECT[sel];
			if (d.log_rows > 6 || d.log_columns > 6) continue;
			j40hip::DqMatrix dq;
			j40hip::load_dq_matrix(PARAM[sel], &dq);
			for (int c = 0; c < 3; ++c) { weight[sel][c].resize(dq.params.size()); for (size_t i = 0; i < dq.params.size(); ++i) weight[sel][c][i] = dq.params[i][(size_t) c]; }
			j40hip::natural_order(j40hip::LOG_ORDER_SIZE[d.order_idx][0], j40hip::LOG_ORDER_SIZE[d.order_idx][1], &order[sel]);
		}
	}

	// pix: rows x columns samples, row-major; coef: canonical layout (what the weights and the order index)
	void analyse(int sel, const float *pix, float *coef) const {
		const j40hip::DctSelect &d = j40hip::DCT_SELECT[sel];
		const int R = 1 << d.log_rows, C = 1 << d.log_columns;
		if (is_special(sel)) {
			const std::vector<double> &m = special[sel];
			for (int i = 0; i < 64; ++i) { double s = 0; for (int k = 0; k < 64; ++k) s += m[(size_t) i * 64 + (size_t) k] * (double) pix[k]; coef[i] = (float) s; }
			return;
		}
		const std::vector<double> &fr = dct[d.log_rows], &fc = dct[d.log_columns];
		std::vector<double> t1((size_t) R * (size_t) C);
		for (int r = 0; r < R; ++r) for (int x = 0; x < C; ++x) {   // along the columns of the block (undoes the decoder's column pass)
			double s = 0; for (int y = 0; y < R; ++y) s += fr[(size_t) r * (size_t) R + (size_t) y] * (double) pix[y * C + x];
			t1[(size_t) r * (size_t) C + (size_t) x] = s;
		}
		for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) {   // along the rows
			double s = 0; for (int x = 0; x < C; ++x) s += fc[(size_t) c * (size_t) C + (size_t) x] * t1[(size_t) r * (size_t) C + (size_t) x];
			coef[C > R ? r * C + c : c * R + r] = (float) s;
		}
	}
};

// the decoder's dequantisation of one coefficient (j40.h:7086-7094), default bias parameters
inline float dequant(float q, int c, float mult_c, float w) {
	static const float BIAS[3] = {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f, 1.0f - 0.049935103337343655f};
	if (-1.0f <= q && q <= 1.0f) q *= BIAS[c]; else q -= 0.145f / q;
	return q * (mult_c / w);
}

} // namespace synthfwd
