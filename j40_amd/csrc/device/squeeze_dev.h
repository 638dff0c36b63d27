// j40_amd/csrc/device/squeeze_dev.h -- the inverse Squeeze step (ISO/IEC 18181-1, Modular "Squeeze" transform) as
// device / host functions. The reference parses the parameters and then raises "TODO" (j40.h:3794-3812, 4518, 4530), so
// there is nothing of it to follow here: the arithmetic below is the standard's, restated from the formulas noted in
// SURVEY.md Appendix C. PARITY UNPINNED against libjxl (absent from this image); pinned by lossless round trips through
// the independent forward transform in tools/jxlsynth_modular.hpp and by the reference's decode of the same picture
// coded without Squeeze (tests/test_squeeze.py).
//
// One step joins an "average" channel (ceil(n / 2) samples along the squeezed axis) and a "residual" channel (floor(n / 2))
// into n samples: out[2k] = A, out[2k + 1] = A - diff with diff = residual[k] + tendency(out[2k - 1], avg[k], avg[k + 1]).
// The recurrence runs along the squeezed axis (out[2k - 1] feeds the next pair); lines across it are independent.
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

// the smooth "tendency" predicted from the previous output sample B, the current average a and the next average n:
// non-zero only on monotone runs, and clamped so that both reconstructed samples stay between their neighbours
J40_DEV int32_t squeeze_tendency(int32_t B, int32_t a, int32_t n) {
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

// one pair: average, residual, the output sample before the pair (the average itself for the first pair) and the next
// average (the average itself for the last one) -> the two samples. Division truncates towards zero like C's.
J40_DEV void unsqueeze_pair(int32_t avg, int32_t res, int32_t left, int32_t next_avg, int32_t *first, int32_t *second) {
	const int32_t diff = res + squeeze_tendency(left, avg, next_avg);
	const int32_t A = avg + diff / 2;
	*first = A; *second = A - diff;
}

// a whole line of `n_avg` averages and `n_res` residuals (n_res = n_avg or n_avg - 1) with element strides; writes
// n_avg + n_res samples. Samples are int16 with wrap-around, like every Modular buffer here (j40.h:3169).
template <typename SRC, typename SRC2, typename DST>
J40_DEV void unsqueeze_line(SRC avg, int32_t avg_stride, SRC2 res, int32_t res_stride, int32_t n_avg, int32_t n_res, DST out, int32_t out_stride) {
	int32_t left = 0;
	for (int32_t k = 0; k < n_res; ++k) {
		const int32_t a = avg[(int64_t) k * avg_stride];
		const int32_t next = k + 1 < n_avg ? (int32_t) avg[(int64_t) (k + 1) * avg_stride] : a;
		int32_t p, q;
		unsqueeze_pair(a, res[(int64_t) k * res_stride], k > 0 ? left : a, next, &p, &q);
		out[(int64_t) (2 * k) * out_stride] = (int16_t) p; out[(int64_t) (2 * k + 1) * out_stride] = (int16_t) q;
		left = (int16_t) q;
	}
	if (n_avg > n_res) out[(int64_t) (2 * n_res) * out_stride] = avg[(int64_t) n_res * avg_stride];
}

} // namespace j40hip
