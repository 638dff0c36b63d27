// j40_amd/csrc/device/props_dev.h -- the MA-tree properties of a sample that are functions of its neighbours (properties 4..14 of
// ISO/IEC 18181-1 table H.4; the reference evaluates them inside j40__modular_channel, j40.h:4181-4200). ONE statement of them for
// every decoder of Modular samples in the tree: the host decoder (modular.cpp), the general section kernel (modular_dev.h) and the
// LfGroup lane decoders (lf_lanes_dev.h, lf_rows_dev.h). Neighbours arrive with the image-edge fallbacks already applied
// (j40.h:3965-3990); `x` is the sample's column.
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

// magnitudes (4, 5) and copies (6, 7) of N and W; 8: what W misses of its own gradient-style estimate from WW, NW, NWW (W itself in
// the first column); 9: the gradient estimate W + N - NW; 10..14: first differences across the neighbourhood
J40_DEVM int32_t neighbour_property(int32_t prop, int32_t x, int32_t w, int32_t n, int32_t nw, int32_t ne, int32_t nn, int32_t ww, int32_t nww) {
	switch (prop) {
	case 4: return n < 0 ? -n : n;
	case 5: return w < 0 ? -w : w;
	case 6: return n;
	case 7: return w;
	case 8: return x > 0 ? w - (ww + nw - nww) : w;
	case 9: return w + n - nw;
	case 10: return w - nw;
	case 11: return nw - n;
	case 12: return n - ne;
	case 13: return n - nn;
	default: return w - ww;   // 14
	}
}

} // namespace j40hip
