// j40_amd/csrc/device/k2_iter_dev.h -- how a persistent workgroup of the pixel kernels (kernels.hip: k_vardct_dct, k_vardct_special,
// k_vardct_large in their batch-wide form) walks its run of tiles. A launch covers one class of transforms over every frame of the
// batch, cut into tiles of `per_wg` varblocks; tile_prefix[f] = tiles of the frames before frame f. Workgroup `block` of `grid` takes
// a contiguous run of tiles: one search for its first tile's frame, then it walks along. A run stays inside one frame for hundreds
// of tiles, so the frame's list, count and output are fetched when the run ENTERS a frame (`entered`), not per tile -- per tile the
// bind is arithmetic on registers (round 4: an instrumented build put a third of a tile's time into its prologue, a chain of
// dependent loads of which these were the head). Compiled for the CPU as well: tests/hostsim walks the runs of all workgroups and
// checks that every tile is taken exactly once, with the right frame, list and first varblock.
#pragma once
#include "plan.h"

namespace j40hip {

#ifdef __HIPCC__
#define J40_K2_DEV __device__ __forceinline__
// wave-uniform values said to be so (v_readfirstlane): they then live in scalar registers across the tiles of a frame
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ size_t uni(size_t v) { return (size_t) uni((uint32_t) v) | (size_t) uni((uint32_t) (v >> 32)) << 32; }
template <typename T> __device__ __forceinline__ T *uni(T *p) { return (T *) uni((size_t) p); }
#else
#define J40_K2_DEV static inline
template <typename T> static inline T uni(T v) { return v; }
#endif

struct K2Iter { int32_t tile, tile_end, frame, frame_first, frame_end; };   // tiles [frame_first, frame_end) are `frame`'s

J40_K2_DEV K2Iter k2_run_begin(const int32_t *tile_prefix, int32_t nframes, int32_t block, int32_t grid) {
	K2Iter it = {0, 1, 0, 0, 0};
	const int32_t total = tile_prefix[nframes], chunk = (total + grid - 1) / grid;
	it.tile = block * chunk; it.tile_end = total < it.tile + chunk ? total : it.tile + chunk;
	int32_t lo = 0, hi = nframes - 1;   // last frame whose prefix <= tile
	while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (tile_prefix[mid] <= it.tile) lo = mid; else hi = mid - 1; }
	it.frame = lo;
	return it;
}

// the next tile of the run: its frame (whose list, count, output replace the caller's when the run enters it) and `first`, the
// tile's first varblock in the list. Returns false when the run is done.
J40_K2_DEV bool k2_run_bind(K2Iter &it, const K2Frame *batch, const int32_t *tile_prefix, int32_t class_a, int32_t class_b, int32_t per_wg,
		const DevVarblock *&list, int32_t &count, uint8_t *&rgba, size_t &stride, int32_t &frame, int32_t &first, bool &entered) {
	if (it.tile >= it.tile_end) return false;
	entered = it.tile >= it.frame_end;   // (frame_end starts at 0: the first tile always enters)
	if (entered) {
		while (tile_prefix[it.frame + 1] <= it.tile) ++it.frame;   // (frames without tiles of this class)
		it.frame = uni(it.frame);
		it.frame_first = uni(tile_prefix[it.frame]); it.frame_end = uni(tile_prefix[it.frame + 1]);
		const K2Frame &fr = batch[it.frame];
		const int32_t a = fr.class_start[class_a];
		list = uni(fr.sorted + a); count = uni(fr.class_start[class_b] - a); rgba = uni(fr.rgba); stride = uni(fr.stride);
	}
	frame = it.frame;
	first = (it.tile - it.frame_first) * per_wg;
	++it.tile;
	return true;
}

} // namespace j40hip
