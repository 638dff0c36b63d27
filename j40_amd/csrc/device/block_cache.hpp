// j40_amd/csrc/device/block_cache.hpp -- the recycling allocator behind the device memory cache (runtime.hip), as a class over an
// allocation backend so that its bookkeeping can be tested on the CPU (tests/hostsim: a backend with a byte budget).
//
// Device memory is recycled across frames: hipMalloc / hipFree of a 1.2 GB working set cost far more than an 8K decode. Blocks go
// back to a free list when a frame lets go of them and are handed out again to requests of similar size. A block remembers whether
// its coefficient planes are all-zero ("clean": the pixel kernels leave them that way).
//
// hipMalloc itself takes about a millisecond (13 ms for 0.8 GB) and serialises callers: a pipeline that grows to its 2800 resident
// frames (12 MB of codestream and LfGroup planes each) spent its first second inside it, the launching thread waiting behind the
// workers. Requests between 256 KB and 512 MB are therefore rounded up to a size class (eighth-of-a-power-of-two steps) and served
// from SLABS: one allocation of up to 1 GB carved into blocks of one class. A slab goes back to the device only when all of its
// blocks are idle (and the cache is over its limit, or is trimmed).
//
// Not thread-safe: the owner serialises calls (acquire() reports when it wants the backend called OUTSIDE the owner's lock, see
// runtime.hip -- the backend's allocation is the slow part).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include <algorithm>

namespace j40hip_rt {

struct BlockCacheCore {
	struct Block { void *ptr; size_t bytes; bool clean; int slab; };
	struct Slab { uint8_t *base; size_t block_bytes; int total, idle; };
	std::vector<Block> idle;          // a stack: the newest block is tried first
	std::vector<Slab> slabs;
	size_t idle_bytes = 0;

	static size_t size_class(size_t bytes) {
		bytes = (bytes + 4095) & ~(size_t) 4095;
		if (bytes < ((size_t) 256 << 10) || bytes > ((size_t) 512 << 20)) return bytes;
		size_t top = (size_t) 1 << 18;
		while (top * 2 <= bytes) top *= 2;
		const size_t step = top / 8;
		return (bytes + step - 1) / step * step;
	}
	static bool slab_class(size_t class_bytes) { return class_bytes >= ((size_t) 256 << 10) && class_bytes <= ((size_t) 512 << 20); }
	static int slab_blocks(size_t class_bytes) { return (int) std::max<size_t>(2, std::min<size_t>(64, ((size_t) 1 << 30) / class_bytes)); }

	// an idle block of class `bytes` (or up to a quarter larger), or null
	void *take(size_t bytes, size_t *got, bool *clean) {
		for (size_t i = idle.size(); i-- > 0; ) if (idle[i].bytes >= bytes && idle[i].bytes <= bytes + bytes / 4) {
			const Block b = idle[i];
			idle.erase(idle.begin() + (long) i);
			idle_bytes -= b.bytes;
			if (b.slab >= 0) --slabs[(size_t) b.slab].idle;
			*got = b.bytes; *clean = b.clean;
			return b.ptr;
		}
		return nullptr;
	}
	// a freshly allocated slab of `n` blocks of `bytes`: the first block is the caller's, the others become idle
	void adopt_slab(void *base, size_t bytes, int n) {
		size_t si = 0;
		while (si < slabs.size() && slabs[si].base) ++si;
		if (si == slabs.size()) slabs.push_back(Slab{nullptr, 0, 0, 0});
		slabs[si] = Slab{(uint8_t *) base, bytes, n, n - 1};
		for (int i = 1; i < n; ++i) idle.push_back({(uint8_t *) base + (size_t) i * bytes, bytes, false, (int) si});
		idle_bytes += bytes * (size_t) (n - 1);
	}
	// a block comes back. Returns what the caller has to free now: the block itself (*free_ptr = ptr: not kept), a whole slab
	// (*free_ptr = its base: all of it idle and the cache over `limit`), or nothing (null)
	void give(void *ptr, size_t bytes, bool clean, size_t limit, void **free_ptr) {
		*free_ptr = nullptr;
		for (size_t si = 0; si < slabs.size(); ++si) {
			Slab &sl = slabs[si];
			if (!sl.base || (uint8_t *) ptr < sl.base || (uint8_t *) ptr >= sl.base + sl.block_bytes * (size_t) sl.total) continue;
			// a slab's block: it can only stay -- or the whole slab go
			idle.push_back({ptr, sl.block_bytes, clean, (int) si});
			idle_bytes += sl.block_bytes;
			if (++sl.idle == sl.total && idle_bytes > limit) {
				size_t w = 0;
				for (size_t i = 0; i < idle.size(); ++i) if (idle[i].slab != (int) si) idle[w++] = idle[i];
				idle.resize(w);
				idle_bytes -= sl.block_bytes * (size_t) sl.total;
				*free_ptr = sl.base;
				sl.base = nullptr; sl.total = sl.idle = 0;
			}
			return;
		}
		if (idle_bytes + bytes <= limit) { idle.push_back({ptr, bytes, clean, -1}); idle_bytes += bytes; return; }
		*free_ptr = ptr;
	}
	// everything that can go: plain idle blocks and slabs all of whose blocks are idle; the idle blocks of a slab that still has
	// blocks in use stay. `out`: what the caller has to free
	void trim(std::vector<void *> *out) {
		std::vector<Block> keep;
		size_t kept = 0;
		for (const Block &b : idle) {
			if (b.slab < 0) out->push_back(b.ptr);
			else if (slabs[(size_t) b.slab].idle < slabs[(size_t) b.slab].total) { keep.push_back(b); kept += b.bytes; }
		}
		for (Slab &sl : slabs) if (sl.base && sl.idle == sl.total) { out->push_back(sl.base); sl.base = nullptr; sl.total = sl.idle = 0; }
		idle.swap(keep); idle_bytes = kept;
	}
};

} // namespace j40hip_rt
