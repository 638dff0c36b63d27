// j40_amd/csrc/device/runtime_shared.hpp -- what runtime.hip shares with async.hip: the per-device cache of device memory blocks
// and the one-time upload of the kernels' constant tables
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <hip/hip_runtime.h>

extern "C" void j40hip_cache_counters(uint64_t *out);   // runtime.hip: what the device memory cache did (J40HIP_ASYNC_TIMING)

namespace j40hip_rt {

// a block of at least `bytes` (rounded up to 4 KB) from the device's cache or from hipMalloc; null when the device is out of memory
void *cache_acquire(int device, size_t bytes, size_t *got, bool *clean);
// gives a block back; nothing may still be running on it
void cache_release(int device, void *ptr, size_t bytes, bool clean);
void cache_trim(int device);
bool ensure_constant_tables(int device);

// host-side staging of the plan: every array lands in one blob at a 256-byte aligned offset, one copy moves it. The blob lives
// in PINNED host memory owned by the calling thread (grown on demand, reused by that thread's next upload), so the copy is a
// true asynchronous DMA that overlaps the kernels of other frames; j40hip_thread_release gives it back.
struct PinnedStage {
	uint8_t *ptr = nullptr; size_t cap = 0;
	bool reserve(size_t n, size_t keep) {
		if (n <= cap) return true;
		size_t want = std::max(n + n / 4, (size_t) 1 << 20);
		void *q = nullptr;
		if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); return false; }
		if (keep) memcpy(q, ptr, keep);
		if (ptr) (void) hipHostFree(ptr);
		ptr = (uint8_t *) q; cap = want;
		return true;
	}
	void release() { if (ptr) (void) hipHostFree(ptr); ptr = nullptr; cap = 0; }
};

} // namespace j40hip_rt
