// j40_amd/csrc/device/hostcopy.hpp -- device-to-host copies on a measured SDMA engine (hostcopy.hip says why)
#pragma once
#include <cstddef>
#include <cstdint>

namespace j40hip_rt {

// Issues the copy of `bytes` from device memory of HIP device `device` into pinned host memory (hipHostMalloc / hipHostRegister) on the
// device's fastest SDMA engine. NOT ordered with any HIP stream: the source must be complete. 0: issued, *ticket names it;
// 1: not available (no HSA, an unpinned destination, J40HIP_COPY_ENGINE=hip): the caller uses hipMemcpyAsync.
int hostcopy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes, uint64_t *ticket);
// synchronous, only when an engine has been measured already (false otherwise, and when the copy could not be issued: use hipMemcpy);
// the source must be complete
bool hostcopy_d2h_sync(int device, void *dst_host, const void *src_dev, size_t bytes);
bool hostcopy_ready(int device);
// an upload from pinned host memory, synchronous (the thread sleeps while it runs), on an SDMA engine kept apart from the copies back;
// false: not available (no engine measured yet -- a pipeline does that -- or no HSA): the caller's hipMemcpyAsync
bool hostcopy_h2d_sync(int device, void *dst_dev, const void *src_host, size_t bytes);
int hostcopy_state(uint64_t ticket);            // 0 running, 1 done, -1 failed
bool hostcopy_wait(uint64_t ticket);            // sleeps until it is through; false: the copy failed
void hostcopy_release(int device, uint64_t ticket);   // the ticket's signal goes back to the pool (after done / failed)
// the engine chosen for `device` (-1: none), the measured device-to-host GB/s per engine (0: not measured) and the runtime's masks
// {free, recommended, set aside for uploads} (three words); measures on first use
int hostcopy_engine(int device, double *gbps16, uint32_t *masks2);
void hostcopy_shutdown();

}  // namespace j40hip_rt
