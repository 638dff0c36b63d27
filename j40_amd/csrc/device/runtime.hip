// j40_amd/csrc/device/runtime.hip -- device half of the thin C-ABI (include/j40hip.h): builds the
// flat frame plan in HBM, launches the hot-path kernels on the caller's stream, reads status back.
//
// No CPU fallback lives here: without a HIP device every entry point returns "!gpu".
#include <hip/hip_runtime.h>
#include <deque>
#include <condition_variable>
#include <chrono>
#include <thread>
#include <algorithm>
#include <mutex>
#include <atomic>
#include <cmath>
#include <unistd.h>
#include "../capi.hpp"
#include "../tables.hpp"
#include "../plan_build.hpp"
#include "kernels.h"
#include "runtime_shared.hpp"
#include "hostcopy.hpp"
#include "block_cache.hpp"
#include "async.hpp"

using namespace j40hip;
using namespace j40hip_rt;

namespace {

constexpr uint32_t ERR_GPU = ('!' << 24) | ('g' << 16) | ('p' << 8) | 'u';
constexpr uint32_t ERR_MEM = ('!' << 24) | ('m' << 16) | ('e' << 8) | 'm';
constexpr uint32_t ERR4(char a, char b, char c, char d) { return ((uint32_t) (uint8_t) a << 24) | ((uint32_t) (uint8_t) b << 16) | ((uint32_t) (uint8_t) c << 8) | (uint32_t) (uint8_t) d; }

// no exception crosses the C ABI: a parse error keeps its code, anything else (std::bad_alloc from a vector, ...) is "!mem"
template <typename F> uint32_t guarded(F f) {
	try { return f(); }
	catch (const DecodeError &e) { return e.code; }
	catch (const std::exception &) { return ERR_MEM; }
}

struct DeviceBuffer {
	void *ptr = nullptr; size_t bytes = 0;
	bool alloc(size_t n);
	void release() { if (ptr) (void) hipFree(ptr); ptr = nullptr; }
};

// The device memory cache: one BlockCacheCore (block_cache.hpp: free list, size classes, slabs) per device behind one mutex; the
// slow part -- hipMalloc / hipFree -- happens outside the lock.
std::mutex g_cache_mutex;
std::condition_variable g_cache_cv;      // a slab of some class has been adopted (or its allocation failed)
BlockCacheCore g_cache[16];
std::vector<size_t> g_slab_pending[16];  // size classes whose slab some thread is allocating right now
// Upper bound on what the cache of ONE device keeps idle, per process (J40HIP_CACHE_GB overrides; 0 disables recycling and slabs).
// When an allocation fails the cache is emptied and the allocation tried again (cache_trim), so idle blocks never turn into a
// spurious "!gpu". Default: 60 % of the device's memory -- a pipeline returns the working sets of a whole batch at once (256 8K
// frames: 54 GB), and hipFree / hipMalloc of such blocks cost tens of milliseconds each and synchronise the device. Processes that
// share a device (several ranks on one GPU, multi-tenant serving) each keep up to this much: set J40HIP_CACHE_GB to the device's
// memory divided by their number, less what the frames in flight need.
// what the cache did, for J40HIP_ASYNC_TIMING (j40hip_cache_counters): calls, and the milliseconds spent waiting for the lock, searching
// the free list, inside hipMalloc and inside hipFree
struct CacheCounters { std::atomic<uint64_t> acquires{0}, hits{0}, slab_mallocs{0}, plain_mallocs{0}, frees{0}, lock_us{0}, take_us{0}, malloc_us{0}, free_us{0}, idle_blocks{0}; };
CacheCounters g_cache_counters;
inline uint64_t cache_us() { return (uint64_t) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::mutex g_limit_mutex;
size_t g_limit[16]; bool g_limit_known[16];
size_t cache_limit_bytes(int device) {   // (never called with g_cache_mutex held: hipMemGetInfo takes its time)
	if (device < 0 || device >= 16) return 0;
	{ std::lock_guard<std::mutex> lock(g_limit_mutex); if (g_limit_known[device]) return g_limit[device]; }
	size_t limit = (size_t) 48 << 30;
	if (const char *e = getenv("J40HIP_CACHE_GB")) limit = (size_t) std::max(0, atoi(e)) << 30;
	else {
		int cur = -1; size_t free_b = 0, total_b = 0;
		const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = total_b / 10 * 6; else (void) hipGetLastError();
		if (switched) (void) hipSetDevice(cur);
	}
	std::lock_guard<std::mutex> lock(g_limit_mutex);
	g_limit[device] = limit; g_limit_known[device] = true;
	return limit;
}

} // namespace

// frees every cached block of `device` that can be freed (they are idle by construction: blocks enter the cache after a device
// synchronisation); the idle blocks of a slab that still has blocks in use stay
void j40hip_rt::cache_trim(int device) {
	if (device < 0 || device >= 16) return;
	std::vector<void *> gone;
	{ std::lock_guard<std::mutex> lock(g_cache_mutex); g_cache[device].trim(&gone); }
	for (void *q : gone) (void) hipFree(q);
}

void *j40hip_rt::cache_acquire(int device, size_t bytes, size_t *got, bool *clean) {
	bytes = BlockCacheCore::size_class(bytes);
	const bool cached = device >= 0 && device < 16;
	const size_t limit = cached ? cache_limit_bytes(device) : 0;
	bool slab = cached && limit > 0 && BlockCacheCore::slab_class(bytes);
	if (cached) {
		// One thread per size class allocates a slab; whoever else misses the class meanwhile waits for it and looks again (when a
		// pipeline starts, every worker misses the empty cache at the same moment: each of them used to allocate a slab of its own)
		const uint64_t tl0 = cache_us();
		std::unique_lock<std::mutex> lock(g_cache_mutex);
		const uint64_t tl1 = cache_us();
		g_cache_counters.lock_us += tl1 - tl0; ++g_cache_counters.acquires;
		for (;;) {
			const uint64_t tt0 = cache_us();
			void *q = g_cache[device].take(bytes, got, clean);
			g_cache_counters.take_us += cache_us() - tt0; g_cache_counters.idle_blocks = g_cache[device].idle.size();
			if (q) { ++g_cache_counters.hits; return q; }
			std::vector<size_t> &pend = g_slab_pending[device];
			if (!slab || std::find(pend.begin(), pend.end(), bytes) == pend.end()) { if (slab) pend.push_back(bytes); break; }
			g_cache_cv.wait(lock);
		}
	}
	void *p = nullptr;
	if (slab) {
		// a slab is up to 64 blocks / 1 GB; smaller when the device is short of memory or the cache near its limit (its idle blocks count)
		int n = BlockCacheCore::slab_blocks(bytes);
		size_t free_b = 0, total_b = 0, idle_b = 0;
		{ std::lock_guard<std::mutex> lock(g_cache_mutex); idle_b = g_cache[device].idle_bytes; }
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); free_b = 0; }
		while (n > 1 && (bytes * (size_t) n > free_b / 4 || idle_b + bytes * (size_t) (n - 1) > limit)) n /= 2;
		const uint64_t tm0 = cache_us();
		if (n > 1 && hipMalloc(&p, bytes * (size_t) n) != hipSuccess) { (void) hipGetLastError(); p = nullptr; }
		g_cache_counters.malloc_us += cache_us() - tm0; ++g_cache_counters.slab_mallocs;
		{
			std::lock_guard<std::mutex> lock(g_cache_mutex);
			if (p) g_cache[device].adopt_slab(p, bytes, n);
			std::vector<size_t> &pend = g_slab_pending[device];
			pend.erase(std::find(pend.begin(), pend.end(), bytes));
		}
		g_cache_cv.notify_all();
		if (p) { *got = bytes; *clean = false; return p; }
	}
	const uint64_t tm0 = cache_us();
	const hipError_t first_try = hipMalloc(&p, bytes);
	g_cache_counters.malloc_us += cache_us() - tm0; ++g_cache_counters.plain_mallocs;
	if (first_try != hipSuccess) {
		// out of device memory while blocks sit idle in the cache: give them back and try once more
		(void) hipGetLastError();
		cache_trim(device);
		if (hipMalloc(&p, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	}
	*got = bytes; *clean = false;
	return p;
}

void j40hip_rt::cache_release(int device, void *ptr, size_t bytes, bool clean) {
	if (!ptr) return;
	void *gone = ptr;
	if (device >= 0 && device < 16) {
		const size_t limit = cache_limit_bytes(device);
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		g_cache[device].give(ptr, bytes, clean, limit, &gone);
	}
	if (gone) { const uint64_t tf0 = cache_us(); (void) hipFree(gone); g_cache_counters.free_us += cache_us() - tf0; ++g_cache_counters.frees; }
}

// out[10]: acquires, hits, slab allocations, plain allocations, frees, then microseconds: lock, free-list search, hipMalloc, hipFree; idle blocks now
extern "C" __attribute__((visibility("default"))) void j40hip_cache_counters(uint64_t *out) {
	const CacheCounters &c = g_cache_counters;
	out[0] = c.acquires; out[1] = c.hits; out[2] = c.slab_mallocs; out[3] = c.plain_mallocs; out[4] = c.frees;
	out[5] = c.lock_us; out[6] = c.take_us; out[7] = c.malloc_us; out[8] = c.free_us; out[9] = c.idle_blocks;
}

// ---- pinned host memory for pixels that go back to the caller (the public API's image planes): pinning 133 MB takes tens of
// milliseconds (0.2 s for 133 MB measured, as long again to unpin), so planes are recycled by size across images.
// What sits idle is bounded three ways (a drop-in caller never calls j40hip_shutdown, and pinned memory cannot be swapped):
//   * J40HIP_PINNED_POOL_GB (default: the smaller of 32 GB -- 240 planes of an 8K image; with 128 callers and a 16 GB bound every
//     j40_free beyond the 123rd plane unpinned it and the next image pinned a new one -- and a quarter of the machine's memory;
//     0: nothing kept);
//   * a plane that does not fit is made room for by unpinning the planes that have been idle longest (a process that moves on to
//     another image size does not keep the old size's planes and pin / unpin every image of the new one);
//   * planes idle for more than J40HIP_PINNED_IDLE_S seconds (default 30) are unpinned at the library's next acquire or release.
namespace {
struct PinnedIdle { void *ptr; size_t bytes; double since; };
std::mutex g_pinned_mutex;
std::vector<PinnedIdle> g_pinned_idle;   // oldest first
size_t g_pinned_idle_bytes = 0;
size_t pinned_limit() {
	static const size_t v = [] {
		if (const char *e = getenv("J40HIP_PINNED_POOL_GB")) return (size_t) std::max(0, atoi(e)) << 30;
		const long pages = sysconf(_SC_PHYS_PAGES), page = sysconf(_SC_PAGESIZE);
		const size_t ram = pages > 0 && page > 0 ? (size_t) pages * (size_t) page : (size_t) 128 << 30;
		return std::min((size_t) 32 << 30, ram / 4);
	}();
	return v;
}
double pinned_idle_seconds() { static const double v = [] { const char *e = getenv("J40HIP_PINNED_IDLE_S"); return e && atof(e) > 0 ? atof(e) : 30.0; }(); return v; }
double pinned_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// (under g_pinned_mutex) moves to `gone`: planes idle for too long, then the oldest ones until `incoming` more bytes fit the bound
void pinned_make_room(size_t incoming, std::vector<void *> *gone) {
	const double now = pinned_now(), keep = pinned_idle_seconds();
	size_t n = 0;
	while (n < g_pinned_idle.size() && (now - g_pinned_idle[n].since > keep || g_pinned_idle_bytes + incoming > pinned_limit())) {
		gone->push_back(g_pinned_idle[n].ptr); g_pinned_idle_bytes -= g_pinned_idle[n].bytes; ++n;
	}
	g_pinned_idle.erase(g_pinned_idle.begin(), g_pinned_idle.begin() + (long) n);
}
}
extern "C" __attribute__((visibility("default"))) void *j40hip_pinned_acquire(size_t bytes) {
	bytes = (bytes + 4095) & ~(size_t) 4095;
	std::vector<void *> gone;
	void *q = nullptr;
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		for (size_t i = g_pinned_idle.size(); i-- > 0; ) if (g_pinned_idle[i].bytes == bytes) {   // the most recently used plane of this size
			q = g_pinned_idle[i].ptr;
			g_pinned_idle.erase(g_pinned_idle.begin() + (long) i); g_pinned_idle_bytes -= bytes;
			break;
		}
		pinned_make_room(q ? 0 : bytes, &gone);   // (a miss: the plane pinned now will come back to the pool)
	}
	for (void *g : gone) (void) hipHostFree(g);
	if (q) return q;
	if (hipHostMalloc(&q, bytes ? bytes : 4096, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	return q;
}
extern "C" __attribute__((visibility("default"))) void j40hip_pinned_release(void *ptr, size_t bytes) {
	if (!ptr) return;
	bytes = (bytes + 4095) & ~(size_t) 4095;
	std::vector<void *> gone;
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		pinned_make_room(bytes, &gone);
		if (g_pinned_idle_bytes + bytes <= pinned_limit()) { g_pinned_idle.push_back({ptr, bytes, pinned_now()}); g_pinned_idle_bytes += bytes; ptr = nullptr; }
	}
	for (void *g : gone) (void) hipHostFree(g);
	if (ptr) (void) hipHostFree(ptr);
}
// (what the pool holds: tests/test_api_threads.py)
extern "C" __attribute__((visibility("default"))) void j40hip_pinned_pool_stats(uint64_t *idle_bytes, uint64_t *idle_planes, uint64_t *limit_bytes) {
	std::lock_guard<std::mutex> lock(g_pinned_mutex);
	if (idle_bytes) *idle_bytes = g_pinned_idle_bytes;
	if (idle_planes) *idle_planes = g_pinned_idle.size();
	if (limit_bytes) *limit_bytes = pinned_limit();
}
static void pinned_trim() {
	std::vector<PinnedIdle> gone;
	{ std::lock_guard<std::mutex> lock(g_pinned_mutex); gone.swap(g_pinned_idle); g_pinned_idle_bytes = 0; }
	for (auto &b : gone) (void) hipHostFree(b.ptr);
}

namespace {

bool DeviceBuffer::alloc(size_t n) {
	bytes = n;
	if (hipMalloc(&ptr, n ? n : 16) == hipSuccess) return true;
	(void) hipGetLastError();
	int device = 0;
	if (hipGetDevice(&device) == hipSuccess) cache_trim(device);
	return hipMalloc(&ptr, n ? n : 16) == hipSuccess;
}

thread_local PinnedStage t_stage;   // (no destructor: at process exit the runtime may be gone before the thread's storage)

struct Stager {
	size_t size = 0; bool ok = true;
	// deferred: put() notes the copy, flush() makes them all, shared out by bytes over a few threads (an 8K frame's plan is 30 MB: a
	// millisecond of one core's memcpy on the single-image path); the sources have to live until then
	bool deferred = false;
	struct Copy { size_t off; const uint8_t *src; size_t bytes; };
	std::vector<Copy> copies;
	template <typename T> size_t put(const T *src, size_t n) {
		const size_t off = (size + 255) & ~(size_t) 255, end = off + sizeof(T) * n + 16;
		if (!ok || !t_stage.reserve(end, size)) { ok = false; return 0; }
		if (n) { if (deferred) copies.push_back({off, (const uint8_t *) src, sizeof(T) * n}); else memcpy(t_stage.ptr + off, src, sizeof(T) * n); }
		size = end;   // (deferred: a grown buffer keeps the bytes below `size` -- nothing of the noted copies is there yet, and nothing needs to be)
		return off;
	}
	void flush(int threads) {
		if (!deferred || !ok) { copies.clear(); return; }
		size_t total = 0;
		for (const Copy &c : copies) total += c.bytes;
		const int n = total < ((size_t) 4 << 20) ? 1 : std::max(1, std::min(threads, 8));
		uint8_t *base = t_stage.ptr;
		auto work = [&](int t) {
			const size_t lo = total / (size_t) n * (size_t) t, hi = t + 1 == n ? total : total / (size_t) n * (size_t) (t + 1);
			size_t at = 0;
			for (const Copy &c : copies) {
				const size_t a = std::max(lo, at), b = std::min(hi, at + c.bytes);
				if (a < b) memcpy(base + c.off + (a - at), c.src + (a - at), b - a);
				at += c.bytes;
			}
		};
		std::vector<std::thread> pool;
		try { for (int t = 1; t < n; ++t) pool.emplace_back(work, t); } catch (const std::exception &) {}
		const int started = (int) pool.size() + 1;
		work(0);
		for (int t = started; t < n; ++t) work(t);   // (threads that could not be had: their share here)
		for (auto &th : pool) th.join();
		copies.clear();
	}
	size_t reserve(size_t bytes) {   // room in the device block that nothing is copied into (the bytes staged for it are whatever is there)
		const size_t off = (size + 255) & ~(size_t) 255, end = off + bytes + 16;
		if (!ok || !t_stage.reserve(end, size)) { ok = false; return 0; }
		size = end;
		return off;
	}
	const uint8_t *data() const { return t_stage.ptr; }
};

// the constant tables of the pixel kernels go up once per device (they never change)
std::mutex g_const_mutex;
bool g_const_done[16];
} // namespace
bool j40hip_rt::ensure_constant_tables(int device) {
	if (device < 0 || device >= 16) return false;
	std::lock_guard<std::mutex> lock(g_const_mutex);
	if (g_const_done[device]) return true;
	upload_constant_tables(half_secants(), afv_basis(), srgb_u8_thresholds(), nullptr);
	upload_lf_tail_tables(half_secants(), lf2llf_scales(), nullptr);
	if (hipStreamSynchronize(nullptr) != hipSuccess) return false;
	return g_const_done[device] = true;
}

struct j40hip_device_state {
	int device = 0;
	std::vector<DeviceBuffer> buffers;
	void *plan_block = nullptr, *work_block = nullptr;   // VarDCT frames: the uploaded plan and the working set (recycled, see cache_acquire)
	size_t plan_block_bytes = 0, work_block_bytes = 0;
	bool force_dense = false;                             // dense coefficient planes although the frame has one pass (after ERR_EVOF)
	size_t num_blocks = 0;                                // entries of plan.block_events / 4
	DevPlan plan;
	bool is_modular = false;
	int64_t first_group = 0, num_groups = 0;       // range decoded by this process
	std::vector<DevVarblock> vb_sorted;             // by DctSelect; host copy, fetched from the device on demand (host_vb_sorted)
	size_t vb_count = 0;
	int32_t class_start[28];
	DevVarblock *d_vb_sorted = nullptr;
	// sharded decode (j40hip_frame_set_group_range): the varblocks of the selected groups, same layout as vb_sorted
	DevVarblock *d_vb_range = nullptr; int32_t range_class_start[28]; size_t vb_range_capacity = 0;
	float *d_large_scratch = nullptr;
	size_t coeff_floats = 0;
	int32_t total_sections = 0;
	HfLaunchInfo hf;
	// Modular frames
	DevModPlan mod;
	int32_t mod_sections = 0, mod_passes = 1, mod_sections_per_pass = 0;   // sections = LfGlobal's (0 or 1) + passes * per_pass
	bool mod_local_rcts = false;
	bool has_trailers = false;           // VarDCT frame whose sections go on with the extra channels' Modular sub-image
	bool idle = false;                   // j40hip_frame_mark_idle: nothing is pending on this frame's memory, freeing it needs no device-wide wait
	bool trailers_pending = false;       // ... decoded by a batch since: j40hip_frame_status validates the sub-images before it reports
	ModLaunchInfo mod_info = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	std::vector<uint32_t> mod_section_offsets;
	// group: the group whose section's sub-image the op belongs to (mod_sub_ops; a ranged decode skips the ops of groups it did not decode), -1: the frame's
	struct ModOp { int kind; int16_t *a, *b, *c; const int16_t *src, *aux; size_t n; int32_t p0, p1, p2, p3, p4, p5; int16_t *const *dst_list; const int8_t *wpp; int32_t group; };
	std::vector<ModOp> mod_ops;          // inverse transforms of the frame, in execution order
	std::vector<ModOp> mod_sub_ops;      // before them: inverse transforms of the sections' own sub-images and their paste (kind 3)
	std::vector<int16_t *> final_planes; // channel list after the inverse transforms
	std::vector<int32_t> final_w, final_h;
	int32_t alpha_channel = -1;
	int32_t *pal_wp_scratch = nullptr;
	uint32_t *mod_extra_status = nullptr;
	std::vector<uint32_t> status_host;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	// the single-image path's two phases (decode_two_phase): the groups by decreasing section size, the `two_k` first of them decoded
	// beside the rest with their block_events entries in a table of their own; -1: not looked at yet, 0: not for this frame
	int32_t two_k = -1;
	uint32_t *d_two_order = nullptr, *d_two_shadow = nullptr;   // (one block of the device memory cache: two_block)
	void *two_block = nullptr; size_t two_block_bytes = 0;
	std::vector<uint32_t> two_order;
	hipStream_t two_stream = nullptr; hipEvent_t two_ev[3] = {nullptr, nullptr, nullptr};
	// the restoration filters (decode_impl, restore_*): made at the first decode that runs them, kept with the frame
	float *d_xyb = nullptr, *d_xyb_tmp = nullptr, *d_sigma = nullptr; int16_t *d_sharp = nullptr;
	const float *d_restored = nullptr;   // where the last decode's filtered planes lie (d_xyb or d_xyb_tmp)
	uint32_t restore_err = 0;            // the last decode's "gab0" / "epf0" / "shrp" (reported behind the sections' codes)
	int restore_ran = 0;                 // the mode the last decode ran the filters in (0: it did not)
	float restore_ms = 0;

	template <typename T> T *upload(const T *src, size_t n, hipStream_t s, bool &ok) {
		DeviceBuffer b;
		if (!b.alloc(sizeof(T) * n)) { ok = false; return nullptr; }
		buffers.push_back(b);
		if (n && hipMemcpyAsync(b.ptr, src, sizeof(T) * n, hipMemcpyHostToDevice, s) != hipSuccess) ok = false;
		return (T *) b.ptr;
	}
	template <typename T> T *scratch(size_t n, bool &ok) {
		DeviceBuffer b;
		if (!b.alloc(sizeof(T) * n)) { ok = false; return nullptr; }
		buffers.push_back(b);
		return (T *) b.ptr;
	}
};

extern "C" void j40hip_release_device(j40hip_frame *f) {
	if (!f || !f->dev) return;
	(void) hipSetDevice(f->dev->device);
	if (f->dev->plan_block || f->dev->work_block) {
		if (!f->dev->idle) (void) hipDeviceSynchronize();   // nothing may still be running on memory that is about to be handed to another frame
		cache_release(f->dev->device, f->dev->plan_block, f->dev->plan_block_bytes, false);
		cache_release(f->dev->device, f->dev->work_block, f->dev->work_block_bytes, false);
	}
	if (f->dev->two_block) {
		if (!f->dev->idle) (void) hipDeviceSynchronize();
		cache_release(f->dev->device, f->dev->two_block, f->dev->two_block_bytes, false);
	}
	for (auto &b : f->dev->buffers) b.release();
	for (auto &e : f->dev->ev) if (e) (void) hipEventDestroy(e);
	delete f->dev;
	f->dev = nullptr;
}

extern "C" int j40hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

static uint32_t upload_modular(j40hip_frame *h, int device) {
	HostModPlan hp;
	if (uint32_t e = build_modular_plan(h->frame, h->cs, h->cs_size, &hp)) return e;
	j40hip_device_state *st = new j40hip_device_state();
	h->dev = st; st->device = device; st->is_modular = true;
	st->first_group = 0; st->num_groups = h->frame.fh.num_groups;   // (j40hip_frame_set_group_range narrows it)
	hipStream_t s = nullptr;
	bool ok = true;
	DevModPlan &plan = st->mod;
	memset(&plan, 0, sizeof plan);
	plan.frame = st->upload(&hp.frame, 1, s, ok);
	plan.codestream = st->upload(hp.codestream.data(), hp.codestream.size(), s, ok);
	plan.pool_u8 = st->upload(hp.pool_u8.data(), hp.pool_u8.size(), s, ok);
	plan.pool_i32 = st->upload(hp.pool_i32.data(), hp.pool_i32.size(), s, ok);
	plan.pool_u64 = st->upload(hp.pool_u64.data(), hp.pool_u64.size(), s, ok);
	plan.clusters = st->upload(hp.clusters.data(), hp.clusters.size(), s, ok);
	plan.spec = st->upload(hp.specs.data(), hp.specs.size(), s, ok);
	plan.tree = st->upload(hp.tree.data(), hp.tree.size(), s, ok);
	plan.sections = st->upload(hp.sections.data(), hp.sections.size(), s, ok);
	if (!hp.coop_trees.empty()) plan.coop_trees = st->upload(hp.coop_trees.data(), hp.coop_trees.size(), s, ok);
	st->mod_local_rcts = !hp.local_rct.empty();
	if (st->mod_local_rcts) plan.local_rct = st->upload(hp.local_rct.data(), hp.local_rct.size(), s, ok);
	st->mod_sections = (int32_t) hp.sections.size(); st->mod_passes = hp.num_passes; st->mod_sections_per_pass = hp.sections_per_pass;
	st->mod_info = {hp.max_tree_nodes, hp.max_num_dist, hp.max_clusters, hp.max_table_bytes, hp.frame.max_width, hp.any_wp ? 1 : 0, hp.coop_width, hp.coop_sections + hp.split_sections == (int32_t) hp.sections.size(), hp.quad_sections, hp.quad_spec, hp.quad_width, hp.coop_sections, hp.quad_sections ? hp.specs[(size_t) hp.quad_spec].table_span : 0u, hp.split_sections, hp.split_width, hp.split_channels};
	for (const DevModSection &sec : hp.sections) st->mod_section_offsets.push_back(sec.byte_off);
	const int32_t nch = hp.frame.num_channels;
	struct Ref { int16_t *p; int32_t w, h; };
	std::vector<Ref> planes;
	{
		// the coded channels: one allocation, each plane at a 256-byte aligned offset (a squeezed 16384 x 16384 frame has 70+ of them)
		std::vector<size_t> at((size_t) nch); size_t total = 0;
		for (int32_t c = 0; c < nch; ++c) { at[(size_t) c] = total; total += ((size_t) std::max(hp.plane_w[(size_t) c], 0) * (size_t) std::max(hp.plane_h[(size_t) c], 0) * 2 + 2 + 255) & ~(size_t) 255; }
		uint8_t *blockp = (uint8_t *) st->scratch<uint8_t>(total ? total : 256, ok);
		std::vector<DevPlaneRef> refs((size_t) nch);
		for (int32_t c = 0; c < nch; ++c) {
			int16_t *p = blockp ? (int16_t *) (blockp + at[(size_t) c]) : nullptr;
			refs[(size_t) c] = DevPlaneRef{p, hp.plane_w[(size_t) c], hp.plane_h[(size_t) c], hp.plane_meta[(size_t) c], 0};
			planes.push_back({p, hp.plane_w[(size_t) c], hp.plane_h[(size_t) c]});
		}
		plan.planes = st->upload(refs.data(), refs.size(), s, ok);
		if (!hp.chan_rects.empty()) plan.chan_rects = st->upload(hp.chan_rects.data(), hp.chan_rects.size(), s, ok);
	}
	bool palette_wp = false;
	for (const Transform &t : hp.transforms) palette_wp |= t.kind == Transform::PALETTE && t.nb_deltas > 0 && t.d_pred == 6;
	if (hp.frame.tree_uses_wp) plan.wp_scratch = st->scratch<int32_t>((size_t) hp.sections.size() * (size_t) (2 * hp.frame.max_width * 5) + 16, ok);
	if (palette_wp) st->pal_wp_scratch = st->scratch<int32_t>((size_t) 2 * (size_t) hp.frame.width * 5 + 16, ok);
	plan.lz_window_size = hp.lz_window_size;
	if (hp.lz_window_size) plan.lz_window = st->scratch<int32_t>((size_t) hp.sections.size() * hp.lz_window_size, ok);
	plan.status = st->scratch<uint32_t>(hp.sections.size() + 1, ok);
	// sections decoded in two passes (modular_split.hip): their residual tokens -- also their LZ77 windows -- and the passes' notes
	if (hp.split_sections) { plan.residuals = st->scratch<int32_t>(hp.split_samples + 64, ok); plan.split_state = st->scratch<uint32_t>(3 * hp.sections.size() + 4, ok); }
	st->mod_extra_status = const_cast<uint32_t *>(plan.status) + hp.sections.size();
	st->total_sections = (int32_t) hp.sections.size();

	// inverse transforms, last to first (j40.h:4513-4521), resolved to plane pointers now: of the frame, and before that of the
	// sub-images of the sections that list a palette of their own (undone there, then pasted over the section's rectangle)
	static const uint8_t PERM[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
	auto schedule = [&](std::vector<Ref> &planes, const std::vector<Transform> &trs, const int8_t *wpb, std::vector<j40hip_device_state::ModOp> &ops, int32_t group) {
		const size_t ops_before = ops.size();
		for (size_t ti = trs.size(); ti-- > 0; ) {
			const Transform &t = trs[ti];
			if (t.kind == Transform::RCT) {
				j40hip_device_state::ModOp op; memset(&op, 0, sizeof op);
				Ref c[3] = {planes[(size_t) t.begin_c], planes[(size_t) t.begin_c + 1], planes[(size_t) t.begin_c + 2]};
				op.kind = 0; op.a = c[0].p; op.b = c[1].p; op.c = c[2].p; op.n = (size_t) c[0].w * (size_t) c[0].h; op.p0 = t.rct_type % 7;
				ops.push_back(op);
				for (int i = 0; i < 3; ++i) planes[(size_t) (t.begin_c + PERM[t.rct_type / 7][i])] = c[i];
			} else if (t.kind == Transform::PALETTE) {
				const int32_t first = t.begin_c + 1;
				const Ref idx = planes[(size_t) first], pal = planes[0];
				const size_t n = (size_t) idx.w * (size_t) idx.h;
				std::vector<Ref> outs;
				for (int32_t i = 0; i < t.num_c - 1; ++i) outs.push_back({st->scratch<int16_t>(n ? n : 1, ok), idx.w, idx.h});
				outs.push_back(idx);   // the index channel becomes the last colour channel, in place
				if (t.nb_deltas > 0) {
					std::vector<int16_t *> ptrs; for (const Ref &o : outs) ptrs.push_back(o.p);
					j40hip_device_state::ModOp op; memset(&op, 0, sizeof op);
					op.kind = 2; op.src = idx.p; op.aux = pal.p; op.p0 = pal.w; op.p1 = t.num_c; op.p2 = idx.w; op.p3 = idx.h; op.p4 = t.nb_colours; op.p5 = t.nb_deltas | (t.d_pred << 24);
					op.dst_list = st->upload(ptrs.data(), ptrs.size(), s, ok);
					op.wpp = st->upload(wpb, 12, s, ok);
					ops.push_back(op);
				} else {
					for (int32_t i = 0; i < t.num_c; ++i) {
						j40hip_device_state::ModOp op; memset(&op, 0, sizeof op);
						op.kind = 1; op.src = idx.p; op.aux = t.nb_colours > 0 ? pal.p + (size_t) i * (size_t) pal.w : nullptr; op.a = outs[(size_t) i].p; op.n = n; op.p0 = i; op.p1 = t.nb_colours;
						ops.push_back(op);
					}
				}
				std::vector<Ref> next(planes.begin() + 1, planes.begin() + first);
				next.insert(next.end(), outs.begin(), outs.end());
				next.insert(next.end(), planes.begin() + first + 1, planes.end());
				planes.swap(next);
			} else if (t.kind == Transform::SQUEEZE) {
				// one step: every squeezed channel and its residual channel are joined into a new plane (the recurrence runs along
				// the squeezed axis, so it is not done in place); the residual channels then leave the list
				const int32_t nc = (int32_t) planes.size(), end_c = t.begin_c + t.num_c, offset = t.in_place ? end_c : nc - t.num_c;
				if (t.begin_c < 0 || t.num_c < 1 || end_c > nc || offset + t.num_c > nc || offset < end_c) { ok = false; break; }
				for (int32_t c = t.begin_c; c < end_c; ++c) {
					const Ref avg = planes[(size_t) c], res = planes[(size_t) (offset + c - t.begin_c)];
					Ref out = {nullptr, t.horizontal ? avg.w + res.w : avg.w, t.horizontal ? avg.h : avg.h + res.h};
					if ((t.horizontal ? res.h != avg.h || (res.w != avg.w && res.w != avg.w - 1) : res.w != avg.w || (res.h != avg.h && res.h != avg.h - 1))) { ok = false; break; }
					const size_t n = (size_t) std::max(out.w, 0) * (size_t) std::max(out.h, 0);
					out.p = st->scratch<int16_t>(n ? n : 1, ok);
					j40hip_device_state::ModOp op; memset(&op, 0, sizeof op);
					op.kind = 4; op.src = avg.p; op.aux = res.p; op.a = out.p; op.p0 = avg.w; op.p1 = avg.h; op.p2 = res.w; op.p3 = res.h; op.p4 = t.horizontal ? 1 : 0;
					ops.push_back(op);
					planes[(size_t) c] = out;
				}
				if (ok) planes.erase(planes.begin() + offset, planes.begin() + offset + t.num_c);
			} else { ok = false; }
		}
		for (size_t k = ops_before; k < ops.size(); ++k) ops[k].group = group;
	};
	if (!hp.sub_images.empty()) {
		std::vector<DevSubPlane> subp(hp.sub_w.size());
		int32_t widest = hp.frame.width;
		for (size_t k = 0; k < subp.size(); ++k) {
			const size_t n = (size_t) hp.sub_w[k] * (size_t) hp.sub_h[k];
			subp[k] = DevSubPlane{st->scratch<int16_t>(n ? n : 1, ok), hp.sub_w[k], hp.sub_h[k], hp.sub_meta[k], 0};
			widest = std::max(widest, hp.sub_w[k]);
		}
		plan.sub_planes = st->upload(subp.data(), subp.size(), s, ok);
		for (const HostModPlan::SubImage &si : hp.sub_images) {
			if (!si.paste) continue;
			for (const Transform &t : si.transforms) palette_wp |= t.kind == Transform::PALETTE && t.nb_deltas > 0 && t.d_pred == 6;
			std::vector<Ref> sp;
			for (int32_t k = 0; k < si.num_planes; ++k) sp.push_back({subp[(size_t) (si.first_plane + k)].ptr, subp[(size_t) (si.first_plane + k)].w, subp[(size_t) (si.first_plane + k)].h});
			// (sections: LfGlobal's first, then passes x groups)
			const int32_t lead_sections = (int32_t) hp.sections.size() - hp.sections_per_pass * hp.num_passes;
			const int32_t sub_group = si.section >= lead_sections && hp.sections_per_pass > 0 ? (si.section - lead_sections) % hp.sections_per_pass : -1;
			schedule(sp, si.transforms, si.wp, st->mod_sub_ops, sub_group);
			const DevModSection &sec = hp.sections[(size_t) si.section];
			for (size_t c = 0; c < sp.size() && ok; ++c) {   // paste: rows of the sub-image over the section's rectangle
				const Ref &dst = planes[(size_t) sec.first_channel + c];
				if (sp[c].w != sec.gw || sp[c].h != sec.gh || (size_t) sec.first_channel + c >= planes.size()) { ok = false; break; }
				j40hip_device_state::ModOp op; memset(&op, 0, sizeof op);
				op.kind = 3; op.src = sp[c].p; op.a = dst.p + (size_t) sec.gy * (size_t) dst.w + (size_t) sec.gx; op.p0 = sp[c].w; op.p1 = sp[c].h; op.p2 = dst.w; op.group = sub_group;
				st->mod_sub_ops.push_back(op);
			}
		}
		if (palette_wp && !st->pal_wp_scratch) st->pal_wp_scratch = st->scratch<int32_t>((size_t) 2 * (size_t) widest * 5 + 16, ok);
	}
	{
		int8_t gwp[12]; const WPParams &wp = h->frame.gmodular.wp;
		gwp[0] = wp.p1; gwp[1] = wp.p2; for (int i = 0; i < 5; ++i) gwp[2 + i] = wp.p3[i]; for (int i = 0; i < 4; ++i) gwp[7 + i] = wp.w[i]; gwp[11] = 0;
		schedule(planes, hp.transforms, gwp, st->mod_ops, -1);
	}
	for (const Ref &p : planes) { st->final_planes.push_back(p.p); st->final_w.push_back(p.w); st->final_h.push_back(p.h); }
	st->alpha_channel = hp.alpha_channel;
	// the renderer needs three full-size colour planes (j40.h:7923)
	bool renderable = planes.size() >= 3;
	for (size_t c = 0; renderable && c < 3; ++c) renderable = planes[c].w == hp.frame.width && planes[c].h == hp.frame.height;
	if (st->alpha_channel >= 0) renderable = renderable && (size_t) st->alpha_channel < planes.size() && planes[(size_t) st->alpha_channel].w == hp.frame.width && planes[(size_t) st->alpha_channel].h == hp.frame.height;
	for (auto &e : st->ev) if (hipEventCreate(&e) != hipSuccess) ok = false;
	if (hipStreamSynchronize(s) != hipSuccess) ok = false;
	if (!ok) { j40hip_release_device(h); return ERR_GPU; }
	if (!renderable) { j40hip_release_device(h); return ERR_TODO; }
	return 0;
}

static uint32_t decode_modular(j40hip_frame *h, void *rgba_dev, size_t stride_bytes, hipStream_t s, float *ms3) {
	j40hip_device_state *st = h->dev;
	const DevModPlan &plan = st->mod;
	const Frame &fr = h->frame;
	if (ms3) (void) hipEventRecord(st->ev[0], s);
	if (hipMemsetAsync(plan.status, 0, sizeof(uint32_t) * ((size_t) st->total_sections + 1), s) != hipSuccess) return ERR_GPU;
	if (ms3) (void) hipEventRecord(st->ev[1], s);
	// LfGlobal's section and the first pass together, then every further pass on its own: a pass rewrites what the one before
	// it wrote (j40.h:7025-7033), so they must not overlap; the sections' own RCTs only matter for the last pass
	const int32_t per_pass = st->mod_sections_per_pass, lead = st->mod_sections - per_pass * st->mod_passes;
	// (sharded decodes, j40hip_frame_set_group_range: LfGlobal's section and this process' groups of every pass)
	const bool ranged = per_pass > 0 && !(st->first_group == 0 && st->num_groups == (int64_t) per_pass);
	const int32_t g0 = ranged ? (int32_t) st->first_group : 0, gn = ranged ? (int32_t) st->num_groups : per_pass;
	if (ranged) { launch_modular_sections(plan, 0, lead, st->mod_info, s); launch_modular_sections(plan, lead + g0, gn, st->mod_info, s); }
	else launch_modular_sections(plan, 0, lead + per_pass, st->mod_info, s);
	for (int32_t p = 1; p < st->mod_passes; ++p) launch_modular_sections(plan, lead + p * per_pass + g0, gn, st->mod_info, s);
	if (st->mod_local_rcts) launch_section_inverse_rcts(plan, lead + (st->mod_passes - 1) * per_pass + g0, gn, s);
	if (ms3) (void) hipEventRecord(st->ev[2], s);
	for (const std::vector<j40hip_device_state::ModOp> *ops : {&st->mod_sub_ops, &st->mod_ops}) for (const auto &op : *ops) {
		if (ranged && op.group >= 0 && (op.group < g0 || op.group >= g0 + gn)) continue;   // the sub-image of a group this process did not decode
		if (op.kind == 0) launch_inverse_rct(op.a, op.b, op.c, op.n, op.p0, s);
		else if (op.kind == 1) launch_inverse_palette_plain(op.src, op.aux, op.a, op.n, op.p0, op.p1, fr.im.bpp, s);
		else if (op.kind == 2) launch_inverse_palette_predicted(op.src, op.aux, op.p0, op.dst_list, op.p1, op.p2, op.p3, op.p4, op.p5 & 0xffffff, op.p5 >> 24, fr.im.bpp, op.wpp, st->pal_wp_scratch, st->mod_extra_status, s);
		else if (op.kind == 4) launch_inverse_squeeze(op.src, op.aux, op.a, op.p0, op.p1, op.p2, op.p3, op.p4 != 0, s);
		else launch_paste_plane(op.src, op.p0, op.p1, op.a, op.p2, s);
	}
	const int16_t *alpha = st->alpha_channel >= 0 ? st->final_planes[(size_t) st->alpha_channel] : nullptr;
	if (ranged) {   // only this process' pixels are written
		int32_t rects[3][4];
		const int nr = group_range_rects(g0, gn, fr.fh.width, fr.fh.height, fr.fh.group_size_shift, rects);
		for (int k = 0; k < nr; ++k) launch_pack_planes_rect(st->final_planes[0], st->final_planes[1], st->final_planes[2], alpha, fr.fh.width, rects[k][0], rects[k][1], rects[k][2] - rects[k][0], rects[k][3] - rects[k][1], fr.im.bpp, (uint8_t *) rgba_dev, stride_bytes, s);
	} else launch_pack_planes(st->final_planes[0], st->final_planes[1], st->final_planes[2], alpha, fr.fh.width, fr.fh.height, fr.im.bpp, (uint8_t *) rgba_dev, stride_bytes, s);
	if (ms3) {
		(void) hipEventRecord(st->ev[3], s);
		if (hipEventSynchronize(st->ev[3]) != hipSuccess) return ERR_GPU;
		float a = 0, b = 0, c = 0;
		(void) hipEventElapsedTime(&a, st->ev[0], st->ev[1]); (void) hipEventElapsedTime(&b, st->ev[1], st->ev[2]); (void) hipEventElapsedTime(&c, st->ev[2], st->ev[3]);
		ms3[0] = b; ms3[1] = c; ms3[2] = a;
	}
	return hipGetLastError() == hipSuccess ? 0 : ERR_GPU;
}

// ---- LfGroup streams on the device (lf_decode.hip): Frame::lf_decoder for frames parsed with j40hip_frame_parse_on ----
// A parsing thread stages its frame's inputs (codestream, tree, alias tables) into a device block of its own, hands the frame's
// tasks to the device's LF service and sleeps; the service thread gathers the tasks of every waiting frame into ONE launch
// (a frame alone is 12 wavefronts for ~0.15 s; and launches from a hundred streams would queue up behind one another on the
// few hardware queues a process gets), at most two launches in flight; the parsing thread then copies its planes back.
struct LfDecodeContext { int device; hipStream_t stream; };
thread_local PinnedStage t_lf_out;        // results land here; LfDeviceTask's pointers point into it until the thread's next call
thread_local hipEvent_t t_lf_done = nullptr;

struct LfRequest { std::vector<DevLfTask> tasks; bool done = false, ok = false; };
struct LfService {
	std::mutex m;
	std::condition_variable cv_work, cv_done;
	std::deque<LfRequest *> pending;
	bool started = false, stop = false;
	int device = 0;
	std::thread thread;
};
// The services live on the heap and are taken down by j40hip_shutdown only (which stops and JOINS their threads): a process that
// never calls it leaves them asleep on their condition variables until it ends, and no destructor of a static object runs with a
// thread still waiting on it (that was undefined behaviour, and hung interpreters at exit).
std::mutex g_lf_service_mutex;
LfService *g_lf_services[16] = {nullptr};
LfService *lf_service(int device) {
	std::lock_guard<std::mutex> lock(g_lf_service_mutex);
	if (!g_lf_services[device]) g_lf_services[device] = new LfService();
	return g_lf_services[device];
}

void lf_service_main(LfService *sv) {
	struct Flight { std::vector<LfRequest *> reqs; int slot; };
	hipStream_t stream[2] = {nullptr, nullptr}; hipEvent_t done[2] = {nullptr, nullptr};
	PinnedStage host_tasks[2]; void *dev_tasks[2] = {nullptr, nullptr}; size_t dev_cap[2] = {0, 0};
	auto dev_reserve = [&](int slot, size_t bytes) {   // grow-only
		if (bytes <= dev_cap[slot]) return true;
		if (dev_tasks[slot]) (void) hipFree(dev_tasks[slot]);
		dev_tasks[slot] = nullptr; dev_cap[slot] = 0;
		if (hipMalloc(&dev_tasks[slot], bytes * 2) != hipSuccess) { (void) hipGetLastError(); return false; }
		dev_cap[slot] = bytes * 2;
		return true;
	};
	bool usable = hipSetDevice(sv->device) == hipSuccess;
	for (int i = 0; i < 2 && usable; ++i) usable = hipStreamCreateWithFlags(&stream[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&done[i], hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
	std::deque<Flight> inflight;
	int next_slot = 0;
	for (;;) {
		std::vector<LfRequest *> take;
		{
			std::unique_lock<std::mutex> lock(sv->m);
			sv->cv_work.wait(lock, [&] { return sv->stop || !sv->pending.empty() || !inflight.empty(); });
			if (sv->stop && sv->pending.empty() && inflight.empty()) break;
			if (!sv->pending.empty() && inflight.size() < 2) {
				if (sv->pending.size() < 24) sv->cv_work.wait_for(lock, std::chrono::milliseconds(3));   // let the other parsing threads catch up: one launch for all
				while (!sv->pending.empty() && take.size() < 160) { take.push_back(sv->pending.front()); sv->pending.pop_front(); }
			}
		}
		if (!take.empty()) {
			const int slot = next_slot; next_slot ^= 1;
			size_t n = 0;
			for (const LfRequest *r : take) n += r->tasks.size();
			bool ok = usable && host_tasks[slot].reserve(sizeof(DevLfTask) * n + 64, 0) && dev_reserve(slot, sizeof(DevLfTask) * n + 64);
			if (ok) {
				DevLfTask *h = (DevLfTask *) host_tasks[slot].ptr; size_t k = 0;
				for (const LfRequest *r : take) for (const DevLfTask &t : r->tasks) h[k++] = t;
				ok = hipMemcpyAsync(dev_tasks[slot], h, sizeof(DevLfTask) * n, hipMemcpyHostToDevice, stream[slot]) == hipSuccess;
				if (ok) { launch_lf_groups((const DevLfTask *) dev_tasks[slot], (int32_t) n, stream[slot]); ok = hipGetLastError() == hipSuccess; }
				ok = ok && hipEventRecord(done[slot], stream[slot]) == hipSuccess;
			}
			if (ok) inflight.push_back(Flight{std::move(take), slot});
			else {
				(void) hipGetLastError();
				std::unique_lock<std::mutex> lock(sv->m);
				for (LfRequest *r : take) { r->done = true; r->ok = false; }
				sv->cv_done.notify_all();
			}
			if (inflight.size() < 2) continue;   // room for another launch: look for more work first
		}
		if (!inflight.empty()) {
			Flight fl = std::move(inflight.front()); inflight.pop_front();
			const bool ok = hipEventSynchronize(done[fl.slot]) == hipSuccess;
			std::unique_lock<std::mutex> lock(sv->m);
			for (LfRequest *r : fl.reqs) { r->done = true; r->ok = ok; }
			sv->cv_done.notify_all();
		}
	}
	for (int i = 0; i < 2; ++i) {
		if (stream[i]) { (void) hipStreamSynchronize(stream[i]); (void) hipStreamDestroy(stream[i]); }
		if (done[i]) (void) hipEventDestroy(done[i]);
		if (dev_tasks[i]) (void) hipFree(dev_tasks[i]);
		host_tasks[i].release();
	}
}

static bool lf_device_decode(void *ctx_, const Frame &f, const uint8_t *cs, size_t cs_size, std::vector<LfDeviceTask> &tasks) {
	const LfDecodeContext &ctx = *(const LfDecodeContext *) ctx_;
	if (tasks.empty() || cs_size + 16 >= ((size_t) 1 << 29) || ctx.device < 0 || ctx.device >= 16) return false;
	if (hipSetDevice(ctx.device) != hipSuccess) return false;
	DevCoopTree tree; std::vector<uint64_t> alias; int32_t log_alpha = 0;
	if (!build_lf_coop(f, &tree, &alias, &log_alpha)) return false;
	// this frame's inputs go up in one staged copy; one block of device memory holds them, the output planes and the results
	std::vector<size_t> out_off(tasks.size());
	size_t out_elems = 0;
	for (size_t i = 0; i < tasks.size(); ++i) {
		const size_t cells = (size_t) tasks[i].w8 * (size_t) tasks[i].h8, c64 = (size_t) tasks[i].w64 * (size_t) tasks[i].h64;
		out_off[i] = out_elems; out_elems += (6 * cells + 2 * c64 + 63) & ~(size_t) 63;   // lf[3], xfromy, bfromy, info (2 * cells), sharpness
	}
	Stager sg;
	const size_t o_cs = sg.put(cs, cs_size); (void) sg.reserve(32);   // (the decoder's word window reads a little past the last section)
	const size_t o_tree = sg.put(&tree, 1), o_alias = sg.put(alias.data(), alias.size());
	const size_t copy_bytes = sg.size;
	const size_t o_res = sg.reserve(sizeof(DevLfResult) * tasks.size()), o_out = sg.reserve(sizeof(int16_t) * out_elems);
	if (!sg.ok) return false;
	memset(t_stage.ptr + o_cs + cs_size, 0, 32);
	size_t block_bytes = 0; bool clean = false;
	uint8_t *block = (uint8_t *) cache_acquire(ctx.device, sg.size, &block_bytes, &clean);
	if (!block) return false;
	const size_t res_bytes = sizeof(DevLfResult) * tasks.size(), out_bytes = sizeof(int16_t) * out_elems, res_off = (out_bytes + 255) & ~(size_t) 255;
	bool ok = t_lf_out.reserve(res_off + res_bytes + 64, 0);
	if (ok && !t_lf_done) ok = hipEventCreateWithFlags(&t_lf_done, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;   // (waits here are sleeps, not spins)
	auto wait = [&]() { return hipEventRecord(t_lf_done, ctx.stream) == hipSuccess && hipEventSynchronize(t_lf_done) == hipSuccess; };
	ok = ok && hipMemcpyAsync(block, sg.data(), copy_bytes, hipMemcpyHostToDevice, ctx.stream) == hipSuccess && wait();
	LfRequest req;
	if (ok) {
		req.tasks.resize(tasks.size());
		for (size_t i = 0; i < tasks.size(); ++i) {
			const LfDeviceTask &t = tasks[i];
			DevLfTask &d = req.tasks[i];
			d.codestream = block + o_cs; d.tree = (const DevCoopTree *) (block + o_tree); d.alias = (const uint64_t *) (block + o_alias); d.log_alpha_size = log_alpha;
			d.byte_off = (uint32_t) t.byte_off; d.size = (uint32_t) t.size; d.bit_off = t.bit_off;
			d.w8 = t.w8; d.h8 = t.h8; d.w64 = t.w64; d.h64 = t.h64; d.sidx0 = t.sidx0; d.sidx2 = t.sidx2; d.nbvb_bits = t.nbvb_bits;
			{   // this task's planes, one after the other: lf[3], xfromy, bfromy, varblock info (room for one varblock per cell), sharpness
				const size_t cells = (size_t) t.w8 * (size_t) t.h8, c64 = (size_t) t.w64 * (size_t) t.h64;
				int16_t *p = (int16_t *) (block + o_out) + out_off[i];
				for (int c = 0; c < 3; ++c) d.lf[c] = p + (size_t) c * cells;
				d.xfromy = p + 3 * cells; d.bfromy = d.xfromy + c64; d.info = d.bfromy + c64; d.sharp = d.info + 2 * cells; d.info_capacity = (uint32_t) (2 * cells);
			}
			d.result = (DevLfResult *) (block + o_res) + i;
		}
		LfService &sv = *lf_service(ctx.device);
		std::unique_lock<std::mutex> lock(sv.m);
		if (!sv.started) { sv.started = true; sv.device = ctx.device; sv.thread = std::thread(lf_service_main, &sv); }
		sv.pending.push_back(&req);
		sv.cv_work.notify_all();
		sv.cv_done.wait(lock, [&] { return req.done; });
		ok = req.ok;
	}
	ok = ok && hipMemcpyAsync(t_lf_out.ptr, block + o_out, out_bytes, hipMemcpyDeviceToHost, ctx.stream) == hipSuccess;
	ok = ok && hipMemcpyAsync(t_lf_out.ptr + res_off, block + o_res, res_bytes, hipMemcpyDeviceToHost, ctx.stream) == hipSuccess;
	ok = ok && wait();
	if (!ok) (void) hipStreamSynchronize(ctx.stream);   // nothing of this call may still be in flight when the block goes back
	cache_release(ctx.device, block, block_bytes, false);
	if (!ok) { (void) hipGetLastError(); return false; }
	const DevLfResult *res = (const DevLfResult *) (t_lf_out.ptr + res_off);
	const int16_t *out = (const int16_t *) t_lf_out.ptr;
	for (size_t i = 0; i < tasks.size(); ++i) {
		LfDeviceTask &t = tasks[i];
		const size_t cells = (size_t) t.w8 * (size_t) t.h8, c64 = (size_t) t.w64 * (size_t) t.h64;
		const int16_t *p = out + out_off[i];
		t.status = res[i].status; t.nb_varblocks = res[i].nb_varblocks;
		for (int c = 0; c < 3; ++c) t.lf[c] = p + (size_t) c * cells;
		t.xfromy = p + 3 * cells; t.bfromy = t.xfromy + c64; t.info0 = t.bfromy + c64; t.info1 = t.info0 + (t.nb_varblocks > 0 ? t.nb_varblocks : 0);
		t.sharp = t.info0 + 2 * cells;
	}
	return true;
}

// j40hip_frame_parse_ex with the LfGroup streams decoded on `device` (flags bit 1 must be set: the LF tail runs there too); the call
// blocks (asleep) while the device works. Frames the device decoder cannot take are parsed on the host as usual.
extern "C" j40hip_frame *j40hip_frame_parse_on(const void *buf, size_t size, int threads, uint32_t flags, int device, void *stream, uint32_t *err) {
	LfDecodeContext ctx = {device, (hipStream_t) stream};
	const bool usable = (flags & 1u) && device >= 0 && device < j40hip_device_count();
	return j40hip_frame_parse_with(buf, size, threads, flags, usable ? lf_device_decode : nullptr, usable ? &ctx : nullptr, err);
}
extern "C" int j40hip_frame_lf_on_device(const j40hip_frame *f) { return f && f->frame.lf_decoded_on_device ? 1 : 0; }

static thread_local HostPlan t_host_plan;

// the frame's varblock list on the host (sharded decodes, stage dumps): copied back from the device when first asked for
static bool host_vb_sorted(j40hip_device_state *st) {
	if (st->vb_sorted.size() == st->vb_count) return true;
	st->vb_sorted.resize(st->vb_count);
	if (hipSetDevice(st->device) != hipSuccess || hipMemcpy(st->vb_sorted.data(), st->d_vb_sorted, sizeof(DevVarblock) * st->vb_count, hipMemcpyDeviceToHost) != hipSuccess) { st->vb_sorted.clear(); return false; }
	return true;
}

// `s`: the stream the copies and fills are enqueued on; the call returns once they have completed (the plan is staged in the
// calling thread's pinned buffer, which the next upload of this thread reuses)
static uint32_t upload_impl(j40hip_frame *h, int device, hipStream_t s) {
	if (!h) return ERR_GPU;
	if (h->dev) j40hip_release_device(h);
	if (j40hip_device_count() <= device || hipSetDevice(device) != hipSuccess) return ERR_GPU;
	if (h->frame.fh.is_modular) return upload_modular(h, device);
	HostPlan &hp = t_host_plan;   // (this thread's, storage kept from frame to frame)
	hp.reset();
	hp.force_dense = h->force_dense;
	static const bool timing = getenv("J40HIP_API_TIMING") != nullptr;   // (where an upload's time goes: plan build, staging, copy + LfGroup tail)
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tu0 = timing ? now() : 0;
	if (uint32_t e = build_vardct_plan(h->frame, h->cs, h->cs_size, &hp, h->threads)) return e;
	const double tu1 = timing ? now() : 0;

	j40hip_device_state *st = new j40hip_device_state();
	h->dev = st; st->device = device; st->force_dense = h->force_dense;
	bool ok = ensure_constant_tables(device);
	DevPlan &plan = st->plan;
	memset(&plan, 0, sizeof plan);
	st->hf = hp.hf;
	st->vb_count = hp.vb_sorted.size();   // (the list itself stays on the device: host_vb_sorted fetches it for the rare callers)
	memcpy(st->class_start, hp.class_start, sizeof st->class_start);
	Stager sg;
	sg.deferred = h->threads > 1;
	const size_t o_cs = sg.put(hp.codestream.data(), hp.codestream.size()), o_u8 = sg.put(hp.pool_u8.data(), hp.pool_u8.size());
	const size_t o_u16 = sg.put(hp.pool_u16.data(), hp.pool_u16.size()), o_i32 = sg.put(hp.pool_i32.data(), hp.pool_i32.size());
	const size_t o_u64 = sg.put(hp.pool_u64.data(), hp.pool_u64.size()), o_f32 = sg.put(hp.pool_f32.data(), hp.pool_f32.size());
	const size_t o_cl = sg.put(hp.clusters.data(), hp.clusters.size()), o_spec = sg.put(hp.coeff_specs.data(), hp.coeff_specs.size());
	const size_t o_lfg = sg.put(hp.lf_groups.data(), hp.lf_groups.size()), o_sec = sg.put(hp.sections.data(), hp.sections.size());
	const size_t o_gb = sg.put(hp.group_blocks.data(), hp.group_blocks.size()), o_gbs = sg.put(hp.group_block_start.data(), hp.group_block_start.size());
	const size_t o_frame = sg.put(&hp.frame, 1), o_blocks = sg.put(hp.blocks.data(), hp.blocks.size()), o_lfi = sg.put(hp.lfindices.data(), hp.lfindices.size());
	const size_t cells = hp.blocks.size();
	size_t o_llf[3], o_raw[3] = {0, 0, 0};
	if (hp.lf_tail_pending) {   // the LLF arrays are an output of the device's LfGroup tail: only their place is reserved
		for (int c = 0; c < 3; ++c) { o_raw[c] = sg.put(hp.lfraw[c].data(), hp.lfraw[c].size()); o_llf[c] = 0; }   // (reserved behind everything that is copied, below)
	} else for (int c = 0; c < 3; ++c) o_llf[c] = sg.put(hp.llf[c].data(), hp.llf[c].size());
	const size_t o_vbc = sg.put(hp.vb_coeffoff_qfidx.data(), hp.vb_coeffoff_qfidx.size()), o_vbh = sg.put(hp.vb_hfmul_inv.data(), hp.vb_hfmul_inv.size());
	const size_t o_xfy = sg.put(hp.xfromy.data(), hp.xfromy.size()), o_bfy = sg.put(hp.bfromy.data(), hp.bfromy.size());
	const size_t o_vbs = sg.put(hp.vb_sorted.data(), hp.vb_sorted.size());
	const size_t o_evr = sg.put(hp.ev_range.data(), hp.ev_range.size());
	const size_t copy_bytes = sg.size;
	if (hp.lf_tail_pending) for (int c = 0; c < 3; ++c) o_llf[c] = sg.reserve(sizeof(float) * cells);
	bool dummy_clean = false;
	if (!sg.ok) ok = false;
	sg.flush(h->threads);
	const double tu2 = timing ? now() : 0;
	st->plan_block = ok ? cache_acquire(device, sg.size, &st->plan_block_bytes, &dummy_clean) : nullptr;
	if (!st->plan_block || hipMemcpyAsync(st->plan_block, sg.data(), copy_bytes, hipMemcpyHostToDevice, s) != hipSuccess) ok = false;
	uint8_t *pb = (uint8_t *) st->plan_block;
	plan.codestream = pb + o_cs; plan.pool_u8 = pb + o_u8; plan.pool_u16 = (const uint16_t *) (pb + o_u16); plan.pool_i32 = (const int32_t *) (pb + o_i32);
	plan.pool_u64 = (const uint64_t *) (pb + o_u64); plan.pool_f32 = (const float *) (pb + o_f32); plan.clusters = (const DevCluster *) (pb + o_cl);
	plan.coeff_specs = (const DevCodeSpec *) (pb + o_spec); plan.lf_groups = (const DevLfGroup *) (pb + o_lfg); plan.sections = (const DevSection *) (pb + o_sec);
	plan.group_blocks = (const DevGroupBlock *) (pb + o_gb); plan.group_block_start = (const uint32_t *) (pb + o_gbs); plan.frame = (const DevFrame *) (pb + o_frame);
	plan.block_ctx_map_off = hp.block_ctx_map_off;
	plan.blocks = (const int32_t *) (pb + o_blocks); plan.lfindices = pb + o_lfi;
	for (int c = 0; c < 3; ++c) { plan.llf[c] = (const float *) (pb + o_llf[c]); plan.lfraw[c] = hp.lf_tail_pending ? (const int16_t *) (pb + o_raw[c]) : nullptr; }
	plan.vb_coeffoff_qfidx = (const int32_t *) (pb + o_vbc); plan.vb_hfmul_inv = (const float *) (pb + o_vbh);
	plan.xfromy = (const int16_t *) (pb + o_xfy); plan.bfromy = (const int16_t *) (pb + o_bfy);
	st->d_vb_sorted = (DevVarblock *) (pb + o_vbs);
	plan.ev_range = (const uint32_t *) (pb + o_evr);
	// working set: the coefficients -- event lists plus the per-block table (single-pass frames) or three dense planes in one
	// allocation (multi-pass frames; hf_lanes_dev.h addresses a lane's channel by offset) --, the non-zero scratch, status
	// words, LZ77 windows, the scratch of the 128/256-sized transforms
	st->coeff_floats = hp.coeff_floats;
	st->num_blocks = hp.group_blocks.size();
	const int32_t num_groups = hp.frame.num_groups;
	{
		auto align = [](size_t v) { return (v + 255) & ~(size_t) 255; };
		const bool sparse = hp.frame.sparse_coeffs != 0;
		const size_t stride = (st->coeff_floats + 63) & ~(size_t) 63;
		const size_t coeff_bytes = sparse ? sizeof(CoeffEvent) * hp.ev_capacity : sizeof(float) * 3 * stride;
		const size_t w_coeffs = 0, w_blk = align(w_coeffs + coeff_bytes), blk_bytes = sparse ? sizeof(uint32_t) * 4 * st->num_blocks : 0;
		const size_t w_nz = align(w_blk + blk_bytes), w_status = align(w_nz + (size_t) num_groups * 32 * 32 * 3);
		const size_t w_endbit = align(w_status + sizeof(uint32_t) * hp.sections.size());
		const size_t w_lz = align(w_endbit + (hp.frame.sections_have_trailer ? sizeof(uint32_t) * hp.sections.size() : 0)), lz_bytes = sizeof(int32_t) * (size_t) num_groups * hp.lz_window_size;
		// (the 128/256-sized transforms' scratch doubles as the LfGroup tail's: three planes of dequantised, smoothed LF samples, used
		// once at upload, long before any decode)
		const size_t w_large = align(w_lz + lz_bytes), large_bytes = std::max(sizeof(float) * (size_t) hp.max_large * 6 * 65536, hp.lf_tail_pending ? sizeof(float) * 3 * cells : (size_t) 0);
		bool unused_clean = false;
		st->work_block = cache_acquire(device, w_large + large_bytes + 256, &st->work_block_bytes, &unused_clean);
		uint8_t *wb = (uint8_t *) st->work_block;
		if (!wb) ok = false;
		else {
			if (sparse) {
				plan.events = (CoeffEvent *) (wb + w_coeffs); plan.block_events = (uint32_t *) (wb + w_blk);
				if (hipMemsetAsync(plan.block_events, 0, blk_bytes, s) != hipSuccess) ok = false;   // recycled memory: no entry may point outside the event list
			}
			else for (int c = 0; c < 3; ++c) plan.coeffs[c] = (float *) (wb + w_coeffs) + (size_t) c * stride;
			plan.coeff_stride = (uint32_t) stride;
			plan.nonzeros = (int8_t *) (wb + w_nz); plan.status = (uint32_t *) (wb + w_status);
			plan.section_end_bit = hp.frame.sections_have_trailer ? (uint32_t *) (wb + w_endbit) : nullptr;
			plan.lz_window_size = hp.lz_window_size;
			plan.lz_window = hp.lz_window_size ? (int32_t *) (wb + w_lz) : nullptr;
			st->d_large_scratch = hp.max_large ? (float *) (wb + w_large) : nullptr;
			if (hp.lf_tail_pending && ok) {   // the LfGroup tail: LF integers -> LLF coefficients, on the upload stream behind the copy
				int32_t max_cells = 0;
				for (const DevLfGroup &g : hp.lf_groups) max_cells = std::max(max_cells, g.width8 * g.height8);
				launch_lf_tail(plan, (int32_t) hp.lf_groups.size(), max_cells, cells, (float *) (wb + w_large), st->d_vb_sorted, (int32_t) st->vb_count, st->class_start[18], hp.lf_smooth ? 1 : 0, hp.inv_m_lf, s);
			}
		}
	}
	st->total_sections = (int32_t) hp.sections.size();
	st->has_trailers = hp.frame.sections_have_trailer != 0 && !h->from_view;
	st->first_group = 0; st->num_groups = num_groups;
	for (auto &e : st->ev) if (hipEventCreate(&e) != hipSuccess) ok = false;
	// (asleep while the copy runs, like lf_device_decode: a pipeline may have many more uploading threads than CPUs)
	if (!t_lf_done && hipEventCreateWithFlags(&t_lf_done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { t_lf_done = nullptr; (void) hipGetLastError(); }
	if (t_lf_done ? (hipEventRecord(t_lf_done, s) != hipSuccess || hipEventSynchronize(t_lf_done) != hipSuccess) : hipStreamSynchronize(s) != hipSuccess) ok = false;
	if (timing) fprintf(stderr, "[j40hip upload] plan build %.2f ms (%d threads), staging %.2f ms (%.1f MB), copy + LfGroup tail + wait %.2f ms\n", tu1 - tu0, h->threads, tu2 - tu1, (double) copy_bytes / 1e6, now() - tu2);
	if (!ok) { j40hip_release_device(h); return ERR_GPU; }
	return 0;
}
static uint32_t j40hip_frame_upload_body(j40hip_frame *h, int device) { return upload_impl(h, device, nullptr); }

extern "C" void j40hip_frame_force_dense(j40hip_frame *h, int dense) { if (h) h->force_dense = dense != 0; }

static uint32_t j40hip_frame_set_group_range_body(j40hip_frame *h, int64_t first_group, int64_t num_groups) {
	if (!h || !h->dev) return ERR_GPU;
	if (first_group < 0 || num_groups < 0 || first_group + num_groups > h->frame.fh.num_groups) return ERR_RNGE;
	j40hip_device_state *st = h->dev;
	if (st->is_modular) {
		// Modular frames: the groups' sections are independent of each other (no predictor looks across a group's edge), and so are
		// the per-pixel inverse transforms (RCT, plain palette); a palette with predicted deltas or a Squeeze step reads across
		// groups, and frames coded with Squeeze have no one-section-per-group layout at all: those are decoded whole
		const bool whole = first_group == 0 && num_groups == h->frame.fh.num_groups;
		if (!whole) {
			if (st->mod_sections_per_pass != (int32_t) h->frame.fh.num_groups) return ERR_TODO;
			for (const auto &op : st->mod_ops) if (op.kind == 2 || op.kind == 4) return ERR_TODO;
		}
		st->first_group = first_group; st->num_groups = num_groups;
		return 0;
	}
	st->first_group = first_group; st->num_groups = num_groups;
	if (first_group == 0 && num_groups == h->frame.fh.num_groups) return 0;
	// varblocks never straddle a group (the largest transform is one group wide), so the pixel kernels' work lists are
	// the full lists filtered by the group of each block's top-left pixel
	const FrameHeader &fh = h->frame.fh;
	const int32_t shift = fh.group_size_shift;
	std::vector<DevVarblock> sel;
	if (!host_vb_sorted(st)) return ERR_GPU;
	for (const DevVarblock &vb : st->vb_sorted) {
		const int64_t gid = ((int64_t) vb.py >> shift) * fh.gcolumns + ((int64_t) vb.px >> shift);
		if (gid >= first_group && gid < first_group + num_groups) sel.push_back(vb);
	}
	size_t k = 0;   // sel keeps the DctSelect order of vb_sorted
	for (int d = 0; d <= 27; ++d) { while (k < sel.size() && sel[k].dctsel < d) ++k; st->range_class_start[d] = (int32_t) k; }
	if (hipSetDevice(st->device) != hipSuccess) return ERR_GPU;
	if (sel.size() > st->vb_range_capacity) {
		bool ok = true;
		st->d_vb_range = st->scratch<DevVarblock>(sel.size(), ok);
		if (!ok) return ERR_GPU;
		st->vb_range_capacity = sel.size();
	}
	if (!sel.empty() && hipMemcpy(st->d_vb_range, sel.data(), sizeof(DevVarblock) * sel.size(), hipMemcpyHostToDevice) != hipSuccess) return ERR_GPU;
	return 0;
}

// dense planes are cleared before every decode (the passes accumulate into them). Sparse coefficients need nothing: the per-block
// table is cleared once per upload, and an entry the entropy kernel does not rewrite (a section failed before reaching the
// block) still describes events of this frame's previous decode -- in range, and the frame is reported as failed anyway.
static uint32_t clear_before_decode(j40hip_device_state *st, hipStream_t s) {
	const DevPlan &plan = st->plan;
	if (plan.events) return 0;
	return hipMemsetAsync(plan.coeffs[0], 0, sizeof(float) * 3 * (size_t) plan.coeff_stride, s) == hipSuccess ? 0 : ERR_GPU;
}

// VarDCT frames with extra channels (latency path): decodes the Modular sub-image that follows the HF coefficients in every
// section -- into planes nobody reads, the reference drops them too (j40.h:7868) -- so that damage there is reported like the
// reference reports it. Needs the entropy kernel's results on the host (where each section's coefficients ended), i.e. it
// synchronises the stream; the statuses it finds are written back into the frame's status array.
static uint32_t validate_trailers(j40hip_frame *h, hipStream_t s) {
	j40hip_device_state *st = h->dev;
	const Frame &fr = h->frame;
	const size_t n = (size_t) st->total_sections;
	std::vector<uint32_t> end_bits(n), status(n);
	if (hipMemcpyAsync(end_bits.data(), st->plan.section_end_bit, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s) != hipSuccess) return ERR_GPU;
	if (hipMemcpyAsync(status.data(), st->plan.status, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s) != hipSuccess) return ERR_GPU;
	if (hipStreamSynchronize(s) != hipSuccess) return ERR_GPU;
	HostModPlan hp;
	std::vector<std::pair<int32_t, uint32_t>> header_errors;
	std::vector<int32_t> section_of;
	if (uint32_t e = build_trailer_plan(fr, h->cs, h->cs_size, end_bits.data(), status.data(), &hp, &header_errors, &section_of)) return e;
	j40hip_device_state tmp;   // owns the buffers of this validation only
	tmp.device = st->device;
	bool ok = true;
	std::vector<uint32_t> found(hp.sections.size(), 0);
	if (!hp.sections.empty()) {
		DevModPlan plan;
		memset(&plan, 0, sizeof plan);
		plan.frame = tmp.upload(&hp.frame, 1, s, ok);
		plan.codestream = st->plan.codestream;
		plan.pool_u8 = tmp.upload(hp.pool_u8.data(), hp.pool_u8.size(), s, ok);
		plan.pool_i32 = tmp.upload(hp.pool_i32.data(), hp.pool_i32.size(), s, ok);
		plan.pool_u64 = tmp.upload(hp.pool_u64.data(), hp.pool_u64.size(), s, ok);
		plan.clusters = tmp.upload(hp.clusters.data(), hp.clusters.size(), s, ok);
		plan.spec = tmp.upload(hp.specs.data(), hp.specs.size(), s, ok);
		plan.tree = tmp.upload(hp.tree.data(), hp.tree.size(), s, ok);
		plan.sections = tmp.upload(hp.sections.data(), hp.sections.size(), s, ok);
		if (!hp.coop_trees.empty()) plan.coop_trees = tmp.upload(hp.coop_trees.data(), hp.coop_trees.size(), s, ok);
		std::vector<DevSubPlane> subp(hp.sub_w.size());
		size_t total = 0;
		for (size_t k = 0; k < subp.size(); ++k) total += (size_t) hp.sub_w[k] * (size_t) hp.sub_h[k] + 1;
		int16_t *pool = tmp.scratch<int16_t>(total + 1, ok);
		for (size_t k = 0, at = 0; k < subp.size() && pool; ++k) { subp[k] = DevSubPlane{pool + at, hp.sub_w[k], hp.sub_h[k], hp.sub_meta[k], 0}; at += (size_t) hp.sub_w[k] * (size_t) hp.sub_h[k] + 1; }
		plan.sub_planes = tmp.upload(subp.data(), subp.size(), s, ok);
		if (hp.frame.tree_uses_wp) plan.wp_scratch = tmp.scratch<int32_t>(hp.sections.size() * (size_t) (2 * hp.frame.max_width * 5) + 16, ok);
		plan.lz_window_size = hp.lz_window_size;
		if (hp.lz_window_size) plan.lz_window = tmp.scratch<int32_t>(hp.sections.size() * hp.lz_window_size, ok);
		plan.status = tmp.scratch<uint32_t>(hp.sections.size() + 1, ok);
		if (hp.split_sections) { plan.residuals = tmp.scratch<int32_t>(hp.split_samples + 64, ok); plan.split_state = tmp.scratch<uint32_t>(3 * hp.sections.size() + 4, ok); }
		if (ok) {
			const ModLaunchInfo info = {hp.max_tree_nodes, hp.max_num_dist, hp.max_clusters, hp.max_table_bytes, hp.frame.max_width, hp.any_wp ? 1 : 0, hp.coop_width, hp.coop_sections + hp.split_sections == (int32_t) hp.sections.size(), hp.quad_sections, hp.quad_spec, hp.quad_width, hp.coop_sections, hp.quad_sections ? hp.specs[(size_t) hp.quad_spec].table_span : 0u, hp.split_sections, hp.split_width, hp.split_channels};
			launch_modular_sections(plan, 0, (int32_t) hp.sections.size(), info, s);
			ok = hipMemcpyAsync(found.data(), plan.status, sizeof(uint32_t) * found.size(), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
		}
	}
	bool any = false;
	for (size_t i = 0; i < found.size(); ++i) if (found[i]) { status[(size_t) section_of[i]] = found[i]; any = true; }
	for (const auto &e : header_errors) { status[(size_t) e.first] = e.second; any = true; }
	if (ok && any) ok = hipMemcpyAsync(st->plan.status, status.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
	else if (ok) ok = hipStreamSynchronize(s) == hipSuccess;
	for (auto &b : tmp.buffers) b.release();
	tmp.buffers.clear();
	return ok ? 0 : ERR_GPU;
}

// ---- restoration filters (device/restore_dev.h, restore_kernels.h) ----
// Off unless asked for: j40 parses the frame header's RestorationFilter bundle and ignores it (j40.h:5339-5366; its j40__gaborish /
// j40__epf are never called), and the default decode matches j40. J40HIP_RESTORATION=1 (or j40hip_frame_set_restoration(f, 1)) runs the
// filters a VarDCT frame signals; =j40 (2) runs them exactly as j40's routines stand, aliased line buffers and all (restore_dev.h).
static int restoration_mode(const j40hip_frame *h) {
	if (h->restoration >= 0) return h->restoration;
	static const int env = [] { const char *e = getenv("J40HIP_RESTORATION"); return !e ? 0 : !strcmp(e, "j40") ? 2 : atoi(e) > 0 ? 1 : 0; }();
	return env;
}
static bool surely_nonzero(float x) { return std::isfinite(x) && std::fabs(x) >= 1e-8f; }   // j40.h:625
// the kernels' parameters from the frame header's; 0 or the reference routines' own complaints: "gab0" (j40.h:7289), "epf0" (j40.h:7384)
static uint32_t restore_params(const FrameHeader &fh, int mode, RestoreParams *p) {
	const FrameHeader::Restoration &r = fh.restoration;
	memset(p, 0, sizeof *p);
	p->width = fh.width; p->height = fh.height; p->w8 = (fh.width + 7) / 8; p->h8 = (fh.height + 7) / 8;
	p->quirk = mode == 2 ? 1 : 0;
	if (r.gab) for (int c = 0; c < 3; ++c) {
		float w0 = 1.0f, w1 = r.gab_weights[c][0], w2 = r.gab_weights[c][1];
		const float wsum = w0 + w1 * 4 + w2 * 4;
		if (!surely_nonzero(wsum)) return ERR4('g', 'a', 'b', '0');
		p->gab_w[c][0] = w0 / wsum; p->gab_w[c][1] = w1 / wsum; p->gab_w[c][2] = w2 / wsum;
	}
	if (r.epf_iters > 0) {
		for (int i = 0; i < 8; ++i) {
			const float q = r.quant_mul * r.sharp_lut[i];
			if (!surely_nonzero(q)) return ERR4('e', 'p', 'f', '0');
			p->inv_quant_sharp_lut[i] = 1.0f / q;
		}
		const float scale[3] = {r.pass0_sigma_scale, 1.0f, r.pass2_sigma_scale};
		for (int k = 0; k < 3; ++k) { p->sigma_scale[k] = scale[k] * 1.9330952441687859f; p->border_scale[k] = p->sigma_scale[k] * r.border_sad_mul; }   // j40.h:7466-7467
		for (int c = 0; c < 3; ++c) p->channel_scale[c] = r.channel_scale[c];
	}
	return 0;
}
static uint32_t decode_restored(j40hip_frame *h, uint8_t *rgba_dev, size_t stride_bytes, int mode, hipStream_t s) {
	j40hip_device_state *st = h->dev;
	const Frame &fr = h->frame;
	const FrameHeader::Restoration &r = fr.fh.restoration;
	const int32_t W = fr.fh.width, H = fr.fh.height;
	const size_t cells = (size_t) ((W + 7) / 8) * (size_t) ((H + 7) / 8), plane = (size_t) W * (size_t) H;
	RestoreParams p;
	uint32_t perr = restore_params(fr.fh, mode, &p);
	if (!perr && r.gab && W < 2) perr = ERR_TODO;   // (j40__gaborish reads sample 1 of every row, j40.h:7304)
	// the sharpness map: frame-wide, in the cell order of `blocks` (LfGroup after LfGroup); "shrp" as j40__epf_recip_sigmas finds it (j40.h:7399)
	std::vector<int16_t> sharp;
	if (!perr && r.epf_iters > 0) {
		sharp.reserve(cells);
		uint16_t ub = 0;
		for (const LfGroup &gg : fr.lf_groups) {
			if (gg.sharpness.size() != (size_t) gg.width8 * (size_t) gg.height8) { perr = ERR_TODO; break; }   // (a frame handle built without it: from a view or an LF bundle)
			for (int16_t v : gg.sharpness) ub |= (uint16_t) v;
			sharp.insert(sharp.end(), gg.sharpness.begin(), gg.sharpness.end());
		}
		if (!perr && !(ub < 8)) perr = ERR4('s', 'h', 'r', 'p');
	}
	if (perr) {   // the filters cannot run: the picture without them, and the complaint behind the sections' own (j40hip_frame_status)
		st->restore_err = perr;
		launch_vardct_frame(st->plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, rgba_dev, stride_bytes, s);
		return 0;
	}
	bool ok = true;
	if (!st->d_xyb) { st->d_xyb = st->scratch<float>(3 * plane, ok); st->d_xyb_tmp = st->scratch<float>(3 * plane, ok); st->d_sigma = st->scratch<float>(cells + 64, ok); }
	if (r.epf_iters > 0 && !st->d_sharp) st->d_sharp = st->upload(sharp.data(), sharp.size(), s, ok);
	if (!ok) return ERR_MEM;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	static const bool timed = getenv("J40HIP_RESTORATION_TIMING") != nullptr;
	launch_vardct_frame_xyb(st->plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, st->d_xyb, (size_t) W * 4, s);
	if (timed && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void) hipEventRecord(e0, s);
	uint32_t *sharp_or = (uint32_t *) (st->d_sigma + cells);   // (the device's own OR of the sharpness values: unused, the host checked)
	if (r.epf_iters > 0) {
		if (hipMemsetAsync(sharp_or, 0, 4, s) != hipSuccess) return ERR_GPU;
		launch_epf_sigma(st->plan, (int32_t) fr.lf_groups.size(), st->d_sharp, p, st->d_sigma, sharp_or, s);
	}
	st->d_restored = launch_restoration(st->d_xyb, st->d_xyb_tmp, (size_t) W, p, r.gab, r.epf_iters, st->d_sigma, s);
	if (e0 && e1) (void) hipEventRecord(e1, s);
	launch_xyb_to_rgba(st->d_restored, (size_t) W, st->plan.frame, W, H, rgba_dev, stride_bytes, s);
	st->restore_ran = mode;
	if (e0 && e1) { if (hipEventSynchronize(e1) == hipSuccess) (void) hipEventElapsedTime(&st->restore_ms, e0, e1); }
	if (e0) (void) hipEventDestroy(e0);
	if (e1) (void) hipEventDestroy(e1);
	return 0;
}

static uint32_t decode_impl(j40hip_frame *h, void *rgba_dev, size_t stride_bytes, hipStream_t s, float *ms3) {
	if (!h || !h->dev) return ERR_GPU;
	j40hip_device_state *st = h->dev;
	if (hipSetDevice(st->device) != hipSuccess) return ERR_GPU;
	if (st->is_modular) return decode_modular(h, rgba_dev, stride_bytes, s, ms3);
	const DevPlan &plan = st->plan;
	const Frame &fr = h->frame;
	const bool whole = st->first_group == 0 && st->num_groups == fr.fh.num_groups;
	st->trailers_pending = false;
	if (ms3) (void) hipEventRecord(st->ev[0], s);
	if (uint32_t e = clear_before_decode(st, s)) return e;
	if (hipMemsetAsync(plan.status, 0, sizeof(uint32_t) * (size_t) st->total_sections, s) != hipSuccess) return ERR_GPU;
	if (ms3) (void) hipEventRecord(st->ev[1], s);
	launch_hf_entropy(plan, st->hf, (int32_t) st->first_group, (int32_t) st->num_groups, s);
	if (ms3) (void) hipEventRecord(st->ev[2], s);
	st->restore_ran = 0; st->restore_err = 0;
	const int rmode = restoration_mode(h);
	if (rmode && whole && (fr.fh.restoration.gab || fr.fh.restoration.epf_iters > 0)) {
		// the restoration filters asked for and signalled: the pixel kernels leave the samples in XYB planes, Gaborish and the
		// edge-preserving filter run over the whole picture, the colour tail follows on the filtered planes (restore_kernels.h)
		if (uint32_t e = decode_restored(h, (uint8_t *) rgba_dev, stride_bytes, rmode, s)) return e;
	} else if (whole) {
		launch_vardct_frame(plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, (uint8_t *) rgba_dev, stride_bytes, s);
	} else {
		// sharded decode: only the varblocks of this process' groups
		launch_vardct_frame(plan, st->range_class_start, st->d_vb_range, st->d_large_scratch, (uint8_t *) rgba_dev, stride_bytes, s);
	}
	if (ms3) {
		(void) hipEventRecord(st->ev[3], s);
		if (hipEventSynchronize(st->ev[3]) != hipSuccess) return ERR_GPU;
		float a = 0, b = 0, c = 0;
		(void) hipEventElapsedTime(&a, st->ev[0], st->ev[1]);
		(void) hipEventElapsedTime(&b, st->ev[1], st->ev[2]);
		(void) hipEventElapsedTime(&c, st->ev[2], st->ev[3]);
		ms3[0] = b; ms3[1] = c; ms3[2] = a;
	}
	if (hipGetLastError() != hipSuccess) return ERR_GPU;
	if (st->has_trailers && whole) return validate_trailers(h, s);   // (synchronises `s`)
	return 0;
}

// ---- batches: throughput mode ----

struct j40hip_batch {
	int device = 0;
	std::vector<j40hip_frame *> frames;
	DevPlan *d_plans = nullptr;
	std::vector<DevPlan> plans_host;   // what d_plans holds (batch_enqueue re-uploads it when a member was uploaded again)
	std::vector<HfLaneWork> work_host;
	size_t plans_cap = 0, work_cap = 0;
	bool arrays_dirty = true;          // plans_host / work_host have not been copied to the device yet
	int side_in_use = 0;               // side streams the current membership spreads its pixel kernels over
	HfLaneWork *d_work = nullptr;
	int32_t num_work = 0;
	bool tables_in_lds = true;
	uint32_t lds_bytes = 0;
	int32_t waves_per_wg = 1;
	bool lanes_fast = true;          // every frame qualifies for k_hf_lanes
	uint32_t lanes_lds_bytes = 0;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	std::vector<hipEvent_t> slots;   // 4 events per recorded decode (j40hip_batch_decode_recorded)
	// the pixel kernels of different frames are independent and individually too small to fill the GPU: they are spread
	// over a few side streams that fork after the entropy launch and join before anything else runs on the caller's stream
	std::vector<hipStream_t> side;
	std::vector<hipEvent_t> side_done;
	hipEvent_t fork = nullptr;
};

extern "C" void j40hip_batch_free(j40hip_batch *b) {
	if (!b) return;
	(void) hipSetDevice(b->device);
	if (b->d_plans) (void) hipFree(b->d_plans);
	if (b->d_work) (void) hipFree(b->d_work);
	for (auto &e : b->ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : b->slots) if (e) (void) hipEventDestroy(e);
	for (auto &e : b->side_done) if (e) (void) hipEventDestroy(e);
	for (auto &st : b->side) if (st) (void) hipStreamDestroy(st);
	if (b->fork) (void) hipEventDestroy(b->fork);
	delete b;
}

// (Re)assigns the members of a batch: plans, the entropy kernel's work list and launch geometry. Device arrays are kept and only
// grown; their contents go up with the next decode (batch_enqueue), stream-ordered. The previous members' decodes must be complete.
static uint32_t batch_assign(j40hip_batch *b, j40hip_frame *const *frames, int64_t n) {
	if (n <= 0 || !frames) return ERR_RNGE;
	b->frames.clear(); b->plans_host.clear(); b->work_host.clear();
	b->tables_in_lds = true; b->lanes_fast = true; b->lanes_lds_bytes = 0; b->lds_bytes = 0;
	for (int64_t i = 0; i < n; ++i) {
		j40hip_frame *h = frames[i];
		if (!h || !h->dev) return ERR_GPU;
		if (h->dev->is_modular) return ERR_TODO;   // Modular frames: decode them one by one
		if (i == 0) b->device = h->dev->device;
		else if (h->dev->device != b->device) return ERR_RNGE;
		b->frames.push_back(h);
		b->plans_host.push_back(h->dev->plan);
		b->tables_in_lds = b->tables_in_lds && h->dev->hf.tables_fit_lds;
		b->lanes_fast = b->lanes_fast && h->dev->hf.lanes_fast;
		b->lanes_lds_bytes = std::max(b->lanes_lds_bytes, h->dev->hf.lanes_lds_bytes);
	}
	// Launch geometry of the entropy kernel. Sections per wavefront: 64 fills the lanes. Wavefronts per workgroup
	// share one copy of their frame's tables in LDS: with few wavefronts in the batch, one per workgroup spreads them
	// over the CUs; with many, sharing keeps the tables from capping the wavefronts a CU can hold.
	int32_t lanes = 64, total_waves = 0;
	if (const char *e = getenv("J40HIP_LANES_PER_WAVE")) lanes = std::max(1, std::min(64, atoi(e)));
	for (j40hip_frame *h : b->frames) total_waves += (h->frame.fh.num_groups + lanes - 1) / lanes;
	static int cus_of[16];   // (hipGetDeviceProperties takes milliseconds)
	if (b->device >= 0 && b->device < 16 && !cus_of[b->device]) { hipDeviceProp_t prop; cus_of[b->device] = hipGetDeviceProperties(&prop, b->device) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
	const int cus = b->device >= 0 && b->device < 16 ? cus_of[b->device] : 256;
	b->waves_per_wg = total_waves <= 2 * cus ? 1 : total_waves <= 4 * cus ? 2 : 4;
	if (const char *e = getenv("J40HIP_WAVES_PER_WG")) b->waves_per_wg = std::max(1, std::min(4, atoi(e)));
	std::vector<HfLaneWork> &work = b->work_host;
	for (size_t i = 0; i < b->frames.size(); ++i) {
		const int32_t groups = b->frames[i]->frame.fh.num_groups;
		for (int32_t g = 0; g < groups; g += lanes) work.push_back({(int32_t) i, g, std::min(lanes, groups - g), 0});
		while (work.size() % (size_t) b->waves_per_wg) work.push_back({(int32_t) i, 0, 0, 0});   // a workgroup stays on one frame
	}
	for (j40hip_frame *h : b->frames) {
		HfLaunchInfo info = h->dev->hf; info.tables_fit_lds = b->tables_in_lds;
		b->lds_bytes = std::max(b->lds_bytes, hf_lanes_lds_bytes(info));
	}
	b->num_work = (int32_t) work.size();
	if (hipSetDevice(b->device) != hipSuccess) return ERR_GPU;
	if (b->plans_host.size() > b->plans_cap) {
		if (b->d_plans) (void) hipFree(b->d_plans);
		b->plans_cap = b->plans_host.size() + b->plans_host.size() / 2;
		if (hipMalloc((void **) &b->d_plans, sizeof(DevPlan) * b->plans_cap) != hipSuccess) { b->d_plans = nullptr; b->plans_cap = 0; return ERR_GPU; }
	}
	if (work.size() > b->work_cap) {
		if (b->d_work) (void) hipFree(b->d_work);
		b->work_cap = work.size() + work.size() / 2;
		if (hipMalloc((void **) &b->d_work, sizeof(HfLaneWork) * b->work_cap) != hipSuccess) { b->d_work = nullptr; b->work_cap = 0; return ERR_GPU; }
	}
	b->arrays_dirty = true;
	// side streams for the pixel kernels: made once, as many as the largest membership so far asks for
	{
		int nside = (int) std::min<size_t>(16, b->frames.size());
		if (const char *e = getenv("J40HIP_SIDE_STREAMS")) nside = std::max(0, std::min(32, atoi(e)));
		if (nside < 2) nside = 0;
		while ((int) b->side.size() < nside) {
			hipStream_t st = nullptr; hipEvent_t ev = nullptr;
			if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ERR_GPU;
			b->side.push_back(st); b->side_done.push_back(ev);
		}
		b->side_in_use = nside;
		if (!b->fork && hipEventCreateWithFlags(&b->fork, hipEventDisableTiming) != hipSuccess) return ERR_GPU;
	}
	for (auto &e : b->ev) if (!e && hipEventCreate(&e) != hipSuccess) return ERR_GPU;
	return 0;
}

static j40hip_batch *batch_create_body(j40hip_frame *const *frames, int64_t n, uint32_t *err) {
	uint32_t dummy; if (!err) err = &dummy;
	j40hip_batch *b = new j40hip_batch();
	*err = batch_assign(b, frames, n);
	if (*err) { j40hip_batch_free(b); return nullptr; }
	return b;
}
extern "C" uint32_t j40hip_batch_reset(j40hip_batch *b, j40hip_frame *const *frames, int64_t n) {
	if (!b) return ERR_GPU;
	return guarded([&] { return batch_assign(b, frames, n); });
}

// ev: four events to record around the three stages (clear | entropy | pixels), or nullptr
static uint32_t batch_enqueue(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, hipStream_t s, hipEvent_t *ev) {
	if (!b) return ERR_GPU;
	if (hipSetDevice(b->device) != hipSuccess) return ERR_GPU;
	// a member that was uploaded again since the batch was made (j40hip_frame_force_dense + j40hip_frame_upload after "evof")
	// has a new plan in new blocks: the array the entropy kernel reads is brought up to date, stream-ordered behind the
	// launches of an earlier decode that may still be reading it. A member without device state fails the batch.
	{
		bool changed = false;
		for (size_t i = 0; i < b->frames.size(); ++i) {
			j40hip_device_state *st = b->frames[i]->dev;
			if (!st || st->is_modular || st->device != b->device) return ERR_GPU;
			if (memcmp(&b->plans_host[i], &st->plan, sizeof(DevPlan)) != 0) { b->plans_host[i] = st->plan; changed = true; }
		}
		if (changed || b->arrays_dirty) {
			// (pageable sources: the runtime copies them out before the calls return; the vectors live until the next reset anyway)
			if (hipMemcpyAsync(b->d_plans, b->plans_host.data(), sizeof(DevPlan) * b->plans_host.size(), hipMemcpyHostToDevice, s) != hipSuccess) return ERR_GPU;
			if (b->arrays_dirty && hipMemcpyAsync(b->d_work, b->work_host.data(), sizeof(HfLaneWork) * b->work_host.size(), hipMemcpyHostToDevice, s) != hipSuccess) return ERR_GPU;
			b->arrays_dirty = false;
		}
	}
	if (ev) (void) hipEventRecord(ev[0], s);
	for (j40hip_frame *h : b->frames) {
		j40hip_device_state *st = h->dev;
		st->trailers_pending = st->has_trailers;
		if (uint32_t e = clear_before_decode(st, s)) return e;
		// (no need to clear the status words: a batch decodes every section of every frame and the entropy kernels store
		// each section's status unconditionally -- 256 tiny fills were 4 % of a step)
	}
	if (ev) (void) hipEventRecord(ev[1], s);
	if (b->lanes_fast && !getenv("J40HIP_GENERIC_LANES")) launch_hf_lanes(b->d_plans, b->d_work, b->num_work, b->waves_per_wg, b->lanes_lds_bytes, s);
	else launch_hf_entropy_lanes(b->d_plans, b->d_work, b->num_work, b->tables_in_lds, b->lds_bytes, s);
	if (ev) (void) hipEventRecord(ev[2], s);
	if (b->side_in_use == 0) {
		for (size_t i = 0; i < b->frames.size(); ++i) {
			j40hip_device_state *st = b->frames[i]->dev;
			launch_vardct_frame(st->plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, (uint8_t *) rgba_dev[i], stride_bytes[i], s);
		}
	} else {
		if (hipEventRecord(b->fork, s) != hipSuccess) return ERR_GPU;
		for (int k = 0; k < b->side_in_use; ++k) if (hipStreamWaitEvent(b->side[(size_t) k], b->fork, 0) != hipSuccess) return ERR_GPU;
		for (size_t i = 0; i < b->frames.size(); ++i) {
			j40hip_device_state *st = b->frames[i]->dev;
			launch_vardct_frame(st->plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, (uint8_t *) rgba_dev[i], stride_bytes[i], b->side[i % (size_t) b->side_in_use]);
		}
		for (size_t k = 0; k < (size_t) b->side_in_use; ++k) {
			if (hipEventRecord(b->side_done[k], b->side[k]) != hipSuccess || hipStreamWaitEvent(s, b->side_done[k], 0) != hipSuccess) return ERR_GPU;
		}
	}
	if (ev) (void) hipEventRecord(ev[3], s);
	return hipGetLastError() == hipSuccess ? 0 : ERR_GPU;
}

static uint32_t events_to_ms(hipEvent_t *ev, float *ms3) {
	if (hipEventSynchronize(ev[3]) != hipSuccess) return ERR_GPU;
	float t0 = 0, t1 = 0, t2 = 0;
	(void) hipEventElapsedTime(&t0, ev[0], ev[1]); (void) hipEventElapsedTime(&t1, ev[1], ev[2]); (void) hipEventElapsedTime(&t2, ev[2], ev[3]);
	ms3[0] = t1; ms3[1] = t2; ms3[2] = t0;
	return 0;
}

static uint32_t batch_decode_impl(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, hipStream_t s, float *ms3) {
	if (!b) return ERR_GPU;
	if (uint32_t e = batch_enqueue(b, rgba_dev, stride_bytes, s, ms3 ? b->ev : nullptr)) return e;
	return ms3 ? events_to_ms(b->ev, ms3) : 0;
}

// asynchronous variant of the timed decode: records the stage events in `slot` and returns; the caller reads them
// with j40hip_batch_elapsed once the stream has been synchronised (keeps several batches in flight on different
// streams while still measuring every launch)
extern "C" uint32_t j40hip_batch_decode_recorded(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream, int32_t slot) {
	if (!b || slot < 0 || slot >= 4096) return ERR_RNGE;
	if (hipSetDevice(b->device) != hipSuccess) return ERR_GPU;
	while (b->slots.size() < 4 * ((size_t) slot + 1)) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return ERR_GPU; b->slots.push_back(e); }
	return guarded([&] { return batch_enqueue(b, rgba_dev, stride_bytes, (hipStream_t) stream, b->slots.data() + 4 * (size_t) slot); });
}
// makes `stream` wait until stage `stage` (1: cleared, 2: entropy decoded, 3: pixels written) of the decode recorded in
// `slot` has completed; used to stagger batches on different streams
extern "C" uint32_t j40hip_batch_wait_stage(j40hip_batch *b, int32_t slot, int32_t stage, void *stream) {
	if (!b || slot < 0 || stage < 0 || stage > 3 || b->slots.size() < 4 * ((size_t) slot + 1)) return ERR_RNGE;
	return hipStreamWaitEvent((hipStream_t) stream, b->slots[4 * (size_t) slot + (size_t) stage], 0) == hipSuccess ? 0 : ERR_GPU;
}
extern "C" uint32_t j40hip_batch_elapsed(j40hip_batch *b, int32_t slot, float *ms3) {
	if (!b || slot < 0 || b->slots.size() < 4 * ((size_t) slot + 1)) return ERR_RNGE;
	return events_to_ms(b->slots.data() + 4 * (size_t) slot, ms3);
}

extern "C" uint32_t j40hip_batch_decode(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream) {
	return guarded([&] { return batch_decode_impl(b, rgba_dev, stride_bytes, (hipStream_t) stream, nullptr); });
}
extern "C" uint32_t j40hip_batch_decode_timed(j40hip_batch *b, void *const *rgba_dev, const size_t *stride_bytes, void *stream, float *ms3) {
	return guarded([&] { return batch_decode_impl(b, rgba_dev, stride_bytes, (hipStream_t) stream, ms3); });
}

extern "C" uint32_t j40hip_frame_decode(j40hip_frame *h, void *rgba_dev, size_t stride_bytes, void *stream) {
	return guarded([&] { return decode_impl(h, rgba_dev, stride_bytes, (hipStream_t) stream, nullptr); });
}

extern "C" uint32_t j40hip_frame_decode_timed(j40hip_frame *h, void *rgba_dev, size_t stride_bytes, void *stream, float *ms3) {
	return guarded([&] { return decode_impl(h, rgba_dev, stride_bytes, (hipStream_t) stream, ms3); });
}

// The VarDCT status words reduced to the frame's verdict: the first failing section in the order the reference reads them
static uint32_t vardct_verdict(const j40hip_frame *h, const std::vector<uint32_t> &status) {
	const Frame &fr = h->frame;
	if (fr.toc.single) return status.empty() ? 0 : status[0];
	size_t best = SIZE_MAX; uint32_t code = 0;
	for (size_t i = 0; i < status.size(); ++i) if (status[i] && fr.toc.pass_groups[i].offset < best) { best = fr.toc.pass_groups[i].offset; code = status[i]; }
	return code;
}

static uint32_t j40hip_frame_status_body(j40hip_frame *h) {
	if (!h || !h->dev) return ERR_GPU;
	j40hip_device_state *st = h->dev;
	if (st->is_modular) {
		st->status_host.assign((size_t) st->total_sections + 1, 0);
		if (hipMemcpy(st->status_host.data(), st->mod.status, sizeof(uint32_t) * st->status_host.size(), hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
		std::vector<std::pair<size_t, uint32_t>> bad;
		for (size_t i = 0; i < (size_t) st->total_sections; ++i) if (st->status_host[i]) bad.push_back({st->mod_section_offsets[i], st->status_host[i]});
		if (!bad.empty()) return std::min_element(bad.begin(), bad.end())->second;
		return st->status_host[(size_t) st->total_sections];
	}
	if (st->trailers_pending) {
		// the frame was last decoded by a batch (asynchronous: it cannot stop for the host in the middle): the extra channels'
		// sub-images behind the coefficients are checked now, so that both modes report damage in them like the reference
		st->trailers_pending = false;
		if (hipSetDevice(st->device) != hipSuccess) return ERR_GPU;
		if (uint32_t e = validate_trailers(h, nullptr)) return e;
	}
	st->status_host.assign((size_t) st->total_sections, 0);
	if (hipMemcpy(st->status_host.data(), st->plan.status, sizeof(uint32_t) * st->status_host.size(), hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
	// the reference reports the first failing section in the order it reads them (TOC order)
	const Frame &fr = h->frame;
	if (fr.toc.single) return st->status_host.empty() ? st->restore_err : st->status_host[0] ? st->status_host[0] : st->restore_err;
	std::vector<std::pair<size_t, uint32_t>> bad;
	for (size_t i = 0; i < st->status_host.size(); ++i) if (st->status_host[i]) bad.push_back({fr.toc.pass_groups[i].offset, st->status_host[i]});
	if (bad.empty()) return st->restore_err;   // (the filters run behind the last section: their complaint comes after every section's)
	return std::min_element(bad.begin(), bad.end())->second;
}

// ---- the single-image path in two phases (round 6) ----
// One image alone is its longest section: every section has a wavefront and a SIMD of its own, the entropy launch lasts as long as the
// longest of them (1.9 x the mean in the bench's 8K frame), and 133 MB of pixels then take 2.3 ms over the link while the device has
// nothing left to do. Here the few longest sections -- those within the copy's time of the longest -- are decoded by a launch of their
// own on a second stream, beside the launch of all the others; when THAT one is through the pixel kernels run over the whole frame
// (the long sections' blocks, whose block_events entries are still zero, come out as their LF only) and the whole image starts over
// the link; when the long sections are through their block_events entries -- written to a table of their own meanwhile, so that the
// first pass never sees one half written -- are merged in, the pixel kernels run again (0.3 ms; same pixels everywhere else), and
// only the long sections' groups, a 256 x 256 rectangle each, follow the image over the link. Same pixels and codes as the one-phase
// decode (tests/test_gpu_parity.py runs both); J40HIP_TWO_PHASE=0: never.
struct TwoPhaseStream {   // a second stream and three events on a device, borrowed for one decode (hipStreamCreate is not for the path of one image)
	int device = -1; hipStream_t s = nullptr; hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};
static std::mutex g_two_phase_mutex;
static std::vector<TwoPhaseStream> g_two_phase_idle;   // (handed back after every decode; j40hip_shutdown destroys them)
static bool two_phase_borrow(int dev, TwoPhaseStream *out) {
	{
		std::lock_guard<std::mutex> lock(g_two_phase_mutex);
		for (size_t i = 0; i < g_two_phase_idle.size(); ++i) if (g_two_phase_idle[i].device == dev) { *out = g_two_phase_idle[i]; g_two_phase_idle.erase(g_two_phase_idle.begin() + (long) i); return true; }
	}
	TwoPhaseStream t;
	t.device = dev;
	if (hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); return false; }
	for (auto &e : t.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
		(void) hipGetLastError();
		for (auto &d : t.ev) if (d) (void) hipEventDestroy(d);
		(void) hipStreamDestroy(t.s);
		return false;
	}
	*out = t;
	return true;
}
static void two_phase_return(const TwoPhaseStream &t) { std::lock_guard<std::mutex> lock(g_two_phase_mutex); g_two_phase_idle.push_back(t); }
static void two_phase_shutdown() {
	std::vector<TwoPhaseStream> all;
	{ std::lock_guard<std::mutex> lock(g_two_phase_mutex); all.swap(g_two_phase_idle); }
	for (TwoPhaseStream &t : all) { if (hipSetDevice(t.device) != hipSuccess) continue; (void) hipStreamDestroy(t.s); for (auto &e : t.ev) if (e) (void) hipEventDestroy(e); }
}

// decides once per upload whether the frame is decoded in two phases and with which groups on the second stream (st->two_k of st->two_order)
static void two_phase_plan(j40hip_frame *h, size_t image_bytes) {
	j40hip_device_state *st = h->dev;
	st->two_k = 0;
	const char *env = getenv("J40HIP_TWO_PHASE");   // (looked at per upload: tests switch it between frames)
	const bool allowed = !env || atoi(env) != 0;
	const Frame &fr = h->frame;
	const int64_t ng = fr.fh.num_groups;
	if (!allowed || st->is_modular || !st->plan.events || !st->plan.block_events || st->has_trailers || fr.toc.single || fr.fh.num_passes != 1 || ng < 64 || image_bytes < ((size_t) 16 << 20)) return;
	if (st->first_group != 0 || st->num_groups != ng || fr.toc.pass_groups.size() != (size_t) ng) return;
	if (!hf_entropy_fast_path(st->plan, st->hf)) return;
	std::vector<uint32_t> order((size_t) ng);
	for (int64_t g = 0; g < ng; ++g) order[(size_t) g] = (uint32_t) g;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return fr.toc.pass_groups[a].size > fr.toc.pass_groups[b].size; });
	// a section takes about a microsecond a byte on its wavefront, the link 57 GB/s: the sections within the copy's time of the longest
	const double copy_us = (double) image_bytes / 57e3, longest = (double) fr.toc.pass_groups[order[0]].size;
	int32_t k = 0;
	while (k < (int32_t) ng && k < 64 && (double) fr.toc.pass_groups[order[(size_t) k]].size > longest - 1.25 * copy_us) ++k;
	if (k < 1 || k > ng / 4) return;
	// the order and the long sections' table: one recycled block (a hipMalloc of 16 MB is a millisecond, on the path of one image)
	const size_t order_bytes = ((size_t) ng * 4 + 255) & ~(size_t) 255, shadow_bytes = 16 * st->num_blocks;
	bool clean = false;
	st->two_block = cache_acquire(st->device, order_bytes + shadow_bytes, &st->two_block_bytes, &clean);
	if (!st->two_block) return;
	st->d_two_order = (uint32_t *) st->two_block; st->d_two_shadow = (uint32_t *) ((uint8_t *) st->two_block + order_bytes);
	if (hipMemcpy(st->d_two_order, order.data(), (size_t) ng * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemsetAsync(st->d_two_shadow, 0, shadow_bytes, nullptr) != hipSuccess) { (void) hipGetLastError(); return; }   // (the fill: ahead of the decode's launches on its stream)
	st->two_order = std::move(order);
	st->two_k = k;
}

// pixels of a whole VarDCT frame into rgba_host; d: the device image (stride_bytes per row). Returns the frame's code like decode_impl +
// j40hip_frame_status would; *done = false: nothing was enqueued, the caller decodes the usual way.
static uint32_t decode_two_phase(j40hip_frame *h, uint8_t *d, uint8_t *rgba_host, size_t stride_bytes, bool *done) {
	*done = false;
	j40hip_device_state *st = h->dev;
	const Frame &fr = h->frame;
	const size_t bytes = stride_bytes * (size_t) fr.fh.height;
	if (st->two_k < 0) two_phase_plan(h, bytes);
	if (st->two_k <= 0 || restoration_mode(h) != 0 || st->first_group != 0 || st->num_groups != fr.fh.num_groups) return 0;   // (a group range set since: the usual way)
	TwoPhaseStream tp;
	if (!two_phase_borrow(st->device, &tp)) return 0;
	struct GiveBack { const TwoPhaseStream &t; ~GiveBack() { (void) hipStreamSynchronize(t.s); two_phase_return(t); } } give_back{tp};   // (whatever way the decode ends: nothing of it is left on the stream)
	*done = true;
	static const bool timing = getenv("J40HIP_API_TIMING") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = timing ? now() : 0;
	const DevPlan &plan = st->plan;
	const int32_t ng = (int32_t) fr.fh.num_groups, k = st->two_k;
	hipStream_t s0 = nullptr, s1 = tp.s;
	st->trailers_pending = false; st->restore_ran = 0; st->restore_err = 0;
	if (hipMemsetAsync(plan.status, 0, sizeof(uint32_t) * (size_t) st->total_sections, s0) != hipSuccess) return ERR_GPU;
	if (hipEventRecord(tp.ev[0], s0) != hipSuccess || hipStreamWaitEvent(s1, tp.ev[0], 0) != hipSuccess) return ERR_GPU;
	DevPlan plan_long = plan;
	plan_long.block_events = st->d_two_shadow;
	launch_hf_entropy_fast_ordered(plan_long, st->hf, st->d_two_order, 0, k, s1);          // the long sections, on their own
	launch_hf_entropy_fast_ordered(plan, st->hf, st->d_two_order, k, ng - k, s0);           // all the others
	launch_vardct_frame(plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, d, stride_bytes, s0);
	if (hipEventRecord(tp.ev[1], s0) != hipSuccess || hipStreamWaitEvent(s1, tp.ev[1], 0) != hipSuccess) return ERR_GPU;
	launch_merge_block_events(plan, st->d_two_order, k, st->d_two_shadow, s1);
	launch_vardct_frame(plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, d, stride_bytes, s1);
	if (hipEventRecord(tp.ev[2], s1) != hipSuccess) return ERR_GPU;
	if (hipGetLastError() != hipSuccess) return ERR_GPU;
	// the image of the first pass over the link while the long sections are still being decoded: the copy the one-phase decode issues,
	// behind the first pass on its stream (the host waits in it)
	const double t1 = timing ? now() : 0;
	if (hipEventSynchronize(tp.ev[1]) != hipSuccess) return ERR_GPU;
	const double t2 = timing ? now() : 0;
	if (!hostcopy_d2h_sync(st->device, rgba_host, d, bytes) && hipMemcpy(rgba_host, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
	const double t3 = timing ? now() : 0;
	// the long sections' groups, one rectangle each, on top: written by a kernel where the device can reach the host's image (pinned
	// memory: the public API's planes) -- a 2-D copy per rectangle is 0.1 ms each --, else copied one by one
	uint8_t *mapped = nullptr;
	{
		hipPointerAttribute_t at;
		if (hipPointerGetAttributes(&at, rgba_host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) mapped = (uint8_t *) at.devicePointer;
		else (void) hipGetLastError();
	}
	const int32_t shift = fr.fh.group_size_shift, gdim = 1 << shift;
	if (mapped) launch_store_group_rects(st->d_two_order, k, fr.fh.gcolumns, shift, fr.fh.width, fr.fh.height, d, mapped, stride_bytes, s1);
	else {
		if (hipEventSynchronize(tp.ev[2]) != hipSuccess) return ERR_GPU;
		for (int32_t i = 0; i < k; ++i) {
			const int64_t g = st->two_order[(size_t) i], gx = g % fr.fh.gcolumns, gy = g / fr.fh.gcolumns;
			const size_t x0 = (size_t) gx << shift, y0 = (size_t) gy << shift;
			const size_t w = std::min<size_t>((size_t) gdim, (size_t) fr.fh.width - x0), rows = std::min<size_t>((size_t) gdim, (size_t) fr.fh.height - y0);
			const size_t off = y0 * stride_bytes + x0 * 4;
			if (hipMemcpy2DAsync(rgba_host + off, stride_bytes, d + off, stride_bytes, w * 4, rows, hipMemcpyDeviceToHost, s1) != hipSuccess) return ERR_GPU;
		}
	}
	const double t4 = timing ? now() : 0;
	if (hipStreamSynchronize(s1) != hipSuccess) return ERR_GPU;
	if (timing) fprintf(stderr, "[j40hip two phases] %d long sections of %d: enqueued %.2f ms, the others + first pass through after %.2f, image over the link %.2f, long sections + second pass + %d rectangles (%s) another %.2f ms\n",
		k, ng, t1 - t0, t2 - t1, t3 - t2, k, mapped ? "a kernel's stores" : "2-D copies", now() - t3);
	(void) t4;
	return j40hip_frame_status(h);
}

static uint32_t j40hip_frame_decode_to_host_body(j40hip_frame *h, void *rgba_host, size_t stride_bytes) {
	if (!h || !h->dev) return ERR_GPU;
	const Frame &fr = h->frame;
	const int device = h->dev->device;
	if (hipSetDevice(device) != hipSuccess) return ERR_GPU;
	// the device image uses the caller's row stride, so one contiguous copy brings it back
	const size_t bytes = stride_bytes * (size_t) fr.fh.height;
	size_t got = 0; bool clean = false;
	void *d = cache_acquire(device, bytes, &got, &clean);
	if (!d) return ERR_GPU;
	bool two_phase = false;
	uint32_t err = decode_two_phase(h, (uint8_t *) d, (uint8_t *) rgba_host, stride_bytes, &two_phase);
	if (two_phase && err != ERR_EVOF) {   // (the pixels are in rgba_host, or the frame has failed; "evof": the dense form below)
		(void) hipDeviceSynchronize();
		cache_release(device, d, got, false);
		return err;
	}
	if (two_phase) (void) hipDeviceSynchronize();
	err = decode_impl(h, d, stride_bytes, nullptr, nullptr);
	if (!err && hipStreamSynchronize(nullptr) != hipSuccess) err = ERR_GPU;
	if (!err) err = j40hip_frame_status(h);
	if (err == ERR_EVOF) {   // a section with more non-zero coefficients than its event region holds: decode with dense planes
		h->force_dense = true;
		err = j40hip_frame_upload(h, device);
		if (!err) err = decode_impl(h, d, stride_bytes, nullptr, nullptr);
		if (!err && hipStreamSynchronize(nullptr) != hipSuccess) err = ERR_GPU;
		if (!err) err = j40hip_frame_status(h);
	}
	// (the decode has been waited for: the copy may go to the SDMA engine a pipeline of this process measured, hostcopy.hpp)
	if (!err && !hostcopy_d2h_sync(device, rgba_host, d, bytes) && hipMemcpy(rgba_host, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) err = ERR_GPU;
	(void) hipDeviceSynchronize();
	cache_release(device, d, got, false);
	return err;
}

extern "C" void j40hip_frame_set_restoration(j40hip_frame *h, int mode) { if (h) h->restoration = mode < 0 ? -1 : mode > 2 ? 2 : mode; }
extern "C" void j40hip_frame_restoration(const j40hip_frame *h, j40hip_restoration *out) {
	if (!h || !out) return;
	const FrameHeader::Restoration &r = h->frame.fh.restoration;
	out->gab_enabled = r.gab ? 1 : 0;
	for (int c = 0; c < 3; ++c) for (int j = 0; j < 2; ++j) out->gab_weights[c][j] = r.gab_weights[c][j];
	out->epf_iters = r.epf_iters;
	for (int i = 0; i < 8; ++i) out->epf_sharp_lut[i] = r.sharp_lut[i];
	for (int c = 0; c < 3; ++c) out->epf_channel_scale[c] = r.channel_scale[c];
	out->epf_quant_mul = r.quant_mul; out->epf_pass0_sigma_scale = r.pass0_sigma_scale; out->epf_pass2_sigma_scale = r.pass2_sigma_scale;
	out->epf_border_sad_mul = r.border_sad_mul; out->epf_sigma_for_modular = r.sigma_for_modular;
}
// the sharpness map of LfGroup gg as decoded (i16 w8*h8), like j40hip_frame_lf_group_plane's planes
extern "C" int j40hip_frame_sharpness(const j40hip_frame *h, int64_t gg, int16_t *out) {
	if (!h || gg < 0 || (size_t) gg >= h->frame.lf_groups.size()) return -1;
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	if (g.sharpness.size() != (size_t) g.width8 * (size_t) g.height8) return -1;
	memcpy(out, g.sharpness.data(), g.sharpness.size() * 2);
	return 0;
}
// after a decode that ran the filters (synchronised): stage 0 the samples as the inverse transforms left them, 1 the filtered ones --
// three planes of width * height floats (X, Y, B); stage 2: the reciprocal-sigma plane (w8 * h8 floats)
extern "C" uint32_t j40hip_frame_read_xyb(j40hip_frame *h, int stage, float *out) {
	if (!h || !h->dev || !h->dev->restore_ran || !h->dev->d_xyb) return ERR_RNGE;
	j40hip_device_state *st = h->dev;
	const FrameHeader &fh = h->frame.fh;
	const size_t plane = (size_t) fh.width * (size_t) fh.height, cells = (size_t) ((fh.width + 7) / 8) * (size_t) ((fh.height + 7) / 8);
	if (hipSetDevice(st->device) != hipSuccess) return ERR_GPU;
	if (stage == 2) return hipMemcpy(out, st->d_sigma, cells * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : ERR_GPU;
	if (stage == 1) return hipMemcpy(out, st->d_restored, 3 * plane * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : ERR_GPU;
	// stage 0: the planes the pixel kernels wrote are the filters' first input; they survive only when the result lies in the other buffer
	// pair at every step's end -- re-run the pixel kernels into the spare buffer instead
	const float *src = st->d_xyb;
	const FrameHeader::Restoration &r = fh.restoration;
	const int steps = (r.gab ? 1 : 0) + (r.epf_iters >= 3 ? 3 : r.epf_iters);
	if (steps >= 2) {   // d_xyb has been written over by the second step: once more, into whichever buffer the result does not occupy
		float *spare = st->d_restored == st->d_xyb ? st->d_xyb_tmp : st->d_xyb;
		launch_vardct_frame_xyb(st->plan, st->class_start, st->d_vb_sorted, st->d_large_scratch, spare, (size_t) fh.width * 4, nullptr);
		if (hipStreamSynchronize(nullptr) != hipSuccess) return ERR_GPU;
		src = spare;
	}
	return hipMemcpy(out, src, 3 * plane * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : ERR_GPU;
}
extern "C" float j40hip_frame_restoration_ms(const j40hip_frame *h) { return h && h->dev ? h->dev->restore_ms : 0.0f; }
// known-answer hook: the filter kernels on caller-supplied planes ([3][h][w] floats, in place), a w8*h8 sharpness map and the HfMul
// reciprocal of the varblock covering each cell; mode 1 / 2 as j40hip_frame_set_restoration; sigma_out (optional): w8*h8 floats
extern "C" uint32_t j40hip_kat_device_restoration(float *xyb, int32_t w, int32_t h, const int16_t *sharpness, const float *hfmul_inv, const j40hip_restoration *r, int mode, int device, float *sigma_out) {
	return guarded([&]() -> uint32_t {
		if (!xyb || !r || w < 1 || h < 1 || j40hip_device_count() <= device || hipSetDevice(device) != hipSuccess) return ERR_GPU;
		if (!ensure_constant_tables(device)) return ERR_GPU;
		FrameHeader fh;
		fh.width = w; fh.height = h;
		fh.restoration.gab = r->gab_enabled != 0;
		for (int c = 0; c < 3; ++c) for (int j = 0; j < 2; ++j) fh.restoration.gab_weights[c][j] = r->gab_weights[c][j];
		fh.restoration.epf_iters = r->epf_iters;
		for (int i = 0; i < 8; ++i) fh.restoration.sharp_lut[i] = r->epf_sharp_lut[i];
		for (int c = 0; c < 3; ++c) fh.restoration.channel_scale[c] = r->epf_channel_scale[c];
		fh.restoration.quant_mul = r->epf_quant_mul; fh.restoration.pass0_sigma_scale = r->epf_pass0_sigma_scale; fh.restoration.pass2_sigma_scale = r->epf_pass2_sigma_scale;
		fh.restoration.border_sad_mul = r->epf_border_sad_mul;
		RestoreParams p;
		if (uint32_t e = restore_params(fh, mode, &p)) return e;
		if (fh.restoration.gab && w < 2) return ERR_TODO;
		const size_t plane = (size_t) w * (size_t) h, cells = (size_t) p.w8 * (size_t) p.h8;
		if (r->epf_iters > 0) { uint16_t ub = 0; for (size_t i = 0; i < cells; ++i) ub |= (uint16_t) sharpness[i]; if (!(ub < 8)) return ERR4('s', 'h', 'r', 'p'); }
		j40hip_device_state tmp; tmp.device = device;
		bool ok = true;
		float *d_a = tmp.upload(xyb, 3 * plane, nullptr, ok), *d_b = tmp.scratch<float>(3 * plane, ok), *d_sigma = tmp.scratch<float>(cells + 64, ok);
		if (ok && r->epf_iters > 0) {
			int16_t *d_sh = tmp.upload(sharpness, cells, nullptr, ok);
			float *d_hf = tmp.upload(hfmul_inv, cells, nullptr, ok);
			if (ok) { (void) hipMemsetAsync(d_sigma + cells, 0, 4, nullptr); launch_epf_sigma_cells(d_sh, d_hf, p, d_sigma, (uint32_t *) (d_sigma + cells), nullptr); }
		}
		if (ok) {
			const float *res = launch_restoration(d_a, d_b, (size_t) w, p, fh.restoration.gab, r->epf_iters, d_sigma, nullptr);
			ok = hipMemcpy(xyb, res, 3 * plane * 4, hipMemcpyDeviceToHost) == hipSuccess;
			if (ok && sigma_out && r->epf_iters > 0) ok = hipMemcpy(sigma_out, d_sigma, cells * 4, hipMemcpyDeviceToHost) == hipSuccess;
		}
		(void) hipDeviceSynchronize();
		for (auto &b : tmp.buffers) b.release();
		tmp.buffers.clear();
		return ok ? 0 : ERR_GPU;
	});
}

extern "C" uint32_t j40hip_frame_read_coeffs(j40hip_frame *h, int64_t gg, int c, float *out) {
	if (!h || !h->dev || h->dev->is_modular) return ERR_GPU;
	j40hip_device_state *st = h->dev;
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	size_t base = 0;
	for (int64_t i = 0; i < gg; ++i) base += h->frame.lf_groups[(size_t) i].blocks.size();
	if (!st->plan.events) {   // dense planes, canonical order
		return hipMemcpy(out, st->plan.coeffs[c] + base * 64, sizeof(float) * g.blocks.size() * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : ERR_GPU;
	}
	// sparse: expand the events of this LF group's blocks into the canonical layout the reference keeps
	std::vector<uint32_t> table(4 * st->num_blocks);
	if (hipMemcpy(table.data(), st->plan.block_events, sizeof(uint32_t) * table.size(), hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
	memset(out, 0, sizeof(float) * g.blocks.size() * 64);
	std::vector<CoeffEvent> ev;
	if (!host_vb_sorted(st)) return ERR_GPU;
	for (const DevVarblock &vb : st->vb_sorted) {
		if ((size_t) vb.llf_base < base || (size_t) vb.llf_base >= base + g.blocks.size()) continue;   // another LF group's block
		const uint32_t *be = table.data() + 4 * (size_t) vb.blk;
		const uint32_t skip = c == 1 ? 0 : c == 0 ? be[1] : be[1] + be[2], n = be[c == 1 ? 1 : c == 0 ? 2 : 3];   // emission order Y, X, B
		if (!n) continue;
		ev.resize(n);
		if (hipMemcpy(ev.data(), st->plan.events + be[0] + skip, sizeof(CoeffEvent) * n, hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
		const std::vector<int32_t> &order = h->frame.orders[0][DCT_SELECT[vb.dctsel].order_idx][(size_t) c];
		float *blk = out + ((size_t) vb.llf_base - base) * 64;
		for (const CoeffEvent &e : ev) blk[order[coeff_event_pos(e)]] = (float) coeff_event_value(e);
	}
	return 0;
}

extern "C" uint32_t j40hip_frame_read_plane_i16(j40hip_frame *h, int c, int16_t *out) {
	if (!h || !h->dev || !h->dev->is_modular) return ERR_GPU;
	j40hip_device_state *st = h->dev;
	if (c < 0 || (size_t) c >= st->final_planes.size()) return ERR_RNGE;
	const size_t n = (size_t) st->final_w[(size_t) c] * (size_t) st->final_h[(size_t) c];
	if (hipMemcpy(out, st->final_planes[(size_t) c], n * 2, hipMemcpyDeviceToHost) != hipSuccess) return ERR_GPU;
	return 0;
}

extern "C" uint32_t j40hip_kat_device_srgb_u8(const float *v_host, size_t n, uint8_t *out_host) {
	if (j40hip_device_count() <= 0) return ERR_GPU;
	float *dv = nullptr; uint8_t *dout = nullptr;
	bool ok = hipMalloc((void **) &dv, n * 4 + 16) == hipSuccess && hipMalloc((void **) &dout, n + 16) == hipSuccess;
	ok = ok && hipMemcpy(dv, v_host, n * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) { int dev = 0; ok = hipGetDevice(&dev) == hipSuccess && ensure_constant_tables(dev); if (ok) launch_kat_srgb_u8(dv, n, dout, nullptr); }
	ok = ok && hipMemcpy(out_host, dout, n, hipMemcpyDeviceToHost) == hipSuccess;
	if (dv) (void) hipFree(dv);
	if (dout) (void) hipFree(dout);
	return ok ? 0 : ERR_GPU;
}

// ---- the guarded entry points of the functions above ----
extern "C" uint32_t j40hip_frame_upload(j40hip_frame *h, int device) { return guarded([&] { return j40hip_frame_upload_body(h, device); }); }
extern "C" uint32_t j40hip_frame_set_group_range(j40hip_frame *h, int64_t first_group, int64_t num_groups) { return guarded([&] { return j40hip_frame_set_group_range_body(h, first_group, num_groups); }); }
extern "C" uint32_t j40hip_frame_status(j40hip_frame *h) { return guarded([&] { return j40hip_frame_status_body(h); }); }
extern "C" int32_t j40hip_frame_two_phase_sections(const j40hip_frame *h) { return h && h->dev ? h->dev->two_k : -1; }
extern "C" uint32_t j40hip_frame_decode_to_host(j40hip_frame *h, void *rgba_host, size_t stride_bytes) { return guarded([&] { return j40hip_frame_decode_to_host_body(h, rgba_host, stride_bytes); }); }
extern "C" j40hip_batch *j40hip_batch_create(j40hip_frame *const *frames, int64_t n, uint32_t *err) {
	try { return batch_create_body(frames, n, err); } catch (const std::exception &) { if (err) *err = ERR_MEM; return nullptr; }
}
extern "C" uint32_t j40hip_frame_upload_on(j40hip_frame *h, int device, void *stream) { return guarded([&] { return upload_impl(h, device, (hipStream_t) stream); }); }
// Takes the library's process-wide state down: stops and joins the LfGroup service threads, gives the cached device memory back.
// No other call into the library may be running or follow on objects created before. Optional: a process may also just end.
extern "C" void j40hip_shutdown(void) {
	std::vector<LfService *> services;
	{ std::lock_guard<std::mutex> lock(g_lf_service_mutex); for (LfService *&sv : g_lf_services) if (sv) { services.push_back(sv); sv = nullptr; } }
	for (LfService *sv : services) {
		{ std::lock_guard<std::mutex> lock(sv->m); sv->stop = true; }
		sv->cv_work.notify_all();
		if (sv->thread.joinable()) sv->thread.join();
		delete sv;
	}
	j40hip_serve_shutdown();
	j40hip_async_shutdown();
	two_phase_shutdown();
	hostcopy_shutdown();
	pinned_trim();
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); n = 0; }
	for (int d = 0; d < n && d < 16; ++d) if (hipSetDevice(d) == hipSuccess) { (void) hipDeviceSynchronize(); cache_trim(d); }
	j40hip_thread_release();
}
extern "C" void j40hip_thread_release(void) { t_stage.release(); t_lf_out.release(); if (t_lf_done) { (void) hipEventDestroy(t_lf_done); t_lf_done = nullptr; } t_host_plan = HostPlan(); }

// j40hip_frame_status in two halves for pipelines: `begin` enqueues the copy of the status words on `stream` (no host wait),
// `end` -- after the caller has waited for that stream -- reduces them to the frame's verdict. VarDCT frames without extra
// channels only (others: ERR_TODO; use j40hip_frame_status).
extern "C" uint32_t j40hip_frame_status_begin(j40hip_frame *h, void *stream) {
	if (!h || !h->dev) return ERR_GPU;
	j40hip_device_state *st = h->dev;
	if (st->is_modular || st->has_trailers) return ERR_TODO;
	st->status_host.assign((size_t) st->total_sections, 0);
	return hipMemcpyAsync(st->status_host.data(), st->plan.status, sizeof(uint32_t) * st->status_host.size(), hipMemcpyDeviceToHost, (hipStream_t) stream) == hipSuccess ? 0 : ERR_GPU;
}
extern "C" uint32_t j40hip_frame_status_end(j40hip_frame *h) {
	if (!h || !h->dev || h->dev->status_host.size() != (size_t) h->dev->total_sections) return ERR_GPU;
	return vardct_verdict(h, h->dev->status_host);
}
extern "C" void j40hip_frame_mark_idle(j40hip_frame *h) { if (h && h->dev) h->dev->idle = true; }
