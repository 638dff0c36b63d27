// j40_amd/csrc/device/async.hip -- see async.hpp. The reference work this replaces per frame, stage by stage:
//   host (here)      container, headers, TOC, LfGlobal, HfGlobal (j40.h:8175-8192), the first bits of every LfGroup section
//   k_lf_groups      the two Modular sub-images of every LfGroup section (j40.h:6722-6790) -- or the host decoder, when the
//                    frame's tree / code spec are outside what the kernel takes, or the pipeline sends the frame that way
//   plan_kernels     LF index, varblock placement, work lists (j40.h:6566-6570, 6634-6701; plan_build.cpp)
//   lf_tail_kernels  LF dequantisation, smoothing, LLF coefficients (j40.h:6544-6590, 6492, 5944)
//   k_hf_lanes       j40__pass_group / j40__hf_coeffs of every section (j40.h:7007, 6888)
//   k_vardct_*       dequantisation, chroma-from-luma, inverse transforms, XYB -> sRGB, pack (j40.h:7053-7247, 7910)
//   k_plan_verdict   the first failing section in file order
// Nothing here touches the oracle, and there is no CPU fallback for the hot path: without a HIP device prepare returns nullptr
// and the caller's single-frame path reports "!gpu".
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <memory>
#include <mutex>
#include <vector>
#include "../capi.hpp"
#include "../plan_front.hpp"
#include "kernels.h"
#include "runtime_shared.hpp"
#include "hostcopy.hpp"
#include "async.hpp"

using namespace j40hip;
using namespace j40hip_rt;

namespace {

constexpr uint32_t ERR_GPU = ('!' << 24) | ('g' << 16) | ('p' << 8) | 'u';
constexpr uint32_t ERR_MEM = ('!' << 24) | ('m' << 16) | ('e' << 8) | 'm';

// ---- the static tables on the device, one copy per distinct encoding (plan_front.hpp) ----
struct StaticEntry {
	int device = 0;
	std::vector<uint8_t> key;
	uint32_t order_off[11 * 13 * 3], dq_off[17], dq_size[17], dq_scan_off[17];
	bool any_dq_error = false;
	float *d_f32 = nullptr; uint16_t *d_u16 = nullptr;
	uint64_t last_use = 0;
	~StaticEntry() { if (d_f32 || d_u16) { (void) hipSetDevice(device); if (d_f32) (void) hipFree(d_f32); if (d_u16) (void) hipFree(d_u16); } }
};
std::mutex g_static_mutex;
std::vector<std::shared_ptr<StaticEntry>> &g_static = *new std::vector<std::shared_ptr<StaticEntry>>();   // (on the heap: no hipFree from a static destructor at process exit; j40hip_shutdown empties it)
uint64_t g_static_clock = 0;

std::shared_ptr<StaticEntry> static_tables_for(const Frame &fr, int device) {
	std::vector<uint8_t> key;
	static_tables_key(fr, &key);
	std::lock_guard<std::mutex> lock(g_static_mutex);
	for (auto &e : g_static) if (e->device == device && e->key == key) { e->last_use = ++g_static_clock; return e; }
	// (built under the lock: the first frames of a run all want the same entry, and the others should find it rather than build it too)
	StaticTables st;
	build_static_tables(fr, &st);
	auto e = std::make_shared<StaticEntry>();
	e->device = device; e->key.swap(st.key);
	memcpy(e->order_off, st.order_off, sizeof e->order_off); memcpy(e->dq_off, st.dq_off, sizeof e->dq_off);
	memcpy(e->dq_size, st.dq_size, sizeof e->dq_size); memcpy(e->dq_scan_off, st.dq_scan_off, sizeof e->dq_scan_off);
	for (int i = 0; i < 17; ++i) e->any_dq_error = e->any_dq_error || st.dq_error[i] != 0;
	if (hipMalloc((void **) &e->d_f32, st.pool_f32.size() * sizeof(float) + 64) != hipSuccess || hipMalloc((void **) &e->d_u16, st.pool_u16.size() * 2 + 64) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	if (hipMemcpy(e->d_f32, st.pool_f32.data(), st.pool_f32.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
	if (hipMemcpy(e->d_u16, st.pool_u16.data(), st.pool_u16.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
	e->last_use = ++g_static_clock;
	if (g_static.size() >= 16) {   // drop the least recently used entry nobody holds
		size_t victim = g_static.size();
		for (size_t i = 0; i < g_static.size(); ++i) if (g_static[i].use_count() == 1 && (victim == g_static.size() || g_static[i]->last_use < g_static[victim]->last_use)) victim = i;
		if (victim < g_static.size()) g_static.erase(g_static.begin() + (long) victim);
	}
	g_static.push_back(e);
	return e;
}

// This thread's pinned staging buffers. A buffer is free again once the copy out of it has completed -- which can take long after
// the call returned (the copy waits its turn behind whatever occupies the stream's hardware queue) -- and the thread must never wait
// for the device: it takes the first free buffer, and makes another one when none is free.
struct AStage { PinnedStage mem; hipEvent_t done = nullptr; bool pending = false; uint64_t seq = 0; };
thread_local std::vector<std::unique_ptr<AStage>> t_astages;   // (no destructor work: j40hip_astage_release, or the process ends)
thread_local uint64_t t_astage_seq = 0;
thread_local FrontPlan t_front;
// J40HIP_ASYNC_TIMING=1: where this thread's host stage spends its time (ms; printed when the thread lets go of its staging)
thread_local double t_prof[8]; thread_local int64_t t_prof_frames = 0;
double prof_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// At most J40HIP_STAGE_BUFFERS (6) buffers per thread; when all of them are still being copied out the thread sleeps on the oldest.
// (Round 3 let a thread make up to 64: when a run floods the pipeline -- thousands of frames queued at once -- every worker kept
// pinning new 12 MB buffers, 40 and more each, for the first twenty seconds, and while memory is being pinned the copies of pixels
// back to the host crawl: 2.6-5.3 s per 256-frame batch instead of 0.6 s)
AStage *astage_acquire(size_t bytes) {
	static const size_t cap = [] { const char *e = getenv("J40HIP_STAGE_BUFFERS"); return (size_t) (e && atoi(e) > 0 ? atoi(e) : 6); }();
	for (auto &s : t_astages) {
		if (s->pending && hipEventQuery(s->done) != hipSuccess) { (void) hipGetLastError(); continue; }
		s->pending = false;
		s->seq = ++t_astage_seq;
		return s->mem.reserve(bytes, 0) ? s.get() : nullptr;
	}
	if (t_astages.size() >= cap) {   // wait for the one whose copy was enqueued first
		AStage *s = t_astages.front().get();
		for (auto &o : t_astages) if (o->seq < s->seq) s = o.get();
		(void) hipEventSynchronize(s->done); s->pending = false;
		s->seq = ++t_astage_seq;
		return s->mem.reserve(bytes, 0) ? s : nullptr;
	}
	std::unique_ptr<AStage> s(new AStage());
	if (hipEventCreateWithFlags(&s->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	// (full size at once: a buffer that grows is a buffer pinned again)
	if (!s->mem.reserve(std::max(bytes, (size_t) 16 << 20), 0)) { (void) hipEventDestroy(s->done); return nullptr; }
	s->seq = ++t_astage_seq;
	t_astages.push_back(std::move(s));
	return t_astages.back().get();
}

// events are recycled: creating and destroying one per frame cost the retiring thread a quarter of a millisecond a frame
std::mutex g_event_mutex;
std::vector<hipEvent_t> g_event_pool;
hipEvent_t event_acquire() {
	{ std::lock_guard<std::mutex> lock(g_event_mutex); if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; } }
	hipEvent_t e = nullptr;
	if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	return e;
}
void event_release(hipEvent_t e) { if (e) { std::lock_guard<std::mutex> lock(g_event_mutex); g_event_pool.push_back(e); } }

} // namespace
// the pixel-kernel streams every batch of a device shares (stream layout 1, j40hip_stream_layout)
static std::mutex g_shared_side_mutex;
static hipStream_t g_shared_side[16][4] = {};

// j40hip_shutdown's share of this file: the table cache, the event pool and the shared streams (nothing of the library may be running)
void j40hip_async_shutdown(void) {
	{ std::lock_guard<std::mutex> lock(g_static_mutex); g_static.clear(); }
	{
		std::lock_guard<std::mutex> lock(g_shared_side_mutex);
		for (int d = 0; d < 16; ++d) for (hipStream_t &st : g_shared_side[d]) if (st) { if (hipSetDevice(d) == hipSuccess) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); } st = nullptr; }
	}
	std::lock_guard<std::mutex> lock(g_event_mutex);
	for (hipEvent_t e : g_event_pool) (void) hipEventDestroy(e);
	g_event_pool.clear();
}
namespace {

struct Layout {
	size_t size = 0;
	size_t take(size_t bytes) { const size_t off = (size + 255) & ~(size_t) 255; size = off + bytes + 16; return off; }
};

} // namespace

struct j40hip_aframe {
	j40hip_frame host;            // the parsed front of the frame (and where its codestream lies)
	int device = 0;
	std::shared_ptr<StaticEntry> st;
	void *plan_block = nullptr, *work_block = nullptr; size_t plan_block_bytes = 0, work_block_bytes = 0;
	DevPlan plan; DevPlanBuild build;
	HfLaunchInfo hf;
	std::vector<DevLfTask> lf_tasks;   // empty: the LfGroup streams were decoded on the host
	DevLfLaneSet lf_set;               // ... else: the frame's sections for k_lf_lanes
	int32_t num_lf_groups = 0, max_lf_cells = 0, num_groups = 0, num_passes = 1;
	size_t cells = 0;
	bool sparse = true;
	hipEvent_t uploaded = nullptr;
	hipStream_t up_stream = nullptr; uint64_t up_seq = 0;   // the stream its copy went on, and its place among that stream's frames
	struct WorkLayout { size_t size = 0, stride = 0, coeffs = 0, blk = 0, nz = 0, status = 0, lz = 0, lfs = 0, recs = 0, gcnt = 0, gbs = 0, ccnt = 0, gb = 0, vbs = 0, llf[3] = {0, 0, 0}; uint32_t lz_window_size = 0; } wl;
};

// the frame's working set: acquired when its batch is launched, its per-block table cleared on the batch's stream
static uint32_t aframe_bind_work(j40hip_aframe *af, hipStream_t s) {
	if (af->work_block) return 0;
	bool dummy = false;
	af->work_block = cache_acquire(af->device, af->wl.size, &af->work_block_bytes, &dummy);
	if (!af->work_block) return ERR_MEM;
	uint8_t *wb = (uint8_t *) af->work_block;
	const j40hip_aframe::WorkLayout &wl = af->wl;
	DevPlan &plan = af->plan;
	if (af->sparse) { plan.events = (CoeffEvent *) (wb + wl.coeffs); plan.block_events = (uint32_t *) (wb + wl.blk); }
	else for (int c = 0; c < 3; ++c) plan.coeffs[c] = (float *) (wb + wl.coeffs) + (size_t) c * wl.stride;
	plan.coeff_stride = (uint32_t) wl.stride;
	plan.nonzeros = (int8_t *) (wb + wl.nz); plan.status = (uint32_t *) (wb + wl.status);
	plan.lz_window_size = wl.lz_window_size; plan.lz_window = wl.lz_window_size ? (int32_t *) (wb + wl.lz) : nullptr;
	plan.group_blocks = (const DevGroupBlock *) (wb + wl.gb); plan.group_block_start = (const uint32_t *) (wb + wl.gbs);
	for (int c = 0; c < 3; ++c) plan.llf[c] = (const float *) (wb + wl.llf[c]);
	DevPlanBuild &bd = af->build;
	bd.vb_recs = (DevVbRec *) (wb + wl.recs); bd.group_count = (uint32_t *) (wb + wl.gcnt); bd.group_block_start = (uint32_t *) (wb + wl.gbs); bd.class_count = (uint32_t *) (wb + wl.ccnt);
	bd.group_blocks = (DevGroupBlock *) (wb + wl.gb); bd.vb_sorted = (DevVarblock *) (wb + wl.vbs); bd.lf_scratch = (float *) (wb + wl.lfs);
	(void) s;
	return 0;
}

// makes `s` wait for the uploads of `n` frames: frames that went up on the same stream completed in order, so the youngest of each
// stream stands for all of them -- one wait per worker thread instead of one per frame (a wait costs the caller 50 us)
static uint32_t wait_for_uploads(j40hip_aframe *const *frames, int n, hipStream_t s) {
	std::vector<const j40hip_aframe *> last;
	for (int i = 0; i < n; ++i) {
		bool found = false;
		for (const j40hip_aframe *&l : last) if (l->up_stream == frames[i]->up_stream) { if (frames[i]->up_seq > l->up_seq) l = frames[i]; found = true; break; }
		if (!found) last.push_back(frames[i]);
	}
	for (const j40hip_aframe *l : last) if (hipStreamWaitEvent(s, l->uploaded, 0) != hipSuccess) return ERR_GPU;
	return 0;
}

// Sizes the device memory cache for a pipeline from its first batch: `work_copies` working sets and `front_copies` front blocks
// (codestream, plan, LfGroup planes) per frame of the batch are acquired and handed back, so that the cache holds them. Growing
// the cache costs hipMalloc calls of 10 ms and more each, which otherwise land wherever the pipeline first runs at full depth.
void j40hip_aframes_reserve(j40hip_aframe *const *frames, int n, int work_copies, int front_copies) {
	struct Held { void *p; size_t bytes; };
	std::vector<Held> held;
	// best effort: stops while a sixth of the device's memory is still free (the frames in flight and the caller's images need room,
	// and a failed hipMalloc inside cache_acquire would trim the very cache this is filling)
	size_t free_b = 0, total_b = 0;
	auto room = [&](size_t want) {
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return false; }
		return free_b > total_b / 6 + want;
	};
	bool go = true;
	for (int c = 0; go && c < std::max(work_copies, front_copies); ++c) for (int i = 0; go && i < n; ++i) {
		const j40hip_aframe *f = frames[i];
		bool dummy = false; size_t got = 0;
		if ((i & 15) == 0) go = room(16 * (f->wl.size + f->plan_block_bytes));
		if (!go) break;
		if (c < work_copies) { if (void *q = cache_acquire(f->device, f->wl.size, &got, &dummy)) held.push_back({q, got}); else go = false; }
		if (go && c < front_copies) { if (void *q = cache_acquire(f->device, f->plan_block_bytes, &got, &dummy)) held.push_back({q, got}); else go = false; }
	}
	if (n > 0) for (const Held &h : held) cache_release(frames[0]->device, h.p, h.bytes, false);
}

void j40hip_aframe_free(j40hip_aframe *f) {
	if (!f) return;
	(void) hipSetDevice(f->device);
	cache_release(f->device, f->plan_block, f->plan_block_bytes, false);
	cache_release(f->device, f->work_block, f->work_block_bytes, false);
	event_release(f->uploaded);
	delete f;
}
int j40hip_aframe_lf_on_device(const j40hip_aframe *f) { return f && !f->lf_tasks.empty(); }
int j40hip_aframe_uploaded(const j40hip_aframe *f) { if (!f || hipEventQuery(f->uploaded) == hipSuccess) return 1; (void) hipGetLastError(); return 0; }
int64_t j40hip_aframe_cells(const j40hip_aframe *f) { return f ? (int64_t) f->cells : 0; }
void j40hip_aframe_size(const j40hip_aframe *f, int64_t *width, int64_t *height) { *width = f->host.frame.fh.width; *height = f->host.frame.fh.height; }
uint32_t j40hip_aframe_after_frame_status(const j40hip_aframe *f) { return j40hip_frame_after_frame_status(&f->host); }

static j40hip_aframe *aframe_prepare_body(const void *buf, size_t size, int device, hipStream_t stream, int lf_on_device) {
	if (j40hip_device_count() <= device || device < 0 || hipSetDevice(device) != hipSuccess || !ensure_constant_tables(device)) return nullptr;
	std::unique_ptr<j40hip_aframe, void (*)(j40hip_aframe *)> af(new j40hip_aframe(), j40hip_aframe_free);
	af->device = device;
	j40hip_frame &h = af->host;
	Frame &fr = h.frame;
	std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
	const double tp0 = prof_now();
	extract_codestream((const uint8_t *) buf, size, &h.cs, &h.cs_size, &h.cs_storage, &h.container_stray_tail);
	h.bare_codestream = h.cs == (const uint8_t *) buf && h.cs_size == size;
	if (!parse_frame_front(h.cs, h.cs_size, &fr, &tasks, &extra_prec, &plain)) return nullptr;
	{   // the restoration filters asked for (J40HIP_RESTORATION) and signalled by the frame: the single-frame path runs them (runtime.hip: decode_restored)
		static const bool restoration = [] { const char *e = getenv("J40HIP_RESTORATION"); return e && (!strcmp(e, "j40") || atoi(e) > 0); }();
		if (restoration && (fr.fh.restoration.gab || fr.fh.restoration.epf_iters > 0)) return nullptr;
	}
	const double tp1 = prof_now();
	af->st = static_tables_for(fr, device);
	if (!af->st || af->st->any_dq_error) return nullptr;   // (a matrix that does not load: whether it matters depends on the varblocks -- the single-frame path sorts it out)
	StaticTables offs;   // (offsets only)
	memcpy(offs.order_off, af->st->order_off, sizeof offs.order_off); memcpy(offs.dq_off, af->st->dq_off, sizeof offs.dq_off);
	memcpy(offs.dq_size, af->st->dq_size, sizeof offs.dq_size); memcpy(offs.dq_scan_off, af->st->dq_scan_off, sizeof offs.dq_scan_off);
	FrontPlan &fp = t_front;
	if (build_front_plan(fr, offs, h.cs_size, extra_prec, lf_on_device != 0 && plain, &fp)) return nullptr;
	const bool dev_lf = fp.lf_device;
	const double tp2 = prof_now();
	const size_t ngg = fp.lf_groups.size(), cells = fp.cells, c64s = fp.c64s, cs_size = h.cs_size;
	const int32_t num_groups = fp.frame.num_groups;
	af->num_lf_groups = (int32_t) ngg; af->max_lf_cells = fp.max_lf_cells; af->num_groups = num_groups; af->num_passes = fp.frame.num_passes; af->cells = cells;
	af->sparse = fp.frame.sparse_coeffs != 0; af->hf = fp.hf;

	// ---- the plan block: what the host copies first, then what the device produces ----
	Layout L;
	const size_t o_cs = L.take(cs_size + LF_CODESTREAM_PAD), o_u8 = L.take(fp.pool_u8.size()), o_i32 = L.take(fp.pool_i32.size() * 4), o_u64 = L.take(fp.pool_u64.size() * 8);
	const size_t o_cl = L.take(fp.clusters.size() * sizeof(DevCluster)), o_spec = L.take(fp.coeff_specs.size() * sizeof(DevCodeSpec)), o_frame = L.take(sizeof(DevFrame));
	const size_t o_lfg = L.take(ngg * sizeof(DevLfGroup)), o_sec = L.take(fp.sections.size() * sizeof(DevSection)), o_evr = L.take(fp.ev_range.size() * 4), o_order = L.take(fp.lane_order.size() * 4);
	const size_t o_lso = L.take(ngg * 4), o_slots = L.take(ngg * sizeof(DevLfSlot));
	const size_t o_tree = L.take(fp.lf_tree.size() * sizeof(DevTreeNode)), o_alias = L.take(fp.lf_alias.size() * 8), o_lfmap = L.take(fp.lf_ctx_map.size()),
		o_lfcfg = L.take(fp.lf_cfg.size() * 4), o_tasks = L.take(ngg * sizeof(DevLfTask));   // (empty without the device decoder's tables)
	size_t o_raw[3], o_xfy, o_bfy, o_info, copy_bytes = L.size;
	for (int c = 0; c < 3; ++c) o_raw[c] = L.take(cells * 2 + 64);
	o_xfy = L.take(c64s * 2); o_bfy = L.take(c64s * 2); o_info = L.take(cells * 4 + 64);
	if (!dev_lf) copy_bytes = L.size;
	const size_t o_sharp = L.take(cells * 2 + 64);   // (device decoder only; reserved either way: blocks of one size recycle through the cache whoever decodes the streams)
	// (everything the plan build and the decode produce lives in the working set, which a frame gets when its batch is launched:
	// a frame waiting for its LfGroup streams holds 11 MB, not 210)

	AStage *sgp = astage_acquire(copy_bytes + 64);
	if (!sgp) return nullptr;
	AStage &sg = *sgp;
	uint8_t *stg = sg.mem.ptr;
	bool dummy = false;
	af->plan_block = cache_acquire(device, L.size, &af->plan_block_bytes, &dummy);
	if (!af->plan_block) return nullptr;
	uint8_t *pb = (uint8_t *) af->plan_block;
	const double tp3 = prof_now();

	memcpy(stg + o_cs, h.cs, cs_size); memset(stg + o_cs + cs_size, 0, LF_CODESTREAM_PAD);   // the lane decoders read past the position they stop at (plan.h)
	auto put = [&](size_t off, const void *src, size_t bytes) { if (bytes) memcpy(stg + off, src, bytes); };
	put(o_u8, fp.pool_u8.data(), fp.pool_u8.size()); put(o_i32, fp.pool_i32.data(), fp.pool_i32.size() * 4); put(o_u64, fp.pool_u64.data(), fp.pool_u64.size() * 8);
	put(o_cl, fp.clusters.data(), fp.clusters.size() * sizeof(DevCluster)); put(o_spec, fp.coeff_specs.data(), fp.coeff_specs.size() * sizeof(DevCodeSpec));
	put(o_frame, &fp.frame, sizeof(DevFrame)); put(o_lfg, fp.lf_groups.data(), ngg * sizeof(DevLfGroup)); put(o_sec, fp.sections.data(), fp.sections.size() * sizeof(DevSection));
	put(o_evr, fp.ev_range.data(), fp.ev_range.size() * 4); put(o_order, fp.lane_order.data(), fp.lane_order.size() * 4); put(o_lso, fp.lf_section_off.data(), ngg * 4);
	DevLfSlot *slots = (DevLfSlot *) (stg + o_slots);
	memset(slots, 0, ngg * sizeof(DevLfSlot));
	const double tp4 = prof_now();
	if (dev_lf) {
		put(o_tree, fp.lf_tree.data(), fp.lf_tree.size() * sizeof(DevTreeNode)); put(o_alias, fp.lf_alias.data(), fp.lf_alias.size() * 8);
		put(o_lfmap, fp.lf_ctx_map.data(), fp.lf_ctx_map.size()); put(o_lfcfg, fp.lf_cfg.data(), fp.lf_cfg.size() * 4);
		af->lf_tasks.resize(ngg);
		for (size_t g = 0; g < ngg; ++g) {
			const LfDeviceTask &t = tasks[g];
			const DevLfGroup &gg = fp.lf_groups[g];
			DevLfTask &d = af->lf_tasks[g];
			memset(&d, 0, sizeof d);
			d.codestream = pb + o_cs; d.tree = nullptr; d.alias = (const uint64_t *) (pb + o_alias); d.log_alpha_size = fp.lf_log_alpha;   // (k_lf_lanes reads the tables of the frame's DevLfLaneSet)
			d.byte_off = (uint32_t) t.byte_off; d.size = (uint32_t) t.size; d.bit_off = t.bit_off;
			d.w8 = t.w8; d.h8 = t.h8; d.w64 = t.w64; d.h64 = t.h64; d.sidx0 = t.sidx0; d.sidx2 = t.sidx2; d.nbvb_bits = t.nbvb_bits;
			// streamed order Y, X, B -> the frame-wide planes X, Y, B
			d.lf[0] = (int16_t *) (pb + o_raw[1]) + gg.cell_base; d.lf[1] = (int16_t *) (pb + o_raw[0]) + gg.cell_base; d.lf[2] = (int16_t *) (pb + o_raw[2]) + gg.cell_base;
			d.xfromy = (int16_t *) (pb + o_xfy) + gg.c64_base; d.bfromy = (int16_t *) (pb + o_bfy) + gg.c64_base;
			d.info = (int16_t *) (pb + o_info) + 2 * (size_t) gg.cell_base; d.info_capacity = (uint32_t) (2 * (size_t) gg.width8 * (size_t) gg.height8);
			d.sharp = (int16_t *) (pb + o_sharp) + gg.cell_base;
			d.result = (DevLfResult *) ((DevLfSlot *) (pb + o_slots) + g);   // (DevLfSlot's four words are DevLfResult's: the last two are the LfGroup decoder's scratch until k_lf_predict has zeroed them)
		}
		{   // the device sees the sections by decreasing size: k_lf_rows gives a lane two neighbours of the list, which should end together
			std::vector<DevLfTask> by_size = af->lf_tasks;
			std::stable_sort(by_size.begin(), by_size.end(), [](const DevLfTask &a, const DevLfTask &b) { return a.size > b.size; });
			put(o_tasks, by_size.data(), ngg * sizeof(DevLfTask));
		}
		DevLfLaneSet &ls = af->lf_set;
		ls.tasks = (const DevLfTask *) (pb + o_tasks); ls.ntasks = (int32_t) ngg;
		ls.tree = (const DevTreeNode *) (pb + o_tree); ls.num_nodes = (int32_t) fp.lf_tree.size();
		ls.ctx_map = pb + o_lfmap; ls.num_dist = (int32_t) fp.lf_ctx_map.size();
		ls.cluster_cfg = (const uint32_t *) (pb + o_lfcfg); ls.num_clusters = (int32_t) fp.lf_cfg.size();
		ls.alias = (const uint64_t *) (pb + o_alias); ls.log_alpha = fp.lf_log_alpha; ls.uses = fp.lf_uses; ls.lds_bytes = fp.lf_lds_bytes;
	} else {
		// the LfGroup streams on this thread (modular.cpp's fast paths); an error becomes the section's status and takes its place
		// among the frame's sections like the device decoder's would
		static const int XYB_FROM_STREAM[3] = {1, 0, 2};
		for (size_t g = 0; g < ngg; ++g) {
			const DevLfGroup &gg = fp.lf_groups[g];
			try {
				BitReader sr(h.cs + fr.toc.lf_groups[g].offset, fr.toc.lf_groups[g].size);
				LfRaw raw;
				read_lf_group_raw(sr, fr, fr.lf_groups[g], &raw);
				const size_t n = (size_t) gg.width8 * (size_t) gg.height8, n64 = (size_t) gg.width64 * (size_t) gg.height64;
				if (raw.lf[0].size() != n || raw.xfromy.size() != n64 || raw.bfromy.size() != n64) { slots[g].status = ERR_TODO; continue; }
				for (int c = 0; c < 3; ++c) memcpy(stg + o_raw[c] + 2 * (size_t) gg.cell_base, raw.lf[XYB_FROM_STREAM[c]].data(), n * 2);
				memcpy(stg + o_xfy + 2 * (size_t) gg.c64_base, raw.xfromy.data(), n64 * 2); memcpy(stg + o_bfy + 2 * (size_t) gg.c64_base, raw.bfromy.data(), n64 * 2);
				slots[g].nb_varblocks = raw.nb_varblocks;
				if ((size_t) raw.nb_varblocks > n) slots[g].status = ERR_VBLK;   // (more varblocks than cells: what the placement would find, j40.h:6689)
				else memcpy(stg + o_info + 4 * (size_t) gg.cell_base, raw.info.data(), raw.info.size() * 2);
			} catch (const DecodeError &e) { slots[g].status = e.code; }
		}
	}

	const double tp5 = prof_now();
	// ---- the working set: laid out now, acquired at the batch's launch (aframe_bind_work) ----
	{
		Layout W;
		j40hip_aframe::WorkLayout &wl = af->wl;
		wl.stride = (cells * 64 + 63) & ~(size_t) 63;
		const size_t coeff_bytes = af->sparse ? sizeof(CoeffEvent) * fp.ev_capacity : sizeof(float) * 3 * wl.stride;
		wl.coeffs = W.take(coeff_bytes); wl.blk = af->sparse ? W.take(16 * cells) : 0; wl.nz = W.take((size_t) num_groups * 32 * 32 * 3); wl.status = W.take(4 * fp.sections.size());
		wl.lz = fp.lz_window_size ? W.take(4 * (size_t) num_groups * fp.lz_window_size) : 0; wl.lfs = W.take(4 * 3 * cells);
		wl.recs = W.take(cells * sizeof(DevVbRec)); wl.gcnt = W.take((size_t) num_groups * 4); wl.gbs = W.take(((size_t) num_groups + 1) * 4); wl.ccnt = W.take(ngg * 28 * 4);
		wl.gb = W.take(cells * sizeof(DevGroupBlock)); wl.vbs = W.take(cells * sizeof(DevVarblock));
		for (int c = 0; c < 3; ++c) wl.llf[c] = W.take(cells * 4);
		wl.size = W.size; wl.lz_window_size = fp.lz_window_size;
	}

	DevPlan &plan = af->plan;
	memset(&plan, 0, sizeof plan);
	plan.frame = (const DevFrame *) (pb + o_frame); plan.codestream = pb + o_cs; plan.pool_u8 = pb + o_u8; plan.pool_u16 = af->st->d_u16; plan.pool_i32 = (const int32_t *) (pb + o_i32);
	plan.pool_u64 = (const uint64_t *) (pb + o_u64); plan.pool_f32 = af->st->d_f32; plan.clusters = (const DevCluster *) (pb + o_cl); plan.coeff_specs = (const DevCodeSpec *) (pb + o_spec);
	plan.lf_groups = (const DevLfGroup *) (pb + o_lfg); plan.sections = (const DevSection *) (pb + o_sec);
	plan.block_ctx_map_off = fp.block_ctx_map_off;
	for (int c = 0; c < 3; ++c) plan.lfraw[c] = (const int16_t *) (pb + o_raw[c]);
	plan.xfromy = (const int16_t *) (pb + o_xfy); plan.bfromy = (const int16_t *) (pb + o_bfy);
	plan.ev_range = (const uint32_t *) (pb + o_evr);
	static const bool raster_lanes = [] { const char *e = getenv("J40HIP_K1_RASTER"); return e && atoi(e) != 0; }();   // (the lanes in group order, as before: for comparisons)
	plan.lane_order = raster_lanes ? nullptr : (const uint32_t *) (pb + o_order);

	DevPlanBuild &bd = af->build;
	bd = fp.build;
	bd.pool_u8 = plan.pool_u8; bd.lf_groups = (DevLfGroup *) (pb + o_lfg); bd.lf_slots = (DevLfSlot *) (pb + o_slots);
	for (int c = 0; c < 3; ++c) bd.lfraw[c] = plan.lfraw[c];
	bd.xfromy = plan.xfromy; bd.bfromy = plan.bfromy; bd.vbinfo = (const int16_t *) (pb + o_info);
	bd.lf_section_off = (const uint32_t *) (pb + o_lso);
	bd.lf_smooth = fp.lf_smooth ? 1 : 0; bd.cells = (uint32_t) cells;
	for (int c = 0; c < 3; ++c) bd.inv_m_lf[c] = fp.inv_m_lf[c];
	// (the working set's pointers: aframe_bind_work; class_start and verdict belong to the batch: j40hip_abatch_launch)

	const double tp6 = prof_now();
	af->uploaded = event_acquire();
	if (!af->uploaded) return nullptr;
	static thread_local uint64_t t_seq = 0;
	af->up_stream = stream; af->up_seq = ++t_seq;
	// the upload: on an SDMA engine of this thread's own, kept apart from the copies back (hostcopy.hpp; the thread sleeps the quarter of a
	// millisecond it takes, and the events below then pass at once), or -- where that is not available -- hipMemcpyAsync on the thread's stream
	if (!j40hip_rt::hostcopy_h2d_sync(device, pb, stg, copy_bytes) && hipMemcpyAsync(pb, stg, copy_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
	bool ok = hipEventRecord(sg.done, stream) == hipSuccess;
	sg.pending = ok;
	ok = ok && hipEventRecord(af->uploaded, stream) == hipSuccess;
	if (!ok) { (void) hipStreamSynchronize(stream); (void) hipGetLastError(); return nullptr; }   // (nothing may be in flight on blocks that go back to the cache)
	{ const double tp7 = prof_now(); t_prof[0] += tp1 - tp0; t_prof[1] += tp2 - tp1; t_prof[2] += tp3 - tp2; t_prof[3] += tp4 - tp3; t_prof[4] += tp5 - tp4; t_prof[5] += tp6 - tp5; t_prof[6] += tp7 - tp6; ++t_prof_frames; }
	return af.release();
}

void j40hip_astage_release(void) {
	if (getenv("J40HIP_ASYNC_TIMING") && t_prof_frames) {
		const double n = (double) t_prof_frames;
		fprintf(stderr, "[j40hip host stage] %lld frames, ms per frame: front parse %.2f, tables + front plan %.2f, staging buffer + plan block %.2f, copy into staging %.2f, LfGroup streams / tasks %.2f, work block + pointers %.2f, enqueue %.2f; staging buffers %zu\n",
			(long long) t_prof_frames, t_prof[0] / n, t_prof[1] / n, t_prof[2] / n, t_prof[3] / n, t_prof[4] / n, t_prof[5] / n, t_prof[6] / n, t_astages.size());
		for (double &v : t_prof) v = 0; t_prof_frames = 0;
	}
	for (auto &s : t_astages) { if (s->pending) (void) hipEventSynchronize(s->done); s->mem.release(); if (s->done) (void) hipEventDestroy(s->done); }
	t_astages.clear();
	t_front = FrontPlan();
}

j40hip_aframe *j40hip_aframe_prepare(const void *buf, size_t size, int device, hipStream_t stream, int lf_on_device) {
	try { return aframe_prepare_body(buf, size, device, stream, lf_on_device); }
	catch (const DecodeError &) { return nullptr; }
	catch (const std::exception &) { return nullptr; }
}

// ---- batches ----

struct j40hip_abatch {
	int device = 0;
	PinnedStage host;                 // the batch's arrays, staged; one copy moves them
	void *dev = nullptr; size_t dev_cap = 0;
	uint32_t *verdict_host = nullptr; size_t verdict_cap = 0;   // pinned, [frames][4]
	int32_t nframes = 0;
	bool have_totals = false; int32_t last_totals[K2_NUM_BATCH_LAUNCHES];   // tiles per pixel-kernel launch of the batch before (k2_batch_grids)
	float *large_scratch = nullptr;
	bool shared_side = false;
	std::vector<hipStream_t> side; std::vector<hipEvent_t> side_done; hipEvent_t fork = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	hipEvent_t k1_ev[2] = {nullptr, nullptr};   // recorded by the device at k_hf_lanes' start and end (hipExtLaunchKernelGGL)
	bool k1_timed = false;
	int cus = 256;
	size_t last_o_k2 = 0;             // where the last launch's K2Frame array lies in `dev` (the stage dump reads class_start there)
};

// How a pipeline's streams are laid over the hardware queues. A process gets four queues per stream priority, streams of one
// priority share them in creation order, and kernels in one queue run one after the other. Layout 0 (the first form): every batch
// slot has its own stream and its own four pixel-kernel streams, all of normal priority -- ten and more streams on four queues, so a
// batch's plan build and entropy decode sat in a queue behind the pixel kernels of the batch before it and the batches ran one
// after the other. Layout 1: ONE set of four pixel-kernel streams per device at normal priority (a queue each, every batch's chains
// in order), the slots' streams (plan build, LfGroup tail, entropy decode) at high priority beside the copies.
// Layout 2 (for comparison): as 1, but every batch on one and the same stream, i.e. one batch after the other: k_hf_lanes then runs
// without another batch's pixel kernels beside it (48-52 ms instead of 52-75) and the steps take 8-15 % longer (195-200 ms
// against 169-186).
int j40hip_stream_layout(void) {
	static const int v = [] { const char *e = getenv("J40HIP_STREAM_LAYOUT"); return e ? atoi(e) : 1; }();
	return v;
}

j40hip_abatch *j40hip_abatch_create(int device) {
	if (hipSetDevice(device) != hipSuccess) return nullptr;
	j40hip_abatch *b = new j40hip_abatch();
	b->device = device;
	bool ok = true;
	for (auto &e : b->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
	for (auto &e : b->k1_ev) ok = ok && hipEventCreate(&e) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&b->fork, hipEventDisableTiming) == hipSuccess;
	int nside = 4;   // (kernels.hip: K2_LAUNCH_STREAM)
	if (const char *e = getenv("J40HIP_SIDE_STREAMS")) nside = std::max(0, std::min(4, atoi(e)));
	b->shared_side = j40hip_stream_layout() >= 1;
	for (int i = 0; i < nside && ok; ++i) {
		hipStream_t st = nullptr; hipEvent_t ev = nullptr;
		if (b->shared_side) {   // the device's four pixel-kernel streams, shared by every batch (see j40hip_stream_layout)
			std::lock_guard<std::mutex> lock(g_shared_side_mutex);
			if (device < 16 && !g_shared_side[device][i]) ok = hipStreamCreateWithFlags(&g_shared_side[device][i], hipStreamNonBlocking) == hipSuccess;
			st = device < 16 ? g_shared_side[device][i] : nullptr;
			ok = ok && st != nullptr;
		} else ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
		if (ok) { b->side.push_back(st); b->side_done.push_back(ev); }
	}
	ok = ok && hipMalloc((void **) &b->large_scratch, (size_t) K2_LARGE_WGS * 6 * 65536 * sizeof(float)) == hipSuccess;
	ok = ok && b->host.reserve((size_t) 8 << 20, 0);   // (pinned memory up front: growing it costs a third of a second a time)
	hipDeviceProp_t prop;
	if (ok && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) b->cus = prop.multiProcessorCount;
	if (!ok) { (void) hipGetLastError(); j40hip_abatch_free(b); return nullptr; }
	return b;
}

void j40hip_abatch_free(j40hip_abatch *b) {
	if (!b) return;
	(void) hipSetDevice(b->device);
	b->host.release();
	if (b->dev) (void) hipFree(b->dev);
	if (b->verdict_host) (void) hipHostFree(b->verdict_host);
	if (b->large_scratch) (void) hipFree(b->large_scratch);
	for (auto &e : b->ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : b->k1_ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : b->side_done) if (e) (void) hipEventDestroy(e);
	if (!b->shared_side) for (auto &s : b->side) if (s) (void) hipStreamDestroy(s);
	if (b->fork) (void) hipEventDestroy(b->fork);
	delete b;
}

static uint32_t abatch_launch_body(j40hip_abatch *b, j40hip_aframe *const *frames, int n, void *const *rgba_dev, const size_t *stride_bytes, hipStream_t s) {
	if (!b || n <= 0) return ERR_GPU;
	if (hipSetDevice(b->device) != hipSuccess) return ERR_GPU;
	if (b->have_totals && b->nframes > 0) memcpy(b->last_totals, b->verdict_host + 4 * (size_t) b->nframes, sizeof b->last_totals);   // (the launch before has been waited for)
	// geometry of the entropy launch (runtime.hip batch_assign): up to 64 sections of one frame per wavefront, 1 / 2 / 4 wavefronts
	// per workgroup sharing one copy of their frame's tables
	bool lanes_fast = true, tables_in_lds = true; uint32_t lanes_lds = 0, generic_lds = 0;
	int32_t total_waves = 0, max_frame_waves = 1, nlf = 0, max_lf_cells = 0; size_t cells_total = 0, max_frame_cells = 0;
	const double tq0 = prof_now();
	uint64_t cc0[10] = {0}, cc1[10] = {0};
	const bool timing = getenv("J40HIP_ASYNC_TIMING") != nullptr;
	if (timing) j40hip_cache_counters(cc0);
	for (int i = 0; i < n; ++i) {
		j40hip_aframe *f = frames[i];
		if (!f || f->device != b->device) return ERR_GPU;
		if (uint32_t e = aframe_bind_work(f, s)) return e;
		lanes_fast = lanes_fast && f->hf.lanes_fast; tables_in_lds = tables_in_lds && f->hf.tables_fit_lds; lanes_lds = std::max(lanes_lds, f->hf.lanes_lds_bytes);
		total_waves += (f->num_groups + 63) / 64; max_frame_waves = std::max(max_frame_waves, (f->num_groups + 63) / 64); nlf += f->num_lf_groups;
		max_lf_cells = std::max(max_lf_cells, f->max_lf_cells); cells_total += f->cells; max_frame_cells = std::max(max_frame_cells, f->cells);
	}
	if (const char *e = getenv("J40HIP_GENERIC_LANES")) if (atoi(e)) lanes_fast = false;
	// (eight wavefronts to a workgroup when the launch fills the machine: one copy of the tables per compute unit instead of two
	// leaves a third of its LDS to whatever else is running -- another batch's pixel kernels, the LfGroup lane decoder -- which
	// otherwise displaces one of the two workgroups and sends it into a second round: 46 -> 91 ms)
	int32_t waves_per_wg = lanes_fast ? (total_waves <= 2 * b->cus ? 1 : total_waves <= 4 * b->cus ? 2 : total_waves <= 6 * b->cus || lanes_lds + 8u * HF_LANE_COLS_BYTES > 150u * 1024u ? 4 : 8) : 1;
	// (a workgroup stays on one frame: with frames of a wavefront or two -- 1920 x 1080 is 40 sections -- larger workgroups would be padding)
	while (waves_per_wg > 1 && waves_per_wg / 2 >= max_frame_waves) waves_per_wg /= 2;
	if (const char *e = getenv("J40HIP_WAVES_PER_WG")) if (lanes_fast) waves_per_wg = std::max(1, std::min(lanes_lds + 8u * HF_LANE_COLS_BYTES > 150u * 1024u ? 4 : 8, atoi(e)));
	// More sections than the machine has lanes for (2048 wavefronts: eight per compute unit is what the tables' LDS leaves room
	// for): the QUEUED form -- every frame gets one workgroup of `queue_waves` wavefronts, whose lanes take the frame's sections by
	// decreasing size, first one each, then from a counter as they finish (k_hf_lanes) -- keeps every lane busy until its frame runs
	// out of sections; one section per lane would run in rounds, each as long as its longest section. Single-pass frames only.
	// J40HIP_K1_QUEUE_WAVES: 0 never, n > 0 always with n wavefronts per frame (tests), unset: decided here.
	int32_t queue_waves = 0;
	{
		static const int forced = [] { const char *e = getenv("J40HIP_K1_QUEUE_WAVES"); return e ? atoi(e) : -1; }();
		bool single_pass = true;
		for (int i = 0; i < n; ++i) single_pass = single_pass && frames[i]->num_passes == 1 && frames[i]->sparse;
		const int32_t capacity = 8 * b->cus;
		if (lanes_fast && single_pass && forced != 0 && (forced > 0 || total_waves > capacity)) {
			int32_t want = forced > 0 ? forced : std::max<int32_t>(1, (int32_t) ((int64_t) max_frame_waves * capacity / total_waves));
			want = std::min(want, std::min<int32_t>(8, max_frame_waves));
			while (want > 1 && lanes_lds + (uint32_t) want * HF_LANE_COLS_BYTES > 80u * 1024u && lanes_lds + 8u * HF_LANE_COLS_BYTES <= 150u * 1024u) --want;   // two workgroups per compute unit
			queue_waves = 1;
			while (queue_waves * 2 <= want) queue_waves *= 2;
			waves_per_wg = queue_waves;
		}
	}
	std::vector<HfLaneWork> work;
	std::vector<uint32_t> queue_start;
	for (int i = 0; i < n; ++i) {
		if (queue_waves) {
			for (int32_t k = 0; k < queue_waves; ++k) work.push_back({i, 64 * k, std::max(0, std::min(64, frames[i]->num_groups - 64 * k)), 1});
			queue_start.push_back((uint32_t) std::min(frames[i]->num_groups, 64 * queue_waves));
			continue;
		}
		const size_t first_entry = work.size();
		for (int32_t g = 0; g < frames[i]->num_groups; g += 64) work.push_back({i, g, std::min(64, frames[i]->num_groups - g), 0});
		while (work.size() % (size_t) waves_per_wg) work.push_back({i, 0, 0, 0});   // a workgroup stays on one frame
		// The lanes take the groups by decreasing section size (DevPlan::lane_order), so a workgroup's wavefronts come longest
		// first. A workgroup's wavefront w runs on SIMD w % 4: the second half of every workgroup is reversed, which puts the
		// longest beside the shortest, the second longest beside the second shortest ... -- four SIMDs with about equal work
		for (size_t a = first_entry; a + (size_t) waves_per_wg <= work.size(); a += (size_t) waves_per_wg) std::reverse(work.begin() + (long) (a + (size_t) waves_per_wg / 2), work.begin() + (long) (a + (size_t) waves_per_wg));
		// (... and, bit 1 of `pad`, the second half at a lower priority than the first when two wavefronts share a SIMD: the launch
		// ends with the wavefronts that have the largest sections, which then wait for nobody. J40HIP_K1_RANK_PRIO=0: all alike)
		static const bool rank_prio = [] { const char *e = getenv("J40HIP_K1_RANK_PRIO"); return e ? atoi(e) != 0 : false; }();
		if (rank_prio && waves_per_wg >= 8) for (size_t a = first_entry; a + (size_t) waves_per_wg <= work.size(); a += (size_t) waves_per_wg) for (size_t k = (size_t) waves_per_wg / 2; k < (size_t) waves_per_wg; ++k) work[a + k].pad |= 2;
	}
	for (int i = 0; i < n; ++i) {
		HfLaunchInfo info = frames[i]->hf; info.tables_fit_lds = tables_in_lds;
		generic_lds = std::max(generic_lds, hf_lanes_lds_bytes(info));
	}
	// ---- the batch's arrays: one staged blob, one copy ----
	Layout L;
	const size_t o_plans = L.take(sizeof(DevPlan) * (size_t) n), o_builds = L.take(sizeof(DevPlanBuild) * (size_t) n), o_k2 = L.take(sizeof(K2Frame) * (size_t) n);
	const size_t o_lfs = L.take(sizeof(DevBatchLf) * (size_t) nlf), o_work = L.take(sizeof(HfLaneWork) * work.size()), o_queue = L.take(4 * (size_t) n);
	const size_t copy_bytes = L.size;
	const size_t o_tiles = L.take(4 * (size_t) K2_NUM_BATCH_LAUNCHES * ((size_t) n + 1)), o_verdict = L.take(16 * (size_t) n + 64);   // (verdicts, then the tile totals)
	if (!b->host.reserve(copy_bytes + 64, 0)) return ERR_MEM;
	if (L.size > b->dev_cap) {
		if (b->dev) (void) hipFree(b->dev);
		b->dev = nullptr; b->dev_cap = 0;
		if (hipMalloc(&b->dev, L.size * 2) != hipSuccess) { (void) hipGetLastError(); return ERR_MEM; }
		b->dev_cap = L.size * 2;
	}
	if ((size_t) n * 4 + 16 > b->verdict_cap) {
		if (b->verdict_host) (void) hipHostFree(b->verdict_host);
		b->verdict_host = nullptr; b->verdict_cap = 0;
		if (hipHostMalloc((void **) &b->verdict_host, 16 * (size_t) n * 2 + 64, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); return ERR_MEM; }
		b->verdict_cap = (size_t) n * 4 * 2 + 16;
	}
	uint8_t *hb = b->host.ptr, *db = (uint8_t *) b->dev;
	b->last_o_k2 = o_k2;
	DevPlan *h_plans = (DevPlan *) (hb + o_plans); DevPlanBuild *h_builds = (DevPlanBuild *) (hb + o_builds); K2Frame *h_k2 = (K2Frame *) (hb + o_k2);
	DevBatchLf *h_lfs = (DevBatchLf *) (hb + o_lfs);
	const DevPlan *d_plans = (const DevPlan *) (db + o_plans); const DevPlanBuild *d_builds = (const DevPlanBuild *) (db + o_builds); K2Frame *d_k2 = (K2Frame *) (db + o_k2);
	const DevBatchLf *d_lfs = (const DevBatchLf *) (db + o_lfs); const HfLaneWork *d_work = (const HfLaneWork *) (db + o_work);
	size_t at_lf = 0;
	for (int i = 0; i < n; ++i) {
		const j40hip_aframe *f = frames[i];
		h_plans[i] = f->plan;
		h_builds[i] = f->build;
		h_builds[i].class_start = d_k2[i].class_start; h_builds[i].verdict = (uint32_t *) (db + o_verdict) + 4 * (size_t) i;
		K2Frame &k = h_k2[i];
		memset(&k, 0, sizeof k);
		k.plan = f->plan; k.sorted = f->build.vb_sorted; k.large_scratch = nullptr; k.rgba = (uint8_t *) rgba_dev[i]; k.stride = stride_bytes[i];
		for (int32_t g = 0; g < f->num_lf_groups; ++g) h_lfs[at_lf++] = DevBatchLf{i, g};
	}
	memcpy(hb + o_work, work.data(), sizeof(HfLaneWork) * work.size());
	if (queue_waves) memcpy(hb + o_queue, queue_start.data(), 4 * queue_start.size());
	if (uint32_t e = wait_for_uploads(frames, n, s)) return e;
	const double tq1 = prof_now();
	if (timing) j40hip_cache_counters(cc1);
	if (hipMemcpyAsync(db, hb, copy_bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ERR_GPU;
	const double tq2 = prof_now();
	// recycled memory: no entry of the per-block tables may point outside the event lists (a section that fails leaves entries unwritten)
	launch_clear_block_events(d_plans, d_builds, n, max_frame_cells, s);
	(void) hipEventRecord(b->ev[0], s);
	launch_plan_build(d_builds, d_lfs, n, nlf, max_lf_cells, s);
	launch_lf_tail_batch(d_plans, d_builds, d_lfs, n, nlf, max_lf_cells, max_frame_cells, s);
	for (int i = 0; i < n; ++i) if (!frames[i]->sparse && hipMemsetAsync(frames[i]->plan.coeffs[0], 0, sizeof(float) * 3 * (size_t) frames[i]->plan.coeff_stride, s) != hipSuccess) return ERR_GPU;
	(void) hipEventRecord(b->ev[1], s);
	const double tq3 = prof_now();
	b->k1_timed = lanes_fast;
	if (lanes_fast) launch_hf_lanes(d_plans, d_work, (int32_t) work.size(), waves_per_wg, lanes_lds, s, b->k1_ev[0], b->k1_ev[1], queue_waves ? (uint32_t *) (db + o_queue) : nullptr);
	else launch_hf_entropy_lanes(d_plans, d_work, (int32_t) work.size(), tables_in_lds, generic_lds, s);
	(void) hipEventRecord(b->ev[2], s);
	int32_t grids[K2_NUM_BATCH_LAUNCHES];
	static const int32_t k2_wgs = [] { const char *e = getenv("J40HIP_K2_WGS"); return e ? std::max(1, atoi(e)) : 32768; }();
	k2_batch_grids(b->have_totals ? b->last_totals : nullptr, cells_total, n, k2_wgs, grids);
	launch_vardct_batch(d_k2, n, (int32_t *) (db + o_tiles), (int32_t *) (db + o_verdict + 16 * (size_t) n), grids, b->large_scratch, s, b->side.data(), (int) b->side.size(), b->fork, b->side_done.data());
	(void) hipEventRecord(b->ev[3], s);
	launch_plan_verdict(d_builds, d_plans, n, s);
	if (hipMemcpyAsync(b->verdict_host, db + o_verdict, 16 * (size_t) n + 64, hipMemcpyDeviceToHost, s) != hipSuccess) return ERR_GPU;
	b->nframes = n; b->have_totals = true;
	if (timing) {
		const double tq4 = prof_now();
		fprintf(stderr, "[j40hip batch launch] bind %.2f (+arrays) %.2f, copy %.2f, plan+tail enqueue %.2f, K1+K2+verdict enqueue %.2f ms; the cache meanwhile (every thread's calls): %llu acquires, %llu hits, %llu + %llu hipMalloc %.2f ms, %llu hipFree %.2f ms, lock %.2f ms, search %.2f ms, %llu idle blocks\n",
			tq1 - tq0, 0.0, tq2 - tq1, tq3 - tq2, tq4 - tq3, (unsigned long long) (cc1[0] - cc0[0]), (unsigned long long) (cc1[1] - cc0[1]), (unsigned long long) (cc1[2] - cc0[2]), (unsigned long long) (cc1[3] - cc0[3]), (double) (cc1[7] - cc0[7]) * 1e-3,
			(unsigned long long) (cc1[4] - cc0[4]), (double) (cc1[8] - cc0[8]) * 1e-3, (double) (cc1[5] - cc0[5]) * 1e-3, (double) (cc1[6] - cc0[6]) * 1e-3, (unsigned long long) cc1[9]);
	}
	return hipGetLastError() == hipSuccess ? 0 : ERR_GPU;
}

uint32_t j40hip_abatch_launch(j40hip_abatch *b, j40hip_aframe *const *frames, int n, void *const *rgba_dev, const size_t *stride_bytes, hipStream_t stream) {
	try { return abatch_launch_body(b, frames, n, rgba_dev, stride_bytes, stream); } catch (const std::exception &) { return ERR_MEM; }
}

void j40hip_abatch_result(const j40hip_abatch *b, int i, uint32_t *code, int *redo) {
	if (!b || i < 0 || i >= b->nframes) { *code = ERR_GPU; *redo = 0; return; }
	*code = b->verdict_host[4 * (size_t) i]; *redo = b->verdict_host[4 * (size_t) i + 1] != 0;
}

uint32_t j40hip_abatch_elapsed(j40hip_abatch *b, float *ms3) {
	if (!b || hipEventSynchronize(b->ev[3]) != hipSuccess) return ERR_GPU;
	(void) hipEventElapsedTime(&ms3[0], b->ev[0], b->ev[1]); (void) hipEventElapsedTime(&ms3[1], b->ev[1], b->ev[2]); (void) hipEventElapsedTime(&ms3[2], b->ev[2], b->ev[3]);
	// [3]: k_hf_lanes alone, from its first wavefront's start to its last one's end (0: the batch took another entropy kernel)
	ms3[3] = 0;
	if (b->k1_timed && hipEventElapsedTime(&ms3[3], b->k1_ev[0], b->k1_ev[1]) != hipSuccess) { (void) hipGetLastError(); ms3[3] = 0; }
	return 0;
}

// ---- the LfGroup streams of several prepared frames in one launch (k_lf_groups), ahead of the frames' batch ----
// A section takes about 0.2 s to decode whatever else runs (one wavefront, a quarter of a million samples one after the other), so
// the sections of many frames go into one launch, on a stream of its own, and a frame joins a batch when its launch has completed.

struct j40hip_alf {
	int device = 0;
	PinnedStage host; void *dev = nullptr; size_t dev_cap = 0;
	hipEvent_t done = nullptr;
	hipEvent_t kev[2] = {nullptr, nullptr};   // recorded by the device at the lane decoder's start and end (hipExtLaunchKernelGGL)
	int frames = 0, sections = 0, waves = 0;
	int lanes_frames = 0;   // frames of the last launch that k_lf_lanes took (their tables exceed k_lf_rows' LDS, or J40HIP_LF_KERNEL=lanes)
};

j40hip_alf *j40hip_alf_create(int device) {
	if (hipSetDevice(device) != hipSuccess) return nullptr;
	j40hip_alf *a = new j40hip_alf();
	a->device = device;
	if (hipEventCreateWithFlags(&a->done, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); delete a; return nullptr; }
	for (hipEvent_t &e : a->kev) if (hipEventCreate(&e) != hipSuccess) { (void) hipGetLastError(); e = nullptr; }
	// (room for a few thousand frames up front: growing pinned memory costs a third of a second a time)
	if (!a->host.reserve((size_t) 4 << 20, 0) || hipMalloc(&a->dev, (size_t) 4 << 20) != hipSuccess) { (void) hipGetLastError(); j40hip_alf_free(a); return nullptr; }
	a->dev_cap = (size_t) 4 << 20;
	return a;
}
void j40hip_alf_free(j40hip_alf *a) {
	if (!a) return;
	(void) hipSetDevice(a->device);
	a->host.release();
	if (a->dev) (void) hipFree(a->dev);
	if (a->done) (void) hipEventDestroy(a->done);
	for (hipEvent_t e : a->kev) if (e) (void) hipEventDestroy(e);
	delete a;
}
uint32_t j40hip_alf_launch(j40hip_alf *a, j40hip_aframe *const *frames, int n, hipStream_t s) {
	if (!a || n <= 0 || hipSetDevice(a->device) != hipSuccess) return ERR_GPU;
	const double tq0 = prof_now();
	std::vector<DevLfLaneSet> sets((size_t) n);
	for (int i = 0; i < n; ++i) sets[(size_t) i] = frames[i]->lf_set;
	std::vector<DevLfWave> waves;
	// k_lf_rows for every frame whose tables fit a wavefront's LDS, k_lf_lanes for the others (alone: one outlier frame no longer
	// sends the whole flight to the slower kernel); the wavefronts of both in one array, the rows' first
	std::vector<int32_t> oversized;
	uint32_t lds = lf_rows_enabled() ? pack_lf_row_waves(sets.data(), n, &waves, &oversized) : 0u;
	const bool rows = lf_rows_enabled();
	const size_t row_waves = rows ? waves.size() : 0;
	uint32_t lds_lanes = 0;
	if (!rows) lds = pack_lf_waves(sets.data(), n, &waves);
	else if (!oversized.empty()) lds_lanes = pack_lf_waves(sets.data(), n, &waves, &oversized);
	a->lanes_frames = rows ? (int) oversized.size() : n;
	const size_t o_waves = (sizeof(DevLfLaneSet) * (size_t) n + 255) & ~(size_t) 255, bytes = o_waves + sizeof(DevLfWave) * waves.size() + 64;
	if (!a->host.reserve(bytes, 0)) return ERR_MEM;
	if (bytes > a->dev_cap) {
		if (a->dev) (void) hipFree(a->dev);
		a->dev = nullptr; a->dev_cap = 0;
		if (hipMalloc(&a->dev, bytes * 2) != hipSuccess) { (void) hipGetLastError(); return ERR_MEM; }
		a->dev_cap = bytes * 2;
	}
	memcpy(a->host.ptr, sets.data(), sizeof(DevLfLaneSet) * (size_t) n);
	memcpy(a->host.ptr + o_waves, waves.data(), sizeof(DevLfWave) * waves.size());
	const double tq1 = prof_now();
	if (uint32_t e = wait_for_uploads(frames, n, s)) return e;
	if (hipMemcpyAsync(a->dev, a->host.ptr, bytes - 64, hipMemcpyHostToDevice, s) != hipSuccess) return ERR_GPU;
	const double tq2 = prof_now();
	a->frames = n; a->waves = (int) waves.size(); a->sections = 0;
	for (const DevLfLaneSet &ls : sets) a->sections += ls.ntasks;
	hipEvent_t k0 = a->kev[0] && a->kev[1] ? a->kev[0] : nullptr, k1 = k0 ? a->kev[1] : nullptr;
	const DevLfWave *dw = (const DevLfWave *) ((const uint8_t *) a->dev + o_waves);
	if (rows) {
		const bool both = row_waves > 0 && waves.size() > row_waves;
		launch_lf_rows((const DevLfLaneSet *) a->dev, dw, (int32_t) row_waves, lds + 64, s, k0, both ? nullptr : k1);
		launch_lf_lanes((const DevLfLaneSet *) a->dev, dw + row_waves, (int32_t) (waves.size() - row_waves), lds_lanes + 64, s, row_waves ? nullptr : k0, k1);
	} else launch_lf_lanes((const DevLfLaneSet *) a->dev, dw, (int32_t) waves.size(), lds + 64, s, k0, k1);
	const double tq3 = prof_now();
	if (hipEventRecord(a->done, s) != hipSuccess || hipGetLastError() != hipSuccess) return ERR_GPU;
	if (getenv("J40HIP_ASYNC_TIMING")) fprintf(stderr, "[j40hip lf launch] %d frames, %zu waves: pack %.2f, copy %.2f, launch %.2f, record %.2f ms\n", n, waves.size(), tq1 - tq0, tq2 - tq1, tq3 - tq2, prof_now() - tq3);
	// (the caller hands the frames to a batch only once j40hip_alf_done says this launch has completed)
	return 0;
}
// the finished launch's own duration (device-recorded start / end), frames, sections and wavefronts; 0 when it could be read
int j40hip_alf_elapsed(j40hip_alf *a, float *ms, int *frames, int *sections, int *waves) {
	if (!a || !a->kev[0] || !a->kev[1]) return -1;
	float t = 0;
	if (hipEventElapsedTime(&t, a->kev[0], a->kev[1]) != hipSuccess) { (void) hipGetLastError(); return -1; }
	if (ms) *ms = t;
	if (frames) *frames = a->frames;
	if (sections) *sections = a->sections;
	if (waves) *waves = a->waves;
	return 0;
}
int j40hip_alf_done(j40hip_alf *a) { if (!a || hipEventQuery(a->done) == hipSuccess) return 1; (void) hipGetLastError(); return 0; }

// ---- stage dump (include/j40hip.h, j40hip_stage_dump_*): ONE image through the pipeline's device stages -- LfGroup streams
// (k_lf_lanes, or the host decoder), plan build (k_plan_place / _scan / _emit), LfGroup tail, entropy decode, pixels, verdict --
// and everything they produced copied back, so that the parity tests can hold each stage's product against the reference's
// j40__lf_group_st (j40.h:6360-6390) ON THE DEVICE THAT COMPUTED IT (tests/test_device_stages.py). Test access only: nothing here
// is on the product's path.
struct j40hip_stage_dump {
	DevPlanBuild build;                 // (host copy: thresholds, geometry; its pointers are device pointers and not used)
	std::vector<DevLfGroup> lf_groups; std::vector<DevLfSlot> slots;
	std::vector<int16_t> lfraw[3], xfromy, bfromy, vbinfo, sharp;
	std::vector<DevVbRec> recs; std::vector<uint32_t> group_block_start; std::vector<DevGroupBlock> group_blocks;
	std::vector<DevVarblock> sorted; int32_t class_start[28];
	std::vector<float> llf[3];
	std::vector<uint8_t> rgba;
	uint32_t verdict[4]; int32_t width = 0, height = 0, num_groups = 0; bool lf_device = false; size_t cells = 0, c64s = 0;
};

extern "C" j40hip_stage_dump *j40hip_stage_dump_create(const void *buf, size_t size, int device, int lf_on_device, uint32_t *err) {
	uint32_t dummy; if (!err) err = &dummy;
	*err = ERR_GPU;
	if (j40hip_device_count() <= device || device < 0 || hipSetDevice(device) != hipSuccess) return nullptr;
	hipStream_t s = nullptr;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
	j40hip_aframe *af = j40hip_aframe_prepare(buf, size, device, s, lf_on_device);
	j40hip_abatch *b = nullptr; j40hip_alf *alf = nullptr; void *img = nullptr;
	std::unique_ptr<j40hip_stage_dump> d(new j40hip_stage_dump());
	auto fail = [&](uint32_t code) {
		(void) hipStreamSynchronize(s);
		if (af) j40hip_aframe_free(af);
		if (b) j40hip_abatch_free(b);
		if (alf) j40hip_alf_free(alf);
		if (img) (void) hipFree(img);
		j40hip_astage_release();
		(void) hipStreamDestroy(s);
		*err = code;
		return (j40hip_stage_dump *) nullptr;
	};
	if (!af) return fail(ERR_TODO);   // (not a frame the batched path takes -- or not a frame at all: the single-frame path would tell)
	try {
		d->lf_device = j40hip_aframe_lf_on_device(af) != 0;
		if (d->lf_device) {
			alf = j40hip_alf_create(device);
			if (!alf) return fail(ERR_GPU);
			if (uint32_t e = j40hip_alf_launch(alf, &af, 1, s)) return fail(e);
		}
		int64_t w = 0, h = 0;
		j40hip_aframe_size(af, &w, &h);
		d->width = (int32_t) w; d->height = (int32_t) h; d->num_groups = af->num_groups; d->cells = af->cells;
		const size_t stride = (size_t) w * 4;
		if (hipMalloc(&img, stride * (size_t) h) != hipSuccess) return fail(ERR_MEM);
		b = j40hip_abatch_create(device);
		if (!b) return fail(ERR_GPU);
		if (uint32_t e = j40hip_abatch_launch(b, &af, 1, &img, &stride, s)) return fail(e);
		if (hipStreamSynchronize(s) != hipSuccess) return fail(ERR_GPU);
		const DevPlanBuild &bd = af->build;
		d->build = bd;
		const size_t ngg = (size_t) af->num_lf_groups, cells = af->cells;
		bool ok = true;
		auto pull = [&](void *dst, const void *src, size_t bytes) { if (bytes && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) ok = false; };
		d->lf_groups.resize(ngg); d->slots.resize(ngg);
		pull(d->lf_groups.data(), bd.lf_groups, ngg * sizeof(DevLfGroup)); pull(d->slots.data(), bd.lf_slots, ngg * sizeof(DevLfSlot));
		size_t c64s = 0;
		for (const DevLfGroup &g : d->lf_groups) c64s = std::max(c64s, (size_t) g.c64_base + (size_t) g.width64 * (size_t) g.height64);
		d->c64s = c64s;
		for (int c = 0; c < 3; ++c) { d->lfraw[c].resize(cells); pull(d->lfraw[c].data(), bd.lfraw[c], cells * 2); d->llf[c].resize(cells); pull(d->llf[c].data(), af->plan.llf[c], cells * 4); }
		d->xfromy.resize(c64s); d->bfromy.resize(c64s); pull(d->xfromy.data(), bd.xfromy, c64s * 2); pull(d->bfromy.data(), bd.bfromy, c64s * 2);
		d->vbinfo.resize(2 * cells); pull(d->vbinfo.data(), bd.vbinfo, 4 * cells);
		if (d->lf_device && !af->lf_tasks.empty()) {   // the sharpness map: only the device decoder keeps it (frame-wide cell array)
			d->sharp.resize(cells);
			pull(d->sharp.data(), (const uint8_t *) af->lf_tasks[0].sharp - 2 * (size_t) d->lf_groups[0].cell_base, cells * 2);
		}
		d->recs.resize(cells); pull(d->recs.data(), bd.vb_recs, cells * sizeof(DevVbRec));
		d->group_block_start.resize((size_t) af->num_groups + 1); pull(d->group_block_start.data(), bd.group_block_start, d->group_block_start.size() * 4);
		d->group_blocks.resize(cells); pull(d->group_blocks.data(), bd.group_blocks, cells * sizeof(DevGroupBlock));
		d->sorted.resize(cells); pull(d->sorted.data(), bd.vb_sorted, cells * sizeof(DevVarblock));
		pull(d->class_start, ((const K2Frame *) ((const uint8_t *) b->dev + b->last_o_k2))->class_start, sizeof d->class_start);
		memcpy(d->verdict, b->verdict_host, sizeof d->verdict);
		d->rgba.resize(stride * (size_t) h); pull(d->rgba.data(), img, d->rgba.size());
		if (!ok) return fail(ERR_GPU);
	} catch (const std::exception &) { return fail(ERR_MEM); }
	(void) fail(0);   // (releases everything but the dump)
	return d.release();
}

extern "C" void j40hip_stage_dump_free(j40hip_stage_dump *d) { delete d; }

extern "C" void j40hip_stage_dump_info(const j40hip_stage_dump *d, uint32_t *out8) {
	out8[0] = d->verdict[0]; out8[1] = d->verdict[1] | (d->lf_device ? 4u : 0u); out8[2] = (uint32_t) d->lf_groups.size(); out8[3] = (uint32_t) d->num_groups;
	out8[4] = (uint32_t) d->width; out8[5] = (uint32_t) d->height; out8[6] = d->verdict[2]; out8[7] = d->verdict[3];
}

extern "C" int j40hip_stage_dump_lf_group_info(const j40hip_stage_dump *d, int64_t gg, int32_t *out10) {
	if (!d || gg < 0 || (size_t) gg >= d->lf_groups.size()) return -1;
	const DevLfGroup &g = d->lf_groups[(size_t) gg];
	const int32_t v[10] = {g.left, g.top, g.width, g.height, g.width8, g.height8, g.width64, g.height64, d->slots[(size_t) gg].placed, (int32_t) d->slots[(size_t) gg].status};
	memcpy(out10, v, sizeof v);
	return 0;
}

extern "C" int j40hip_stage_dump_plane(const j40hip_stage_dump *d, int64_t gg, int which, void *out) {
	if (!d || gg < 0 || (size_t) gg >= d->lf_groups.size()) return -1;
	const DevLfGroup &g = d->lf_groups[(size_t) gg];
	const size_t n = (size_t) g.width8 * (size_t) g.height8, n64 = (size_t) g.width64 * (size_t) g.height64;
	if (which == 0) {   // the reference's block map (j40.h:6360, 6693-6697): (DctSelect + 2) << 20 | varblock at the top-left cell, 1 << 20 | varblock elsewhere
		int32_t *o = (int32_t *) out;
		for (size_t i = 0; i < n; ++i) o[i] = 0;
		const int32_t placed = d->slots[(size_t) gg].placed;
		for (int32_t v = 0; v < placed; ++v) {
			const DevVbRec &r = d->recs[(size_t) g.vb_base + (size_t) v];
			const int32_t vw8 = 1 << (DCT_SELECT[r.dctsel].log_columns - 3), vh8 = 1 << (DCT_SELECT[r.dctsel].log_rows - 3);
			for (int32_t y = 0; y < vh8; ++y) for (int32_t x = 0; x < vw8; ++x) {
				const size_t at = (size_t) (r.y8 + y) * (size_t) g.width8 + (size_t) (r.x8 + x);
				if (at < n) o[at] = ((x == 0 && y == 0 ? r.dctsel + 2 : 1) << 20) | v;
			}
		}
		return 0;
	}
	if (which == 1) {   // the LF index of every cell from the decoded LF integers (j40.h:6566-6570; plan_dev.h computes it at the varblocks' top-left cells)
		const DevPlanBuild &pb = d->build;
		uint8_t *o = (uint8_t *) out;
		for (size_t i = 0; i < n; ++i) {
			const size_t cell = (size_t) g.cell_base + i;
			const int32_t vx = d->lfraw[0][cell], vy = d->lfraw[1][cell], vb = d->lfraw[2][cell];
			uint8_t lfidx = 0;
			for (int32_t t = 0; t < pb.nb_lf_thr[0]; ++t) lfidx = (uint8_t) (lfidx + (vx > pb.lf_thr[0][t]));
			lfidx = (uint8_t) (lfidx * (pb.nb_lf_thr[0] + 1));
			for (int32_t t = 0; t < pb.nb_lf_thr[2]; ++t) lfidx = (uint8_t) (lfidx + (vb > pb.lf_thr[2][t]));
			lfidx = (uint8_t) (lfidx * (pb.nb_lf_thr[2] + 1));
			for (int32_t t = 0; t < pb.nb_lf_thr[1]; ++t) lfidx = (uint8_t) (lfidx + (vy > pb.lf_thr[1][t]));
			o[i] = lfidx;
		}
		return 0;
	}
	if (which == 2 || which == 3) { memcpy(out, (which == 2 ? d->xfromy : d->bfromy).data() + g.c64_base, n64 * 2); return 0; }
	if (which == 4) { if (d->sharp.empty()) return -1; memcpy(out, d->sharp.data() + g.cell_base, n * 2); return 0; }
	if (which >= 5 && which <= 7) { memcpy(out, d->lfraw[which - 5].data() + g.cell_base, n * 2); return 0; }
	return -1;
}

extern "C" int j40hip_stage_dump_varblocks(const j40hip_stage_dump *d, int64_t gg, int32_t *coeffoff_qfidx, float *hfmul_inv, int32_t *x8_y8_dctsel) {
	if (!d || gg < 0 || (size_t) gg >= d->lf_groups.size()) return -1;
	const DevLfGroup &g = d->lf_groups[(size_t) gg];
	const int32_t placed = d->slots[(size_t) gg].placed;
	for (int32_t v = 0; v < placed; ++v) {
		const DevVbRec &r = d->recs[(size_t) g.vb_base + (size_t) v];
		coeffoff_qfidx[v] = (int32_t) r.coeffoff_qfidx;
		hfmul_inv[v] = 1.0f / ((float) r.hfmul_m1 + 1.0f);   // j40.h:6699
		if (x8_y8_dctsel) { x8_y8_dctsel[3 * v] = r.x8; x8_y8_dctsel[3 * v + 1] = r.y8; x8_y8_dctsel[3 * v + 2] = r.dctsel; }
	}
	return placed;
}

extern "C" int j40hip_stage_dump_llf(const j40hip_stage_dump *d, int64_t gg, int c, float *out) {
	if (!d || gg < 0 || (size_t) gg >= d->lf_groups.size() || c < 0 || c > 2) return -1;
	const DevLfGroup &g = d->lf_groups[(size_t) gg];
	memcpy(out, d->llf[c].data() + g.cell_base, (size_t) g.width8 * (size_t) g.height8 * 4);
	return 0;
}

extern "C" int64_t j40hip_stage_dump_group_blocks(const j40hip_stage_dump *d, int64_t group, uint32_t *out3, int64_t capacity) {
	if (!d || group < 0 || group >= d->num_groups) return -1;
	const uint32_t a = d->group_block_start[(size_t) group], b = d->group_block_start[(size_t) group + 1];
	if (b < a || b > d->group_blocks.size()) return -1;
	for (uint32_t k = a; k < b && (int64_t) (k - a) < capacity; ++k) { const DevGroupBlock &gb = d->group_blocks[k]; uint32_t *o = out3 + 3 * (size_t) (k - a); o[0] = gb.coeffoff_qfidx; o[1] = gb.pos_dct; o[2] = gb.bctx3; }
	return (int64_t) (b - a);
}

extern "C" int64_t j40hip_stage_dump_sorted_varblocks(const j40hip_stage_dump *d, int32_t *out8, float *out3, int32_t *class_start28, int64_t capacity) {
	if (!d) return -1;
	memcpy(class_start28, d->class_start, sizeof d->class_start);
	const int64_t n = d->class_start[27];
	if (n < 0 || (size_t) n > d->sorted.size()) return -1;
	for (int64_t k = 0; k < n && k < capacity; ++k) {
		const DevVarblock &v = d->sorted[(size_t) k];
		int32_t *o = out8 + 8 * k; float *f = out3 + 3 * k;
		o[0] = v.px; o[1] = v.py; o[2] = v.effw; o[3] = v.effh; o[4] = v.dctsel; o[5] = v.blk; o[6] = v.llf_base; o[7] = v.coeff_base;
		f[0] = v.mult1; f[1] = v.kx_hf; f[2] = v.kb_hf;
	}
	return n;
}

extern "C" int j40hip_stage_dump_rgba(const j40hip_stage_dump *d, uint8_t *out) { if (!d) return -1; memcpy(out, d->rgba.data(), d->rgba.size()); return 0; }

