// j40_amd/csrc/device/pipeline.hip -- whole-frame throughput pipeline over the thin C-ABI (include/j40hip.h, j40hip_pipeline_*):
// codestream bytes in host memory -> RGBA u8x4 in device or host memory, with every stage of many frames in flight at once.
//
//   host worker threads   VarDCT frames with several sections (async.hip): container, headers, TOC, LfGlobal, HfGlobal (the
//                         reference's j40.h:8175-8192 work) and the first bits of every LfGroup section; the front of the plan and the
//                         codestream go to the device with one asynchronous copy, and the thread moves on -- it never waits for the
//                         device. Whether the thread also decodes the frame's LfGroup streams (j40.h:6722-6790) or leaves them to
//                         the device's lane decoder (k_lf_lanes) is the pipeline's mode or, in mode 0, decided frame by frame
//                         (j40hip_pipeline: lf_mode).
//                         Every other frame (Modular, a single section, extra channels, ...) the thread decodes on its own through
//                         the single-frame entry points (j40hip_frame_parse / upload / decode), on its own stream.
//   one GPU thread        collects prepared frames into batches; per batch ONE enqueue of LfGroup streams -> plan build -> LfGroup tail
//                         -> entropy decode -> pixels -> verdict on the batch slot's stream (j40hip_abatch_launch), then -- host
//                         output -- the copies back on the pipeline's copy stream behind the batch's kernels, in groups of frames
//                         with an event each; the thread polls, it never sleeps on the device while something could be enqueued
//   completion            one 16-byte verdict per frame comes back with the batch's kernels (the frames' device memory goes back to the
//                         cache then); a frame is complete when its group's copies are through; frames the device wants decoded again
//                         (an LfGroup header the device decoder does not take, an event region that overflowed) take the single-frame path
//
// The reference decodes one image on one core (j40.h:8034: its only threading hook is commented out); this is the serving
// shape of the same work: frames are independent, so the host part scales over cores and the device part over a batch.
// Nothing here touches the oracle; without a HIP device creation fails with "!gpu".
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "../../../include/j40hip.h"
#include "../capi.hpp"
#include "async.hpp"
#include "runtime_shared.hpp"
#include "hostcopy.hpp"

static int cpu_quota();   // (CPUs' worth of time the container may use; defined with the serving pipeline below)

namespace {

constexpr uint32_t E_GPU = ('!' << 24) | ('g' << 16) | ('p' << 8) | 'u';
constexpr uint32_t E_MEM = ('!' << 24) | ('m' << 16) | ('e' << 8) | 'm';
constexpr uint32_t E_EVOF = ('e' << 24) | ('v' << 16) | ('o' << 8) | 'f';
constexpr uint32_t E_RNGE = ('r' << 24) | ('n' << 16) | ('g' << 8) | 'e';

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Job {
	int64_t ticket = 0;
	const void *buf = nullptr; size_t size = 0;
	void *rgba = nullptr; size_t stride = 0; bool device_output = false;
	j40hip_aframe *af = nullptr;
	int64_t width = 0, height = 0, cells = 0;
	double ready_ms = 0;          // when it became ready for a batch
	bool lf_failed = false;       // its LfGroup streams could not be launched on the device
	void *dev_rgba = nullptr;     // host output: the device image the copy back reads
	uint64_t copy_ticket = 0;     // ... its copy back on the SDMA engine (hostcopy.hpp), once issued
	uint32_t status = 0;
	bool redo = false;            // its batch wants it decoded again on the single-frame path
	bool counted = true;          // counts towards the pipeline's resident frames (until its device memory has gone back to the cache)
	// j40hip_pipeline_run: the caller sleeps on `waiter` until the image is done, and its pixel memory is asked for once the
	// image's size is known (`alloc`, called on a pipeline thread)
	j40hip_output_alloc alloc = nullptr; void *alloc_ctx = nullptr;
	struct Waiter *waiter = nullptr;
};

struct Waiter { std::mutex m; std::condition_variable cv; bool done = false; uint32_t status = 0; };

struct LfFlight {                 // one launch of LfGroup streams in flight (frames whose streams the device decodes)
	hipStream_t stream = nullptr;
	j40hip_alf *alf = nullptr;
	std::vector<Job *> jobs;
	bool busy = false;
};

// One batch in flight. Its frames are handed back in GROUPS as the device gets through them: `kdone` follows the batch's kernels
// and the copy of its verdicts on the batch's stream; frames whose pixels go to host memory are copied on the pipeline's copy
// stream behind it, and every few frames an event is recorded there -- a frame is complete (its caller woken) when its group's
// event has passed, not when the whole batch's copies are through (256 8K frames are 0.6 s of PCIe traffic).
struct Slot {
	hipStream_t stream = nullptr;
	hipEvent_t kdone = nullptr;
	j40hip_abatch *batch = nullptr;
	std::vector<Job *> jobs;
	std::vector<hipEvent_t> group_ev;   // made on demand, kept
	std::vector<int> group_end;         // jobs [group_end[g - 1], group_end[g]) complete with group_ev[g]
	int next_group = 0;
	bool copies_deferred = false;       // host output on the SDMA engine: the copies are issued when the kernels are seen to be through (progress)
	std::vector<uint8_t> group_has_ev;  // ... groups with a copy that went through hipMemcpyAsync after all (an unpinned destination): group_ev[g] counts
	double t_launch = 0, t_harvest = 0;  // (J40HIP_ASYNC_TIMING)
	bool harvested = false;             // the kernels are through: verdicts read, the frames' device memory handed back (the copies may still run)
	bool failed = false;                // the device reported an error for this batch: everything still pending fails with "!gpu"
	uint32_t launch_err = 0;      // the batch could not be enqueued: every member fails with this
	bool busy = false;
};

} // namespace

struct j40hip_pipeline {
	int device = 0, batch_frames = 32, max_in_flight = 2;
	double max_wait_ms = 0;             // > 0: a prepared frame waits at most this long for a full batch while a slot is free (serving; j40hip_pipeline_set_max_wait_ms)
	int lf_mode = 0;                    // LfGroup streams: 0 decided per frame (see above), 1 always the device, 2 always the host threads
	// mode 0: the device decodes a section in 0.4 s and thousands of them at once (k_lf_lanes: a lane per section, a few dozen
	// wavefronts per launch that the other kernels hardly notice); a host thread decodes a frame's twelve in 12 ms, one frame at a
	// time. So the device takes the stream of frames, and the host threads decode a frame's sections themselves when
	//   * the device's stage is full (lf_cap frames), or
	//   * frames trickle in (less than a quarter batch waiting, nothing in the stage to join): the frame would wait 0.4 s alone.
	// (Measured and dropped: half of the threads decoding a frame's streams themselves whenever the device's stage is primed -- 190 ms
	// a step against 182 with the device alone; the cores are not as idle as the worker threads are: the launching thread and
	// the runtime's own threads need them.)
	// (Letting the host threads also cover the first 0.4 s of a run -- decode the first batches' sections while the first launch is
	// out -- was measured: the fronts of the frames behind them are then prepared late and the gap only moves; 215 against 187 ms
	// a step over 20 steps.)
	int64_t lf_stage = 0, lf_cap = 0;
	double lf_pending_since = 0;
	int64_t lf_device_frames = 0, single_frames = 0;
	std::mutex m;
	std::condition_variable cv_todo, cv_ready, cv_done;
	std::deque<Job *> todo, ready, lf_pending;   // lf_pending: prepared, their LfGroup streams still to be launched on the device
	LfFlight lf_flights[4];
	int lf_flights_used = 4;            // how many of them launch (J40HIP_LF_FLIGHTS); a launch carries up to lf_flight_frames frames
	int64_t lf_flight_frames = 0;
	int64_t lf_auto_min = 0;            // mode 0: with nothing in the device's stage, fewer frames than this waiting are the host threads' (J40HIP_LF_AUTO_MIN)
	double lf_wait_burst = 10.0;        // ... times this when the frames queued behind them can fill the batch (J40HIP_LF_WAIT_BURST)
	double lf_wait_ms = 5.0;            // how long frames wait for a batch's worth of company before a launch takes them alone (J40HIP_LF_WAIT_MS)
	std::vector<uint32_t> results;      // by ticket
	std::vector<uint8_t> finished;      // by ticket
	int64_t submitted = 0, completed = 0, resident = 0, parsing = 0, in_flight_frames = 0;
	std::vector<j40hip_aframe *> garbage;   // frames of retired batches: the worker threads free them (60 us each: a batch's worth kept the launching thread busy for 15 ms and more)
	int64_t full_batches = 0;
	bool reserve = true;                // size the device memory cache for the full depth at the second full batch (flags bit 3 of create_ex: not)
	bool stop = false;
	std::vector<std::thread> workers;
	std::thread gpu;
	std::vector<Slot> slots;
	std::deque<int> in_flight;          // slot indices, oldest first
	hipStream_t copy_stream = nullptr;  // every copy of pixels back to host memory, in launch order: one DMA queue at the link's rate
	bool sdma_copies = false;           // host output goes back on the SDMA engine measured for the device (hostcopy.hpp)
	std::vector<hipStream_t> copy_streams;   // copy_stream first; J40HIP_COPY_STREAMS=n: the groups of a batch's copies go to n streams in turn
	// device images for host output, recycled by size
	std::mutex image_m;
	std::vector<std::pair<void *, size_t>> free_images;
	double parse_ms = 0, single_ms = 0;  // summed over the worker threads: the asynchronous path's host stage / whole single-frame decodes
	double lf_ms = 0, k1_ms = 0, k2_ms = 0, k1_kernel_ms = 0; int64_t launches = 0, launch_frames = 0;   // HIP-event durations of the batches' stages
	double lf_kernel_ms = 0; int64_t lf_launches = 0, lf_launch_frames = 0, lf_launch_sections = 0, lf_launch_waves = 0;   // the LfGroup lane decoder's launches (device-recorded start / end)
	double first_submit_ms = 0, last_done_ms = 0;
	// threads that could not start (no device, no stream). The pipeline is broken -- queued images fail with "!gpu" -- only when no
	// worker is left or the GPU thread is gone: one worker without a stream does not stop the others from serving
	std::atomic<int> workers_alive{0};
	std::atomic<bool> gpu_thread_dead{false};
	bool broken() const { return gpu_thread_dead.load() || workers_alive.load() <= 0; }
};

namespace {

void complete(j40hip_pipeline *p, Job *j) {   // p->m held
	if (j->ticket >= 0 && (size_t) j->ticket < p->results.size()) { p->results[(size_t) j->ticket] = j->status; p->finished[(size_t) j->ticket] = 1; }
	++p->completed;
	p->last_done_ms = now_ms();
	if (Waiter *w = j->waiter) { std::lock_guard<std::mutex> wl(w->m); w->status = j->status; w->done = true; w->cv.notify_one(); }   // (notified under its lock: the waiter's stack frame may be gone right after)
	delete j;
	if (p->completed == p->submitted || (p->completed & 63) == 0) p->cv_done.notify_all();   // (whoever drains polls as well)
}

void *acquire_image(j40hip_pipeline *p, size_t bytes) {
	{
		std::lock_guard<std::mutex> lock(p->image_m);
		for (size_t i = 0; i < p->free_images.size(); ++i) if (p->free_images[i].second == bytes) { void *q = p->free_images[i].first; p->free_images.erase(p->free_images.begin() + (long) i); return q; }
	}
	void *q = nullptr;
	if (hipMalloc(&q, bytes) != hipSuccess) {
		(void) hipGetLastError();
		j40hip_rt::cache_trim(p->device);   // idle blocks of the frame cache: give them back and try once more
		if (hipMalloc(&q, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	}
	return q;
}
void release_image(j40hip_pipeline *p, void *q, size_t bytes) { std::lock_guard<std::mutex> lock(p->image_m); p->free_images.push_back({q, bytes}); }

// the image's pixel memory, once its size is known (j40hip_pipeline_run: asked for here; j40hip_pipeline_submit: the caller's)
uint32_t ensure_output(Job *j) {
	if (j->rgba) return j->stride < (size_t) j->width * 4 ? E_RNGE : 0;
	if (!j->alloc) return E_RNGE;
	size_t stride = 0;
	j->rgba = j->alloc(j->alloc_ctx, j->width, j->height, &stride);
	j->stride = stride;
	if (!j->rgba) return E_MEM;
	return j->stride < (size_t) j->width * 4 ? E_RNGE : 0;
}

// The single-frame path, synchronous on `s` (the calling thread sleeps in the waits): frames the batches do not take, and frames a
// batch wants decoded again. Parses the image itself.
uint32_t decode_single(j40hip_pipeline *p, Job *j, hipStream_t s) {
	uint32_t err = 0;
	j40hip_frame *fr = j40hip_frame_parse_ex(j->buf, j->size, 1, 1u, &err);
	if (!fr) return err ? err : E_MEM;
	int64_t info[21];
	j40hip_frame_info(fr, info);
	j->width = info[0]; j->height = info[1];
	err = ensure_output(j);
	const size_t bytes = j->stride * (size_t) j->height;
	void *dev = nullptr;
	if (!err) { dev = j->device_output ? j->rgba : acquire_image(p, bytes); if (!dev) err = E_MEM; }
	if (!err) err = j40hip_frame_upload_on(fr, p->device, s);
	if (!err) err = j40hip_frame_decode(fr, dev, j->stride, s);
	if (!err && hipStreamSynchronize(s) != hipSuccess) err = E_GPU;
	if (!err) err = j40hip_frame_status(fr);
	if (err == E_EVOF) {   // a section with more non-zero coefficients than its event region holds: dense planes
		j40hip_frame_force_dense(fr, 1);
		err = j40hip_frame_upload_on(fr, p->device, s);
		if (!err) err = j40hip_frame_decode(fr, dev, j->stride, s);
		if (!err && hipStreamSynchronize(s) != hipSuccess) err = E_GPU;
		if (!err) err = j40hip_frame_status(fr);
	}
	if (!err) err = j40hip_frame_after_frame_status(fr);
	// (the stream has been waited for above: the copy may go to the measured SDMA engine)
	if (!err && !j->device_output && !j40hip_rt::hostcopy_d2h_sync(p->device, j->rgba, dev, bytes) && hipMemcpyAsync(j->rgba, dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) err = E_GPU;
	if (hipStreamSynchronize(s) != hipSuccess && !err) err = E_GPU;
	j40hip_frame_mark_idle(fr);   // its stream has been waited for
	j40hip_frame_free(fr);
	if (dev && !j->device_output) release_image(p, dev, bytes);
	return err;
}

void worker_main(j40hip_pipeline *p, int) {
	if (hipSetDevice(p->device) != hipSuccess) { --p->workers_alive; return; }
	// The thread's copies go on a stream of the highest priority: such streams have hardware queues of their own, so a copy does
	// not wait its turn behind another stream's long kernel (streams of one priority share a handful of hardware queues in turn)
	hipStream_t stream = nullptr;
	int prio_low = 0, prio_high = 0;
	(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
	if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_high) != hipSuccess) { (void) hipGetLastError(); if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); --p->workers_alive; return; } }
	double wait_ms = 0, post_ms = 0, busy_ms = 0; int64_t nframes = 0;   // (J40HIP_ASYNC_TIMING)
	for (;;) {
		Job *j = nullptr;
		bool lf_dev = p->lf_mode == 1;
		const double tw0 = now_ms();
		{
			std::unique_lock<std::mutex> lock(p->m);
			// back-pressure: prepared frames hold their working set in HBM until their batch is done
			p->cv_todo.wait(lock, [&] { return p->stop || !p->garbage.empty() || (!p->todo.empty() && p->resident < (int64_t) p->batch_frames * (p->max_in_flight + 1) + p->lf_cap); });
			if (!p->garbage.empty()) {
				std::vector<j40hip_aframe *> mine;
				for (int k = 0; k < 16 && !p->garbage.empty(); ++k) { mine.push_back(p->garbage.back()); p->garbage.pop_back(); }
				lock.unlock();
				for (j40hip_aframe *af : mine) j40hip_aframe_free(af);
				continue;
			}
			if (p->stop) break;
			j = p->todo.front(); p->todo.pop_front();
			++p->resident; ++p->parsing;
			if (p->lf_mode == 0) lf_dev = p->lf_stage < p->lf_cap && !(p->lf_stage == 0 && (int64_t) p->todo.size() < p->lf_auto_min);
			if (lf_dev) ++p->lf_stage;
		}
		const double t0 = now_ms();
		wait_ms += t0 - tw0;
		j->af = j40hip_aframe_prepare(j->buf, j->size, p->device, stream, lf_dev ? 1 : 0);
		const double t1 = now_ms();
		bool single = j->af == nullptr;
		if (j->af) {
			j40hip_aframe_size(j->af, &j->width, &j->height);
			j->cells = j40hip_aframe_cells(j->af);
			if (uint32_t e = ensure_output(j)) {
				(void) hipStreamSynchronize(stream);   // (its copy is in flight)
				j40hip_aframe_free(j->af); j->af = nullptr;
				j->status = e;
			}
		} else j->status = decode_single(p, j, stream);
		const double t2 = now_ms();
		std::unique_lock<std::mutex> lock(p->m);
		--p->parsing;
		if (single) { p->single_ms += t2 - t0; ++p->single_frames; } else p->parse_ms += t1 - t0;
		if (!j->af) {
			if (lf_dev) --p->lf_stage;
			--p->resident;
			complete(p, j);
			p->cv_todo.notify_all(); p->cv_ready.notify_all();
		} else {
			const int on_dev = j40hip_aframe_lf_on_device(j->af);
			p->lf_device_frames += on_dev;
			if (lf_dev && !on_dev) --p->lf_stage;   // (its tables are not the device decoder's kind)
			j->ready_ms = t2;
			(on_dev ? p->lf_pending : p->ready).push_back(j);
			p->cv_ready.notify_all();
		}
		lock.unlock();
		post_ms += now_ms() - t2; busy_ms += t2 - t0; ++nframes;
	}
	if (getenv("J40HIP_ASYNC_TIMING") && nframes) fprintf(stderr, "[j40hip worker] %lld frames, ms per frame: waiting for a job %.2f, working %.2f, handing over %.2f\n", (long long) nframes, wait_ms / (double) nframes, busy_ms / (double) nframes, post_ms / (double) nframes);
	(void) hipStreamSynchronize(stream);
	j40hip_astage_release();
	j40hip_thread_release();
	(void) hipStreamDestroy(stream);
}

// Host output through the SDMA engine (hostcopy.hpp): the batch's kernels are through, its frames' pixels go back now, in launch order.
// A frame whose destination is not pinned memory goes through hipMemcpyAsync on the copy stream instead, with an event for its group.
void issue_copies(j40hip_pipeline *p, Slot &slot) {
	const int ngroups = (int) slot.group_end.size();
	slot.group_has_ev.assign((size_t) ngroups, 0);
	for (int g = 0; g < ngroups; ++g) {
		const int begin = g ? slot.group_end[(size_t) g - 1] : 0, end = slot.group_end[(size_t) g];
		for (int i = begin; i < end; ++i) {
			Job *j = slot.jobs[(size_t) i];
			if (j->device_output || j->status || j->redo || !j->dev_rgba) continue;
			const size_t bytes = j->stride * (size_t) j->height;
			if (j40hip_rt::hostcopy_d2h(p->device, j->rgba, j->dev_rgba, bytes, &j->copy_ticket) == 0) continue;
			j->copy_ticket = 0;
			if (hipMemcpyAsync(j->rgba, j->dev_rgba, bytes, hipMemcpyDeviceToHost, p->copy_stream) != hipSuccess) { (void) hipGetLastError(); j->status = E_GPU; }
			else slot.group_has_ev[(size_t) g] = 1;
		}
		if (slot.group_has_ev[(size_t) g]) {
			while (slot.group_ev.size() <= (size_t) g) { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { (void) hipGetLastError(); break; } slot.group_ev.push_back(e); }
			if (slot.group_ev.size() <= (size_t) g || hipEventRecord(slot.group_ev[(size_t) g], p->copy_stream) != hipSuccess) {
				(void) hipGetLastError(); (void) hipStreamSynchronize(p->copy_stream); slot.group_has_ev[(size_t) g] = 0;   // (no event to ask: wait here, once)
			}
		}
	}
}

// Hands back the frames of `slot` whose group events have passed (`block`: waits for the next group first). Returns true when the
// slot has nothing pending any more (it is then free for another batch). GPU thread only.
bool progress(j40hip_pipeline *p, Slot &slot, bool block) {
	const int ngroups = (int) slot.group_end.size();
	bool first = true;
	if (!slot.harvested && !slot.launch_err && !slot.failed) {
		// The batch's kernels and the copy of its verdicts are through (kdone): every frame's code is known and its working set --
		// 0.2 GB per 8K frame -- goes back to the cache now, not when the copies of the pixels have drained (0.6 s per 256 frames
		// over PCIe, during which the next batches want that memory)
		hipError_t q = block ? hipEventSynchronize(slot.kdone) : hipEventQuery(slot.kdone);
		if (q == hipErrorNotReady) { (void) hipGetLastError(); return false; }
		first = false;
		if (q != hipSuccess) { (void) hipGetLastError(); slot.failed = true; }
		else {
			float ms3[4] = {0, 0, 0, 0};
			const bool timed = j40hip_abatch_elapsed(slot.batch, ms3) == 0;
			std::vector<j40hip_aframe *> dead;
			for (size_t i = 0; i < slot.jobs.size(); ++i) {
				Job *j = slot.jobs[i];
				if (!j->status) {   // (a copy back that could not be enqueued keeps its error)
					uint32_t code = 0; int redo = 0;
					j40hip_abatch_result(slot.batch, (int) i, &code, &redo);
					j->redo = redo != 0;
					if (!redo) j->status = code ? code : j40hip_aframe_after_frame_status(j->af);
				}
				if (j->af) { dead.push_back(j->af); j->af = nullptr; }
			}
			if (slot.copies_deferred) issue_copies(p, slot);
			std::unique_lock<std::mutex> lock(p->m);
			if (timed) { p->lf_ms += ms3[0]; p->k1_ms += ms3[1]; p->k2_ms += ms3[2]; p->k1_kernel_ms += ms3[3]; ++p->launches; p->launch_frames += (int64_t) slot.jobs.size(); }
			// (the back-pressure on the worker threads counts frames that hold device memory: these no longer do -- what waits for the
			// copy back is bounded by the batch slots)
			for (Job *j : slot.jobs) if (j->counted) { j->counted = false; --p->resident; }
			p->garbage.insert(p->garbage.end(), dead.begin(), dead.end());
			p->cv_todo.notify_all();
		}
		slot.harvested = true; slot.t_harvest = now_ms();
	}
	while (slot.next_group < ngroups) {
		const int begin = slot.next_group ? slot.group_end[(size_t) slot.next_group - 1] : 0, end = slot.group_end[(size_t) slot.next_group];
		if (!slot.launch_err && !slot.failed && (!slot.copies_deferred || slot.group_has_ev[(size_t) slot.next_group])) {
			hipEvent_t ev = slot.group_ev[(size_t) slot.next_group];
			hipError_t q = block && first ? hipEventSynchronize(ev) : hipEventQuery(ev);
			if (q == hipErrorNotReady) { (void) hipGetLastError(); return false; }
			if (q != hipSuccess) { (void) hipGetLastError(); slot.failed = true; }
		}
		if (slot.copies_deferred) {
			// the group's copies on the SDMA engine: issued in order on one engine, so the last one is the one to sleep on
			for (int i = end - 1; i >= begin; --i) {
				Job *j = slot.jobs[(size_t) i];
				if (!j->copy_ticket) continue;
				int st = j40hip_rt::hostcopy_state(j->copy_ticket);
				if (st == 0 && block && first) st = j40hip_rt::hostcopy_wait(j->copy_ticket) ? 1 : -1;
				if (st == 0) return false;
				if (st < 0 && !j->status) j->status = E_GPU;
				j40hip_rt::hostcopy_release(p->device, j->copy_ticket); j->copy_ticket = 0;
			}
		}
		first = false;
		std::vector<j40hip_aframe *> dead;
		for (int i = begin; i < end; ++i) {
			Job *j = slot.jobs[(size_t) i];
			if (slot.failed) j->status = E_GPU;
			else if (slot.launch_err) j->status = slot.launch_err;
			else if (j->redo) {
				if (!j->device_output && j->dev_rgba) { release_image(p, j->dev_rgba, j->stride * (size_t) j->height); j->dev_rgba = nullptr; }
				j->status = decode_single(p, j, slot.stream);
			}
			if (j->af) { dead.push_back(j->af); j->af = nullptr; }
			if (!j->device_output && j->dev_rgba) { release_image(p, j->dev_rgba, j->stride * (size_t) j->height); j->dev_rgba = nullptr; }
		}
		{
			std::unique_lock<std::mutex> lock(p->m);
			p->in_flight_frames -= (int64_t) (end - begin);
			for (int i = begin; i < end; ++i) { if (slot.jobs[(size_t) i]->counted) --p->resident; complete(p, slot.jobs[(size_t) i]); slot.jobs[(size_t) i] = nullptr; }
			p->garbage.insert(p->garbage.end(), dead.begin(), dead.end());
			p->cv_todo.notify_all();
		}
		++slot.next_group;
	}
	static const bool timing = getenv("J40HIP_ASYNC_TIMING") != nullptr;
	if (timing) fprintf(stderr, "[j40hip batch] %zu frames in %d groups: launched at %.1f, kernels + verdicts through after %.1f ms, last pixels back after another %.1f ms\n", slot.jobs.size(), ngroups, slot.t_launch, slot.t_harvest - slot.t_launch, now_ms() - slot.t_harvest);
	slot.jobs.clear(); slot.group_end.clear(); slot.next_group = 0; slot.busy = false; slot.launch_err = 0; slot.failed = false; slot.harvested = false; slot.copies_deferred = false;
	return true;
}

// Enqueues one batch on a free slot: device images for the frames whose pixels go to host memory, the whole decode on the slot's
// stream, the copies back on the copy stream in groups. On "!mem" nothing stays enqueued or bound to the slot and the caller may
// try again (later, or with fewer frames); any other error is the batch's verdict (the slot is pushed and retires with it).
uint32_t launch_batch(j40hip_pipeline *p, std::vector<Job *> &take, int si) {
	Slot &slot = p->slots[(size_t) si];
	std::vector<j40hip_aframe *> frames; std::vector<void *> outs; std::vector<size_t> strides;
	uint32_t err = 0;
	bool host_out = false;
	for (Job *j : take) {
		if (!j->dev_rgba) j->dev_rgba = j->device_output ? j->rgba : acquire_image(p, j->stride * (size_t) j->height);
		if (!j->dev_rgba) err = E_MEM;
		host_out = host_out || !j->device_output;
		frames.push_back(j->af); outs.push_back(j->dev_rgba); strides.push_back(j->stride);
	}
	if (!err && !slot.batch) { slot.batch = j40hip_abatch_create(p->device); if (!slot.batch) err = E_GPU; }
	if (!err) err = j40hip_abatch_launch(slot.batch, frames.data(), (int) frames.size(), outs.data(), strides.data(), slot.stream);
	if (err == E_MEM) {
		// (abatch_launch binds the working sets before it enqueues anything, so nothing is in flight; the frames keep what they got)
		for (Job *j : take) if (!j->device_output && j->dev_rgba) { release_image(p, j->dev_rgba, j->stride * (size_t) j->height); j->dev_rgba = nullptr; }
		return E_MEM;
	}
	slot.busy = true; slot.jobs = take; slot.launch_err = err; slot.failed = false; slot.harvested = false; slot.next_group = 0; slot.group_end.clear(); slot.t_launch = now_ms();
	// the second full batch says this is a pipeline that will run at depth: size the device memory cache for it now (the device
	// has two batches to work on meanwhile) rather than wherever the queues first fill up
	if (!err && p->reserve && (int64_t) frames.size() == p->batch_frames && ++p->full_batches == 2)
		j40hip_aframes_reserve(frames.data(), (int) frames.size(), p->max_in_flight - 1, (int) (p->max_in_flight - 1 + p->lf_cap / std::max<int64_t>(1, p->batch_frames)));   // (two batches' worth exist)
	if (hipEventRecord(slot.kdone, slot.stream) != hipSuccess && !slot.launch_err) slot.launch_err = E_GPU;
	const int n = (int) take.size();
	// groups: without copies the whole batch is one group that ends with the kernels; with copies about sixteen per batch
	static const int groups = [] { const char *e = getenv("J40HIP_COPY_GROUPS"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
	const int per_group = host_out ? std::max(1, (n + groups - 1) / groups) : n;
	hipStream_t gs = host_out ? p->copy_stream : slot.stream;
	// host output: on the SDMA engine measured for this device, issued when the kernels are seen to be through (issue_copies) -- or,
	// without it (J40HIP_COPY_ENGINE=hip, no HSA), hipMemcpyAsync on the copy stream(s) behind the batch's kernels
	slot.copies_deferred = host_out && p->sdma_copies;
	if (host_out && !slot.copies_deferred) for (hipStream_t cs : p->copy_streams) if (!slot.launch_err && hipStreamWaitEvent(cs, slot.kdone, 0) != hipSuccess) slot.launch_err = E_GPU;
	for (int i = 0; i < n; ++i) {
		Job *j = take[(size_t) i];
		if (slot.copies_deferred) { if ((i + 1) % per_group == 0 || i + 1 == n) slot.group_end.push_back(i + 1); continue; }
		if (host_out) gs = p->copy_streams[slot.group_end.size() % p->copy_streams.size()];   // (a group's copies and its event on one stream)
		if (!slot.launch_err && !j->device_output && hipMemcpyAsync(j->rgba, j->dev_rgba, j->stride * (size_t) j->height, hipMemcpyDeviceToHost, gs) != hipSuccess) j->status = E_GPU;
		if ((i + 1) % per_group == 0 || i + 1 == n) {
			const size_t g = slot.group_end.size();
			while (slot.group_ev.size() <= g) { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { (void) hipGetLastError(); break; } slot.group_ev.push_back(e); }
			if (slot.group_ev.size() <= g && !slot.launch_err) slot.launch_err = E_GPU;   // (no event: progress() asks none of a slot with a launch error)
			if (!slot.launch_err && hipEventRecord(slot.group_ev[g], gs) != hipSuccess) slot.launch_err = E_GPU;
			slot.group_end.push_back(i + 1);
		}
	}
	if (slot.launch_err) { (void) hipStreamSynchronize(slot.stream); if (host_out) for (hipStream_t cs : p->copy_streams) (void) hipStreamSynchronize(cs); }   // (whatever did get enqueued: nothing may run on memory that is handed back)
	p->in_flight.push_back(si);
	return slot.launch_err;
}

void gpu_main(j40hip_pipeline *p) {
	if (hipSetDevice(p->device) != hipSuccess) { p->gpu_thread_dead = true; return; }
	double t_lf = 0, t_launch = 0, t_retire = 0, t_idle = 0, t_lock = 0; int64_t n_launch = 0, n_lf = 0;   // (J40HIP_ASYNC_TIMING)
	struct Report { double &a, &b, &c, &d, &e; int64_t &n, &m; ~Report() { if (getenv("J40HIP_ASYNC_TIMING")) fprintf(stderr, "[j40hip gpu thread] ms: LfGroup launches %.1f (%lld), batch launches %.1f (%lld), retiring %.1f, waiting %.1f, for the lock %.1f\n", a, (long long) m, b, (long long) n, c, d, e); } } report{t_lf, t_launch, t_retire, t_idle, t_lock, n_launch, n_lf};
	// slots with something to hand back, oldest first; free slots leave the list
	auto retire_ready = [&](bool block_on_oldest) {
		const double tr = now_ms();
		bool any = false;
		for (size_t k = 0; k < p->in_flight.size(); ) {
			Slot &slot = p->slots[(size_t) p->in_flight[k]];
			const int before = slot.next_group;
			if (progress(p, slot, block_on_oldest && k == 0)) { p->in_flight.erase(p->in_flight.begin() + (long) k); any = true; continue; }
			any = any || slot.next_group != before;
			++k;
		}
		t_retire += now_ms() - tr;
		return any;
	};
	for (;;) {
		std::vector<Job *> take;
		{
			const double tl0 = now_ms();
			std::unique_lock<std::mutex> lock(p->m);
			t_lock += now_ms() - tl0;
			auto lf_busy = [&] { bool b = !p->lf_pending.empty(); for (const LfFlight &fl : p->lf_flights) b = b || fl.busy; return b; };
			auto tail = [&] { return p->stop || (p->todo.empty() && p->parsing == 0 && !lf_busy()); };   // nothing else is coming
			p->cv_ready.wait(lock, [&] { return p->stop || !p->ready.empty() || !p->in_flight.empty() || lf_busy(); });
			if (p->stop && p->ready.empty() && p->in_flight.empty() && p->parsing == 0 && !lf_busy()) break;
			// the LfGroup streams the device decodes: finished launches hand their frames on; waiting frames go into the next launch (a
			// launch is latency-bound -- about 0.2 s however many sections it has -- so everything waiting goes in)
			for (LfFlight &fl : p->lf_flights) if (fl.busy && j40hip_alf_done(fl.alf)) {
				{ float ms = 0; int nf = 0, ns = 0, nw = 0; if (j40hip_alf_elapsed(fl.alf, &ms, &nf, &ns, &nw) == 0) { p->lf_kernel_ms += ms; ++p->lf_launches; p->lf_launch_frames += nf; p->lf_launch_sections += ns; p->lf_launch_waves += nw; } }
				for (Job *j : fl.jobs) { j->ready_ms = now_ms(); p->ready.push_back(j); }
				p->lf_stage -= (int64_t) fl.jobs.size();
				fl.jobs.clear(); fl.busy = false;
				p->cv_todo.notify_all();
			}
			// (a launch takes its 0.3 s whether it carries one frame or two batches' worth: wait for a batch's worth unless no launch is in
			// flight at all and the frames have been waiting a while, or nothing else is coming)
			if (p->lf_pending.empty()) p->lf_pending_since = 0;
			else if (p->lf_pending_since == 0) p->lf_pending_since = now_ms();
			bool lf_flying = false;
			for (const LfFlight &fl : p->lf_flights) lf_flying = lf_flying || fl.busy;
			// (... "a while": lf_wait_ms when what is queued behind them cannot fill the batch anyway; ten times that when it can -- the
			// head of a burst: 57 frames left after 5 ms, the batch they belong to then waited for the SECOND launch, 0.17 s later, to
			// bring its other 199, call K's timeline)
			const int64_t coming = (int64_t) p->todo.size() + p->parsing, missing = p->batch_frames - (int64_t) p->lf_pending.size();
			const double lf_waited = p->lf_pending.empty() ? 0.0 : now_ms() - p->lf_pending_since;
			const bool lf_go = !p->lf_pending.empty() && (missing <= 0 || (!lf_flying && lf_waited > (coming >= missing ? p->lf_wait_burst : 1.0) * p->lf_wait_ms) || p->stop || (p->todo.empty() && p->parsing == 0));
			if (lf_go) for (int fi = 0; fi < p->lf_flights_used; ++fi) if (!p->lf_flights[fi].busy) {
				LfFlight &fl = p->lf_flights[fi];
				std::vector<j40hip_aframe *> frames;
				while (!p->lf_pending.empty() && (int64_t) fl.jobs.size() < p->lf_flight_frames) { fl.jobs.push_back(p->lf_pending.front()); frames.push_back(p->lf_pending.front()->af); p->lf_pending.pop_front(); }
				if (!fl.alf) fl.alf = j40hip_alf_create(p->device);
				const double ta = now_ms();
				const uint32_t e = fl.alf ? j40hip_alf_launch(fl.alf, frames.data(), (int) frames.size(), fl.stream) : E_GPU;
				t_lf += now_ms() - ta; ++n_lf;
				if (e) { for (Job *j : fl.jobs) { j->lf_failed = true; p->ready.push_back(j); } p->lf_stage -= (int64_t) fl.jobs.size(); fl.jobs.clear(); }   // (they are decoded again on the single-frame path)
				else fl.busy = true;
				break;
			}
			// This thread never waits for the device while there may be something to enqueue -- a finished LfGroup launch to replace, a
			// batch to launch: batches in flight are handed back group by group as their events pass (polled), and waited for only when
			// nothing else can happen.
			if ((int) p->in_flight.size() < p->max_in_flight) {   // (a launch goes before a retirement: the device should not wait for this thread's bookkeeping)
				const bool waited = p->max_wait_ms > 0 && !p->ready.empty() && now_ms() - p->ready.front()->ready_ms >= p->max_wait_ms;
				const int64_t want = (int64_t) p->ready.size() >= p->batch_frames ? p->batch_frames : (!p->ready.empty() && (waited || tail())) ? (int64_t) p->ready.size() : 0;
				for (int64_t i = 0; i < want; ++i) { take.push_back(p->ready.front()); p->ready.pop_front(); }
				p->in_flight_frames += (int64_t) take.size();
			}
			if (take.empty()) {
				// (polled, never slept on: a frame that becomes ready meanwhile must not wait for a batch in flight; the workers' notifications
				// end the wait early)
				lock.unlock();
				if (retire_ready(false)) continue;
				lock.lock();
				const double tw = now_ms(); p->cv_ready.wait_for(lock, std::chrono::milliseconds(1)); t_idle += now_ms() - tw;
				continue;
			}
		}
		{   // frames whose LfGroup streams could not be launched on the device never enter a batch (their planes were never decoded):
			// the single-frame path, here and now
			std::vector<Job *> keep;
			for (Job *j : take) {
				if (!j->lf_failed) { keep.push_back(j); continue; }
				(void) hipDeviceSynchronize();
				j40hip_aframe_free(j->af); j->af = nullptr;
				j->status = decode_single(p, j, p->slots[0].stream);
				std::unique_lock<std::mutex> lock(p->m);
				--p->in_flight_frames; --p->resident;
				complete(p, j);
				p->cv_todo.notify_all();
			}
			take.swap(keep);
			if (take.empty()) continue;
		}
		const double tb = now_ms();
		for (;;) {
			int si = -1;
			for (size_t i = 0; i < p->slots.size(); ++i) if (!p->slots[i].busy) { si = (int) i; break; }
			if (launch_batch(p, take, si) != E_MEM) break;
			// Out of device memory with the working sets bound at launch: wait for a batch in flight to hand its memory back and try
			// again; with nothing in flight give idle cache blocks back and halve the batch (the rest goes back to the queue); a single
			// frame that still does not fit fails with "!mem"
			if (!p->in_flight.empty()) { retire_ready(true); continue; }
			{   // (what harvested batches handed back waits for the worker threads to free it: free it here before concluding anything)
				std::vector<j40hip_aframe *> dead;
				{ std::unique_lock<std::mutex> lock(p->m); dead.swap(p->garbage); }
				for (j40hip_aframe *af : dead) j40hip_aframe_free(af);
				if (!dead.empty()) continue;
			}
			j40hip_rt::cache_trim(p->device);
			if (take.size() > 1) {
				std::unique_lock<std::mutex> lock(p->m);
				const size_t keep = take.size() / 2;
				while (take.size() > keep) { p->ready.push_front(take.back()); take.pop_back(); --p->in_flight_frames; }
				continue;
			}
			Slot &slot = p->slots[(size_t) si];   // one frame, nothing in flight, nothing cached: it does not fit
			slot.busy = true; slot.jobs = take; slot.launch_err = E_MEM; slot.harvested = false; slot.next_group = 0; slot.group_end.assign(1, (int) take.size());   // (progress() asks no event of a slot with a launch error)
			p->in_flight.push_back(si);
			break;
		}
		t_launch += now_ms() - tb; ++n_launch;
	}
	{ std::unique_lock<std::mutex> lock(p->m); for (j40hip_aframe *af : p->garbage) j40hip_aframe_free(af); p->garbage.clear(); }
	for (Slot &s : p->slots) if (s.batch) { j40hip_abatch_free(s.batch); s.batch = nullptr; }
	for (LfFlight &fl : p->lf_flights) if (fl.alf) { j40hip_alf_free(fl.alf); fl.alf = nullptr; }
	for (auto &im : p->free_images) (void) hipFree(im.first);
	p->free_images.clear();
}

} // namespace

extern "C" {

j40hip_pipeline *j40hip_pipeline_create(int device, int host_threads, int batch_frames, int max_in_flight, uint32_t *err) { return j40hip_pipeline_create_ex(device, host_threads, batch_frames, max_in_flight, 0, err); }

j40hip_pipeline *j40hip_pipeline_create_ex(int device, int host_threads, int batch_frames, int max_in_flight, uint32_t flags, uint32_t *err) {
	uint32_t dummy; if (!err) err = &dummy;
	*err = 0;
	if (j40hip_device_count() <= device || device < 0 || hipSetDevice(device) != hipSuccess) { *err = E_GPU; return nullptr; }
	j40hip_pipeline *p = nullptr;
	try {
		p = new j40hip_pipeline();
		p->device = device;
		p->lf_mode = (int) (flags & 3u) > 2 ? 0 : (int) (flags & 3u);
		// (flags bit 2, opt-in: keep multi-megabyte blocks in the heap instead of separate mmap()s -- many threads freeing such blocks
		// serialise on the process's address-space lock and fault every page in again. Changes process-wide malloc behaviour.)
		p->reserve = !(flags & 8u);
		if (flags & 4u) { (void) mallopt(M_MMAP_THRESHOLD, 1 << 30); (void) mallopt(M_TRIM_THRESHOLD, (int) (((size_t) 1 << 31) - 1)); (void) mallopt(M_TOP_PAD, 64 << 20); }
		p->batch_frames = batch_frames < 1 ? 32 : batch_frames;
		p->max_in_flight = max_in_flight < 1 ? 2 : max_in_flight > 8 ? 8 : max_in_flight;
		// (a frame in this stage holds about 12 MB of device memory; eight batches' worth, at most 2048 frames or two batches')
		p->lf_cap = p->lf_mode == 2 ? 0 : std::min<int64_t>((int64_t) p->batch_frames * 8, std::max<int64_t>(2048, (int64_t) p->batch_frames * 2));
		if (const char *e = getenv("J40HIP_LF_CAP")) p->lf_cap = atoll(e);
		// LfGroup launches in flight at once, and frames per launch. A launch lasts 0.4-0.8 s whatever it carries and occupies a hardware
		// queue of its own for that long; with four of them active beside the batches' streams, the pixel-kernel streams and the copy
		// stream the process has more active queues than the device schedules at once, and the copies back -- blit kernels on their
		// queue -- crawled (2.6-5.3 s per 256-frame batch instead of 0.6 s for the first twenty seconds of a long run, until the
		// LfGroup stage had run ahead: DESIGN.md section 5). Two flights of up to four batches' worth each.
		p->lf_flights_used = 2; p->lf_flight_frames = (int64_t) p->batch_frames * 4;
		if (const char *e = getenv("J40HIP_LF_FLIGHTS")) p->lf_flights_used = std::max(1, std::min(4, atoi(e)));
		if (const char *e = getenv("J40HIP_LF_FLIGHT_FRAMES")) p->lf_flight_frames = std::max<int64_t>(1, atoll(e));
		if (const char *e = getenv("J40HIP_LF_WAIT_MS")) p->lf_wait_ms = std::max(0.0, atof(e));
		if (const char *e = getenv("J40HIP_LF_WAIT_BURST")) p->lf_wait_burst = std::max(1.0, atof(e));
		p->lf_auto_min = p->batch_frames / 4;
		if (const char *e = getenv("J40HIP_LF_AUTO_MIN")) p->lf_auto_min = std::max<int64_t>(0, atoll(e));
		p->slots.resize((size_t) p->max_in_flight + 1);
		for (Slot &s : p->slots) {
			bool made = false;
			if (j40hip_stream_layout() == 2 && &s != &p->slots[0]) { s.stream = p->slots[0].stream; made = true; }   // layout 2: every batch on ONE stream (one after the other), the pixel-kernel streams shared as in 1
			else if (j40hip_stream_layout() >= 1) { int lo = 0, hi = 0; made = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, hi) == hipSuccess; if (!made) (void) hipGetLastError(); }
			if (!made && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { *err = E_GPU; break; }
			if (hipEventCreateWithFlags(&s.kdone, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { *err = E_GPU; break; }
		}
		if (!*err) {
			// (J40HIP_COPY_PRIORITY: low | normal | high -- which set of hardware queues the copy stream's event markers go through)
			int lo = 0, hi = 0; (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
			const char *e = getenv("J40HIP_COPY_PRIORITY");
			const int prio = e && !strcmp(e, "low") ? lo : e && !strcmp(e, "high") ? hi : (lo + hi) / 2;
			// (J40HIP_COPY_STREAM=mask: a stream with a CU mask -- all CUs -- has a hardware queue of its own, shared with no other stream;
			// J40HIP_COPY_STREAMS=n: n such streams, a batch's copy groups dealt out in turn)
			const char *kind = getenv("J40HIP_COPY_STREAM");
			const int ncopy = [] { const char *c = getenv("J40HIP_COPY_STREAMS"); return c && atoi(c) > 0 ? std::min(8, atoi(c)) : 1; }();
			for (int k = 0; k < ncopy && !*err; ++k) {
				hipStream_t cs = nullptr;
				if (kind && !strcmp(kind, "mask")) {
					hipDeviceProp_t pr;
					if (hipGetDeviceProperties(&pr, p->device) == hipSuccess) {
						const int cus = pr.multiProcessorCount;
						std::vector<uint32_t> mask((size_t) (cus + 31) / 32, 0xffffffffu);
						if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
						if (hipExtStreamCreateWithCUMask(&cs, (uint32_t) mask.size(), mask.data()) != hipSuccess) { (void) hipGetLastError(); cs = nullptr; }
					}
				}
				if (!cs && hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio) != hipSuccess) { (void) hipGetLastError(); if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) *err = E_GPU; }
				if (cs) p->copy_streams.push_back(cs);
			}
			if (!p->copy_streams.empty()) p->copy_stream = p->copy_streams[0];
			p->sdma_copies = j40hip_rt::hostcopy_engine(p->device, nullptr, nullptr) >= 0;   // (measures the engines on the process's first pipeline)
		}
		{   // The LfGroup launches run for a quarter of a second each. Streams of one priority share a handful of hardware queues, and a
			// kernel waits for the kernels ahead of it in its queue whichever stream they came from: on a stream of the batches' priority
			// such a launch held up a quarter of the pixel kernels (296 ms per batch against 80). Lowest priority: queues of their own.
			int prio_low = 0, prio_high = 0;
			(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
			for (LfFlight &fl : p->lf_flights) if (!*err && hipStreamCreateWithPriority(&fl.stream, hipStreamNonBlocking, prio_low) != hipSuccess) {
				(void) hipGetLastError();
				if (hipStreamCreateWithFlags(&fl.stream, hipStreamNonBlocking) != hipSuccess) *err = E_GPU;
			}
		}
		if (!*err) {
			// (default: the container's CPU quota less two, at most 16 -- never the visible CPU count: a quota-limited container that runs
			// more busy threads than its quota gets ALL its threads throttled, the HIP runtime's included)
			if (host_threads < 1) host_threads = std::max(2, std::min(16, cpu_quota() - 2));
			if (host_threads > 128) host_threads = 128;   // (each worker owns tens of MB of pinned staging; more than this was never exercised)
			p->gpu = std::thread(gpu_main, p);
			p->workers_alive = host_threads;
			for (int i = 0; i < host_threads; ++i) p->workers.emplace_back(worker_main, p, i);
		}
	} catch (const std::exception &) { *err = E_MEM; }
	if (*err) { if (p) j40hip_pipeline_free(p); return nullptr; }
	return p;
}

// Frames still queued are dropped (their tickets never complete); frames already prepared or in flight are decoded first.
void j40hip_pipeline_free(j40hip_pipeline *p) {
	if (!p) return;
	{ std::unique_lock<std::mutex> lock(p->m); p->stop = true; p->cv_todo.notify_all(); p->cv_ready.notify_all(); }
	for (std::thread &t : p->workers) if (t.joinable()) t.join();
	{ std::unique_lock<std::mutex> lock(p->m); p->cv_ready.notify_all(); }
	if (p->gpu.joinable()) p->gpu.join();
	(void) hipSetDevice(p->device);
	// (images that never ran: a thread asleep in j40hip_pipeline_run on one of them is woken with "!gpu" rather than left there)
	auto abandon = [](Job *j) { if (Waiter *w = j->waiter) { std::lock_guard<std::mutex> wl(w->m); w->status = E_GPU; w->done = true; w->cv.notify_one(); } delete j; };
	for (Job *j : p->todo) abandon(j);
	for (std::deque<Job *> *q : {&p->ready, &p->lf_pending}) for (Job *j : *q) { if (j->af) { (void) hipDeviceSynchronize(); j40hip_aframe_free(j->af); } abandon(j); }
	for (LfFlight &fl : p->lf_flights) { for (Job *j : fl.jobs) { if (j->af) { (void) hipDeviceSynchronize(); j40hip_aframe_free(j->af); } abandon(j); } if (fl.stream) (void) hipStreamDestroy(fl.stream); }
	for (Slot &s : p->slots) {
		for (hipEvent_t e : s.group_ev) if (e) (void) hipEventDestroy(e);
		if (s.kdone) (void) hipEventDestroy(s.kdone);
		if (s.stream && (&s == &p->slots[0] || s.stream != p->slots[0].stream)) (void) hipStreamDestroy(s.stream);
	}
	for (hipStream_t cs : p->copy_streams) if (cs) (void) hipStreamDestroy(cs);
	delete p;
}

uint32_t j40hip_pipeline_submit(j40hip_pipeline *p, const void *buf, size_t size, void *rgba, size_t stride_bytes, int device_output, int64_t *ticket) {
	if (!p || !buf || !rgba) return E_RNGE;
	if (p->broken()) return E_GPU;
	try {
		Job *j = new Job();
		j->buf = buf; j->size = size; j->rgba = rgba; j->stride = stride_bytes; j->device_output = device_output != 0;
		std::unique_lock<std::mutex> lock(p->m);
		j->ticket = (int64_t) p->results.size(); ++p->submitted;
		p->results.push_back(0); p->finished.push_back(0);
		if (p->first_submit_ms == 0) p->first_submit_ms = now_ms();
		if (ticket) *ticket = j->ticket;
		p->todo.push_back(j);
		p->cv_todo.notify_one();
	} catch (const std::exception &) { return E_MEM; }
	return 0;
}

// One image, synchronously: queued like j40hip_pipeline_submit's, the calling thread sleeps until it is done. The pixel memory is
// asked for through `alloc` once the image's size is known. What j40_next_frame calls when it serves many threads (api.cpp).
uint32_t j40hip_pipeline_run(j40hip_pipeline *p, const void *buf, size_t size, j40hip_output_alloc alloc, void *ctx) {
	if (!p || !buf || !alloc) return E_RNGE;
	if (p->broken()) return E_GPU;
	Waiter w;
	Job *j = nullptr;
	try {
		j = new Job();
		j->buf = buf; j->size = size; j->alloc = alloc; j->alloc_ctx = ctx; j->waiter = &w; j->ticket = -1;
		std::unique_lock<std::mutex> lock(p->m);
		++p->submitted;
		if (p->first_submit_ms == 0) p->first_submit_ms = now_ms();
		p->todo.push_back(j);
		p->cv_todo.notify_one();
	} catch (const std::exception &) { delete j; return E_MEM; }
	std::unique_lock<std::mutex> wl(w.m);
	while (!w.done) {
		w.cv.wait_for(wl, std::chrono::milliseconds(200));
		if (w.done || !p->broken()) continue;
		// no worker could start, or the GPU thread could not: an image still in a queue nobody serves comes out again and fails with
		// "!gpu" -- the queue of the workers, or (GPU thread gone, workers alive) the queues behind them; an image a live thread holds
		// completes as usual
		wl.unlock();
		{
			std::unique_lock<std::mutex> lock(p->m);
			std::deque<Job *> *queues[3] = {&p->todo, &p->ready, &p->lf_pending};
			for (int qi = 0; qi < (p->gpu_thread_dead.load() ? 3 : 1); ++qi) for (auto it = queues[qi]->begin(); it != queues[qi]->end(); ++it) if (*it == j) {
				queues[qi]->erase(it); ++p->completed; p->cv_done.notify_all();
				if (qi) { --p->resident; if (qi == 2) --p->lf_stage; }
				j40hip_aframe *af = j->af;
				delete j;
				lock.unlock();
				if (af) { (void) hipSetDevice(p->device); (void) hipDeviceSynchronize(); j40hip_aframe_free(af); }
				return E_GPU;
			}
			p->cv_ready.notify_all();
		}
		wl.lock();
	}
	return w.status;
}

void j40hip_pipeline_set_max_wait_ms(j40hip_pipeline *p, double ms) { if (p) { std::unique_lock<std::mutex> lock(p->m); p->max_wait_ms = ms; } }

uint32_t j40hip_pipeline_drain(j40hip_pipeline *p) {
	if (!p) return E_RNGE;
	std::unique_lock<std::mutex> lock(p->m);
	p->cv_ready.notify_all();
	while (p->completed < p->submitted) {
		if (p->broken()) return E_GPU;
		p->cv_done.wait_for(lock, std::chrono::milliseconds(50));
		p->cv_ready.notify_all();   // (the tail condition of the GPU thread depends on counters the workers change)
	}
	return 0;
}

uint32_t j40hip_pipeline_result(j40hip_pipeline *p, int64_t ticket) {
	if (!p) return E_RNGE;
	std::unique_lock<std::mutex> lock(p->m);
	if (ticket < 0 || (size_t) ticket >= p->results.size() || !p->finished[(size_t) ticket]) return E_RNGE;
	return p->results[(size_t) ticket];
}

/* see include/j40hip.h */
void j40hip_pipeline_stats(j40hip_pipeline *p, double *out) {
	if (!p || !out) return;
	std::unique_lock<std::mutex> lock(p->m);
	out[0] = p->parse_ms; out[1] = p->single_ms; out[2] = (double) p->completed; out[3] = p->last_done_ms - p->first_submit_ms;
	out[4] = p->k1_ms; out[5] = p->k2_ms; out[6] = (double) p->launches; out[7] = (double) p->launch_frames;
}
void j40hip_pipeline_stats_ex(j40hip_pipeline *p, double *out) {
	if (!p || !out) return;
	j40hip_pipeline_stats(p, out);
	std::unique_lock<std::mutex> lock(p->m);
	out[8] = p->lf_ms; out[9] = (double) p->lf_device_frames; out[10] = (double) p->single_frames; out[11] = p->k1_kernel_ms;
}
/* out[0..4]: the LfGroup lane decoder's launches since the last reset -- summed kernel duration in ms (device-recorded start / end
   events), launches, frames, sections, wavefronts */
void j40hip_pipeline_lf_stats(j40hip_pipeline *p, double *out) {
	if (!p || !out) return;
	std::unique_lock<std::mutex> lock(p->m);
	out[0] = p->lf_kernel_ms; out[1] = (double) p->lf_launches; out[2] = (double) p->lf_launch_frames; out[3] = (double) p->lf_launch_sections; out[4] = (double) p->lf_launch_waves;
}

int64_t j40hip_pipeline_lf_device_frames(j40hip_pipeline *p) { if (!p) return 0; std::unique_lock<std::mutex> lock(p->m); return p->lf_device_frames; }

void j40hip_pipeline_reset_stats(j40hip_pipeline *p) {
	if (!p) return;
	std::unique_lock<std::mutex> lock(p->m);
	p->parse_ms = p->single_ms = 0; p->first_submit_ms = 0; p->last_done_ms = 0;
	p->lf_ms = p->k1_ms = p->k2_ms = p->k1_kernel_ms = 0; p->launches = p->launch_frames = 0; p->lf_device_frames = p->single_frames = 0;
	p->lf_kernel_ms = 0; p->lf_launches = p->lf_launch_frames = p->lf_launch_sections = p->lf_launch_waves = 0;
}


// ---- the process-wide pipelines behind the public API (api.cpp): j40_next_frame hands its image to the pipeline of its device when
// several threads are inside the API at once (or J40HIP_SERVE=1), so that callers of the unchanged ten-function sequence share
// batches. One per device, made on first use, taken down by j40hip_shutdown. Knobs (environment, read once): J40HIP_SERVE_THREADS
// (host threads; default: half the container's CPU quota), J40HIP_SERVE_BATCH (frames per entropy launch, 64), J40HIP_SERVE_IN_FLIGHT (6),
// J40HIP_SERVE_LF (host | device | auto: who decodes the LfGroup streams; auto -- bursts of twelve frames per pipeline thread and more go
// to the device's lane decoder, smaller ones to the host threads: a frame should not wait 0.2 s for a launch it has to itself),
// J40HIP_SERVE_WAIT_MS (how long a prepared frame waits for a fuller batch while a slot is free, 100: with blocking callers the
// queue stops growing when every caller has an image in it, and "nothing else is coming" launches the batch at once).
static std::mutex g_serve_mutex;
static j40hip_pipeline *g_serve[16] = {nullptr};

static int cpu_quota() { return j40hip_cpu_quota(); }   // (capi_host.cpp)

j40hip_pipeline *j40hip_serve_pipeline(int device, uint32_t *err) {
	uint32_t dummy; if (!err) err = &dummy;
	*err = 0;
	if (device < 0 || device >= 16) { *err = E_GPU; return nullptr; }
	std::lock_guard<std::mutex> lock(g_serve_mutex);
	if (g_serve[device]) return g_serve[device];
	auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; };
	// (half of the container's CPU quota: the callers, the launching thread and the HIP runtime's own threads need the rest -- a
	// process that runs into its quota has ALL its threads throttled and the copies back crawl. 64 callers over 8K streams on a
	// 16-CPU quota: 9.9 Gpixel/s with 6 or 8 pipeline threads, 8.2 with 12, 6.4 with 16, 6.2 with 4)
	const int threads = std::max(1, env_int("J40HIP_SERVE_THREADS", std::max(2, cpu_quota() / 2)));
	// Who decodes the LfGroup streams of the served frames: a host thread takes 12 ms per 8K frame, the device's lane decoder 0.17-0.2 s
	// per launch whatever it carries. Round 5, 8K streams, 8 pipeline threads: 64 callers 9.9 Gpixel/s with the host threads against 4.1
	// with the device, 128 callers 3.9 (p90 1.4 s) against 7.7 (p90 0.57 s). So "auto": a burst of at least twelve frames per pipeline
	// thread goes to the device (and what arrives while its stage is busy follows), smaller ones stay with the host threads.
	uint32_t lf = 0;
	if (const char *e = getenv("J40HIP_SERVE_LF")) lf = !strcmp(e, "device") ? 1u : !strcmp(e, "auto") ? 0u : 2u;
	j40hip_pipeline *p = j40hip_pipeline_create_ex(device, threads, std::max(1, env_int("J40HIP_SERVE_BATCH", 64)), env_int("J40HIP_SERVE_IN_FLIGHT", 6), lf | 8u, err);
	if (!p) return nullptr;
	if (!getenv("J40HIP_LF_AUTO_MIN")) { std::unique_lock<std::mutex> plock(p->m); p->lf_auto_min = 12 * (int64_t) threads; }
	const char *w = getenv("J40HIP_SERVE_WAIT_MS");
	j40hip_pipeline_set_max_wait_ms(p, w && *w ? atof(w) : 100.0);
	return g_serve[device] = p;
}

void j40hip_serve_shutdown(void) {
	std::lock_guard<std::mutex> lock(g_serve_mutex);
	for (j40hip_pipeline *&p : g_serve) if (p) { j40hip_pipeline_free(p); p = nullptr; }
}

} // extern "C"
