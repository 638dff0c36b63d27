// j40_amd/csrc/device/pipeline.hip -- whole-frame throughput pipeline over the thin C-ABI (include/j40hip.h, j40hip_pipeline_*):
// codestream bytes in host memory -> RGBA u8x4 in device or host memory, with every stage of many frames in flight at once.
//
//   host worker threads   container / header / TOC / LfGlobal / HfGlobal / LfGroup parse (j40hip_frame_parse: the reference's
//                         j40.h:8175-8192 + 7840-7846 work), plan build, plan upload through the thread's pinned staging buffer
//                         on the thread's own HIP stream (j40hip_frame_upload_on)
//   one GPU thread        collects uploaded frames into batches, one entropy launch + pixel kernels per batch on the batch slot's
//                         stream (j40hip_batch_reset / j40hip_batch_decode), then -- host output -- the copy back on the same
//                         stream, so that the copy of batch k overlaps the kernels of batch k + 1 on the other slot's stream
//   completion            per-frame status words come back with an asynchronous copy; frames whose sections overflow their event
//                         region ("evof") or that carry extra channels / are Modular take the single-frame path
//
// The reference decodes one image on one core (j40.h:8034: its only threading hook is commented out); this is the serving
// shape of the same work: frames are independent, so the host part scales over cores and the device part over a batch.
// Nothing here touches the oracle; without a HIP device creation fails with "!gpu".
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "../../../include/j40hip.h"

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
	j40hip_frame *frame = nullptr;
	bool single = false;          // not batchable (Modular frame, extra channels): decoded on its own
	int64_t width = 0, height = 0;
	void *dev_rgba = nullptr;     // host output: the device image the copy back reads
	uint32_t status = 0;
};

struct Slot {                     // one batch in flight
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;
	j40hip_batch *batch = nullptr;
	std::vector<Job *> jobs;
	int64_t batched = 0;          // members of the batch launch (jobs minus the single-frame ones)
	bool busy = false;
};

} // namespace

struct j40hip_pipeline {
	int device = 0, batch_frames = 32, max_in_flight = 2;
	bool lf_on_device = false;          // the workers parse with j40hip_frame_parse_on: LfGroup streams decoded by the device
	int64_t lf_device_frames = 0;
	std::mutex m;
	std::condition_variable cv_todo, cv_ready, cv_done;
	std::deque<Job *> todo, ready;
	std::vector<uint32_t> results;      // by ticket
	std::vector<uint8_t> finished;      // by ticket
	int64_t submitted = 0, completed = 0, resident = 0, parsing = 0;
	bool stop = false;
	std::vector<std::thread> workers;
	std::thread gpu;
	std::vector<Slot> slots;
	std::deque<int> in_flight;          // slot indices, oldest first
	// device images for host output, recycled by size
	std::vector<std::pair<void *, size_t>> free_images;
	double parse_ms = 0, upload_ms = 0;  // summed over the worker threads
	double k1_ms = 0, k2_ms = 0; int64_t launches = 0, launch_frames = 0;   // HIP-event durations of the batches' entropy / pixel stages
	double first_submit_ms = 0, last_done_ms = 0;
	std::atomic<int> worker_errors{0};
};

namespace {

void complete(j40hip_pipeline *p, Job *j) {   // p->m held
	if ((size_t) j->ticket < p->results.size()) { p->results[(size_t) j->ticket] = j->status; p->finished[(size_t) j->ticket] = 1; }
	++p->completed;
	p->last_done_ms = now_ms();
	delete j;
	p->cv_done.notify_all();
}

void worker_main(j40hip_pipeline *p) {
	if (hipSetDevice(p->device) != hipSuccess) { ++p->worker_errors; return; }
	hipStream_t stream = nullptr;
	if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { ++p->worker_errors; return; }
	for (;;) {
		Job *j = nullptr;
		{
			std::unique_lock<std::mutex> lock(p->m);
			// back-pressure: uploaded frames hold their working set in HBM until their batch is done
			p->cv_todo.wait(lock, [&] { return p->stop || (!p->todo.empty() && p->resident < (int64_t) p->batch_frames * (p->max_in_flight + 1)); });
			if (p->stop) break;
			j = p->todo.front(); p->todo.pop_front();
			++p->resident; ++p->parsing;
		}
		const double t0 = now_ms();
		uint32_t err = 0;
		// (the LfGroup tail runs on the device, at upload; with lf_on_device the LfGroup streams too, while this thread sleeps)
		j->frame = p->lf_on_device ? j40hip_frame_parse_on(j->buf, j->size, 1, 1u, p->device, stream, &err) : j40hip_frame_parse_ex(j->buf, j->size, 1, 1u, &err);
		const bool lf_dev = j->frame && j40hip_frame_lf_on_device(j->frame);
		const double t1 = now_ms();
		if (j->frame) {
			int64_t info[21];
			j40hip_frame_info(j->frame, info);
			j->width = info[0]; j->height = info[1];
			j->single = info[2] != 0 || info[19] != 0;   // Modular frame, or a VarDCT frame with extra channels
			if (j->stride < (size_t) j->width * 4) err = E_RNGE;
			if (!err) err = j40hip_frame_upload_on(j->frame, p->device, stream);
		}
		const double t2 = now_ms();
		std::unique_lock<std::mutex> lock(p->m);
		p->parse_ms += t1 - t0; p->upload_ms += t2 - t1; p->lf_device_frames += lf_dev ? 1 : 0;
		--p->parsing;
		if (err) {
			if (j->frame) { j40hip_frame_mark_idle(j->frame); j40hip_frame_free(j->frame); j->frame = nullptr; }
			j->status = err;
			--p->resident;
			complete(p, j);
			p->cv_todo.notify_all(); p->cv_ready.notify_all();
		} else {
			p->ready.push_back(j);
			p->cv_ready.notify_all();
		}
	}
	j40hip_thread_release();
	(void) hipStreamDestroy(stream);
}

void *acquire_image(j40hip_pipeline *p, size_t bytes) {   // GPU thread only
	for (size_t i = 0; i < p->free_images.size(); ++i) if (p->free_images[i].second == bytes) { void *q = p->free_images[i].first; p->free_images.erase(p->free_images.begin() + (long) i); return q; }
	void *q = nullptr;
	if (hipMalloc(&q, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	return q;
}

// the single-frame path, synchronous: frames a batch cannot take, and the dense-plane repeat after "evof"
uint32_t decode_single(j40hip_pipeline *p, Job *j, hipStream_t s) {
	uint32_t err = j40hip_frame_decode(j->frame, j->dev_rgba, j->stride, s);
	if (!err && hipStreamSynchronize(s) != hipSuccess) err = E_GPU;
	if (!err) err = j40hip_frame_status(j->frame);
	if (err == E_EVOF) {
		j40hip_frame_force_dense(j->frame, 1);
		err = j40hip_frame_upload_on(j->frame, p->device, s);
		if (!err) err = j40hip_frame_decode(j->frame, j->dev_rgba, j->stride, s);
		if (!err && hipStreamSynchronize(s) != hipSuccess) err = E_GPU;
		if (!err) err = j40hip_frame_status(j->frame);
	}
	if (!err) err = j40hip_frame_after_frame_status(j->frame);
	if (!err && !j->device_output && hipMemcpyAsync(j->rgba, j->dev_rgba, j->stride * (size_t) j->height, hipMemcpyDeviceToHost, s) != hipSuccess) err = E_GPU;
	if (!err && hipStreamSynchronize(s) != hipSuccess) err = E_GPU;
	return err;
}

void retire(j40hip_pipeline *p, Slot &slot) {   // GPU thread; waits for the slot's work, then hands the results out
	const bool ok = hipEventSynchronize(slot.done) == hipSuccess;
	float ms3[3] = {0, 0, 0};
	const bool timed = ok && slot.batch && slot.batched > 0 && j40hip_batch_elapsed(slot.batch, 0, ms3) == 0;
	for (Job *j : slot.jobs) {
		if (!ok) j->status = E_GPU;
		else if (!j->single) {
			j->status = j40hip_frame_status_end(j->frame);
			if (j->status == E_EVOF) j->status = decode_single(p, j, slot.stream);
			else if (!j->status) j->status = j40hip_frame_after_frame_status(j->frame);
		}
		j40hip_frame_mark_idle(j->frame);   // its stream has been waited for
		j40hip_frame_free(j->frame); j->frame = nullptr;
		if (!j->device_output && j->dev_rgba) p->free_images.push_back({j->dev_rgba, j->stride * (size_t) j->height});
	}
	std::unique_lock<std::mutex> lock(p->m);
	if (timed) { p->k1_ms += ms3[0]; p->k2_ms += ms3[1]; ++p->launches; p->launch_frames += slot.batched; }
	for (Job *j : slot.jobs) { --p->resident; complete(p, j); }
	slot.jobs.clear(); slot.busy = false;
	p->cv_todo.notify_all();
}

void gpu_main(j40hip_pipeline *p) {
	if (hipSetDevice(p->device) != hipSuccess) { ++p->worker_errors; return; }
	for (;;) {
		std::vector<Job *> take;
		{
			std::unique_lock<std::mutex> lock(p->m);
			auto launchable = [&] {
				if (p->ready.empty()) return false;
				if ((int64_t) p->ready.size() >= p->batch_frames) return true;
				return p->todo.empty() && p->parsing == 0;   // the tail: nothing else is coming
			};
			p->cv_ready.wait(lock, [&] { return p->stop || launchable() || (!p->in_flight.empty() && p->ready.empty()); });
			if (p->stop && p->ready.empty() && p->in_flight.empty()) break;
			if (launchable()) {
				while (!p->ready.empty() && (int64_t) take.size() < p->batch_frames) { take.push_back(p->ready.front()); p->ready.pop_front(); }
			}
		}
		if (take.empty()) {   // nothing to launch: retire the oldest batch in flight
			if (!p->in_flight.empty()) { const int s = p->in_flight.front(); p->in_flight.pop_front(); retire(p, p->slots[(size_t) s]); }
			continue;
		}
		if ((int) p->in_flight.size() >= p->max_in_flight) { const int s = p->in_flight.front(); p->in_flight.pop_front(); retire(p, p->slots[(size_t) s]); }
		int si = -1;
		for (size_t i = 0; i < p->slots.size(); ++i) if (!p->slots[i].busy) { si = (int) i; break; }
		Slot &slot = p->slots[(size_t) si];
		slot.busy = true; slot.jobs = take; slot.batched = 0;
		std::vector<j40hip_frame *> frames; std::vector<void *> outs; std::vector<size_t> strides;
		uint32_t err = 0;
		for (Job *j : take) {
			j->dev_rgba = j->device_output ? j->rgba : acquire_image(p, j->stride * (size_t) j->height);
			if (!j->dev_rgba) err = E_MEM;
			if (!j->single) { frames.push_back(j->frame); outs.push_back(j->dev_rgba); strides.push_back(j->stride); }
		}
		if (!err && !frames.empty()) {
			slot.batched = (int64_t) frames.size();
			if (!slot.batch) slot.batch = j40hip_batch_create(frames.data(), (int64_t) frames.size(), &err);
			else err = j40hip_batch_reset(slot.batch, frames.data(), (int64_t) frames.size());
			if (!err) err = j40hip_batch_decode_recorded(slot.batch, outs.data(), strides.data(), slot.stream, 0);
			for (Job *j : take) if (!err && !j->single) {
				err = j40hip_frame_status_begin(j->frame, slot.stream);
				if (!err && !j->device_output && hipMemcpyAsync(j->rgba, j->dev_rgba, j->stride * (size_t) j->height, hipMemcpyDeviceToHost, slot.stream) != hipSuccess) err = E_GPU;
			}
		}
		for (Job *j : take) if (j->single) j->status = err ? err : decode_single(p, j, slot.stream);
		if (err) for (Job *j : take) if (!j->single) j->status = err;
		if (hipEventRecord(slot.done, slot.stream) != hipSuccess) for (Job *j : take) j->status = E_GPU;
		p->in_flight.push_back(si);
	}
	for (Slot &s : p->slots) if (s.batch) { j40hip_batch_free(s.batch); s.batch = nullptr; }
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
		p->lf_on_device = (flags & 1u) != 0;
		// every frame allocates (and frees) tens of megabytes of tables on its worker thread; as separate mmap()s those serialise all the
		// threads on the process's address-space lock and fault every page in again. Keep such blocks in the heap instead.
		if (!getenv("J40HIP_KEEP_MALLOC_DEFAULTS")) { (void) mallopt(M_MMAP_THRESHOLD, 1 << 30); (void) mallopt(M_TRIM_THRESHOLD, (int) (((size_t) 1 << 31) - 1)); (void) mallopt(M_TOP_PAD, 64 << 20); }
		p->batch_frames = batch_frames < 1 ? 32 : batch_frames;
		p->max_in_flight = max_in_flight < 1 ? 2 : max_in_flight > 8 ? 8 : max_in_flight;
		p->slots.resize((size_t) p->max_in_flight + 1);
		for (Slot &s : p->slots) {
			if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { *err = E_GPU; break; }
		}
		if (!*err) {
			if (host_threads < 1) host_threads = (int) std::max(1u, std::thread::hardware_concurrency());
			if (host_threads > 128) host_threads = 128;   // (each worker owns tens of MB of pinned staging; more than this was never exercised)
			p->gpu = std::thread(gpu_main, p);
			for (int i = 0; i < host_threads; ++i) p->workers.emplace_back(worker_main, p);
		}
	} catch (const std::exception &) { *err = E_MEM; }
	if (*err) { if (p) j40hip_pipeline_free(p); return nullptr; }
	return p;
}

void j40hip_pipeline_free(j40hip_pipeline *p) {
	if (!p) return;
	{ std::unique_lock<std::mutex> lock(p->m); p->stop = true; p->cv_todo.notify_all(); p->cv_ready.notify_all(); }
	for (std::thread &t : p->workers) if (t.joinable()) t.join();
	if (p->gpu.joinable()) p->gpu.join();
	(void) hipSetDevice(p->device);
	for (Job *j : p->todo) delete j;
	for (Job *j : p->ready) { if (j->frame) j40hip_frame_free(j->frame); delete j; }
	for (Slot &s : p->slots) { if (s.done) (void) hipEventDestroy(s.done); if (s.stream) (void) hipStreamDestroy(s.stream); }
	delete p;
}

uint32_t j40hip_pipeline_submit(j40hip_pipeline *p, const void *buf, size_t size, void *rgba, size_t stride_bytes, int device_output, int64_t *ticket) {
	if (!p || !buf || !rgba) return E_RNGE;
	if (p->worker_errors.load()) return E_GPU;
	try {
		Job *j = new Job();
		j->buf = buf; j->size = size; j->rgba = rgba; j->stride = stride_bytes; j->device_output = device_output != 0;
		std::unique_lock<std::mutex> lock(p->m);
		j->ticket = p->submitted++;
		p->results.push_back(0); p->finished.push_back(0);
		if (p->first_submit_ms == 0) p->first_submit_ms = now_ms();
		if (ticket) *ticket = j->ticket;
		p->todo.push_back(j);
		p->cv_todo.notify_one();
	} catch (const std::exception &) { return E_MEM; }
	return 0;
}

uint32_t j40hip_pipeline_drain(j40hip_pipeline *p) {
	if (!p) return E_RNGE;
	std::unique_lock<std::mutex> lock(p->m);
	p->cv_ready.notify_all();
	while (p->completed < p->submitted) {
		if (p->worker_errors.load()) return E_GPU;
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

/* out[0] = parse ms summed over the worker threads, out[1] = plan build + upload ms summed, out[2] = frames completed,
 * out[3] = ms from the first submit to the last completion, out[4] / out[5] = entropy / pixel stage ms summed over the batch launches
 * (HIP events on the launch streams), out[6] = batch launches, out[7] = frames in them */
void j40hip_pipeline_stats(j40hip_pipeline *p, double *out4) {
	if (!p || !out4) return;
	std::unique_lock<std::mutex> lock(p->m);
	out4[0] = p->parse_ms; out4[1] = p->upload_ms; out4[2] = (double) p->completed; out4[3] = p->last_done_ms - p->first_submit_ms;
	out4[4] = p->k1_ms; out4[5] = p->k2_ms; out4[6] = (double) p->launches; out4[7] = (double) p->launch_frames;
}

int64_t j40hip_pipeline_lf_device_frames(j40hip_pipeline *p) { if (!p) return 0; std::unique_lock<std::mutex> lock(p->m); return p->lf_device_frames; }

void j40hip_pipeline_reset_stats(j40hip_pipeline *p) {
	if (!p) return;
	std::unique_lock<std::mutex> lock(p->m);
	p->parse_ms = p->upload_ms = 0; p->first_submit_ms = 0; p->last_done_ms = 0;
	p->k1_ms = p->k2_ms = 0; p->launches = p->launch_frames = 0; p->lf_device_frames = 0;
}

} // extern "C"
