// j40_amd/csrc/device/async.hpp -- the pipeline's asynchronous path for VarDCT frames with several sections (async.hip):
// the host parses what precedes the LfGroup sections and stages it; everything behind that -- LfGroup streams (optionally),
// the plan build, the LfGroup tail, entropy decode, pixels, the verdict -- is enqueued for a whole batch of frames on one
// stream with no host wait in between. Internal to the library (pipeline.hip is the caller).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

struct j40hip_aframe;
struct j40hip_abatch;
struct j40hip_alf;

// Host stage of one frame: parse up to the LfGroup sections, build the front plan, copy it (and the codestream) to the device
// asynchronously on `stream`. lf_on_device != 0: leave the LfGroup streams to k_lf_groups when its tables allow it, else (and
// with 0) decode them here and ship the raw planes. Returns nullptr for frames this path does not take and on ANY error: the
// caller then runs the frame through j40hip_frame_parse / upload / decode, which reports whatever there is to report.
j40hip_aframe *j40hip_aframe_prepare(const void *buf, size_t size, int device, hipStream_t stream, int lf_on_device);
// nothing may still be running on the frame's memory
void j40hip_aframe_free(j40hip_aframe *f);
// grows the device memory cache by `work_copies` working sets and `front_copies` front blocks per frame (at a pipeline's first batch)
void j40hip_aframes_reserve(j40hip_aframe *const *frames, int n, int work_copies, int front_copies);
// frees the calling thread's pinned staging buffers (before a thread that prepared frames exits)
void j40hip_astage_release(void);
int j40hip_aframe_lf_on_device(const j40hip_aframe *f);
// whether everything prepare enqueued for the frame (copy, LfGroup streams) has completed; 8x8 cells of the frame
int j40hip_aframe_uploaded(const j40hip_aframe *f);
int64_t j40hip_aframe_cells(const j40hip_aframe *f);
void j40hip_aframe_size(const j40hip_aframe *f, int64_t *width, int64_t *height);
// what the reference says about bytes behind the frame (j40hip_frame_after_frame_status)
uint32_t j40hip_aframe_after_frame_status(const j40hip_aframe *f);

j40hip_abatch *j40hip_abatch_create(int device);
void j40hip_abatch_free(j40hip_abatch *b);
// Enqueues the whole decode of `n` prepared frames on `stream`; frames[i] writes RGBA u8x4 to rgba_dev[i] with stride_bytes[i].
// Returns 0 or "!gpu" / "!mem". The results are readable once `stream` has been waited for.
uint32_t j40hip_abatch_launch(j40hip_abatch *b, j40hip_aframe *const *frames, int n, void *const *rgba_dev, const size_t *stride_bytes, hipStream_t stream);
// frame i of the last launch: its verdict (0 or the 4-char code of the first failing section in file order) and whether the frame
// has to be decoded again on the single-frame path (an LfGroup section the device decoder cannot take, or an event region that
// overflowed)
void j40hip_abatch_result(const j40hip_abatch *b, int i, uint32_t *code, int *redo);
// ms of the last launch's stages: [0] plan build + LfGroup tail, [1] entropy decode, [2] pixels (HIP events on the batch's stream: a
// stage's time includes what its kernels waited for), [3] k_hf_lanes' own duration as the device recorded it (start of its first
// wavefront to the end of its last: what rocprofv3 reports)
uint32_t j40hip_abatch_elapsed(j40hip_abatch *b, float *ms4);

// The LfGroup streams of `n` prepared frames (those with j40hip_aframe_lf_on_device) in one k_lf_groups launch on `stream`; a frame
// may join a batch once j40hip_alf_done says the launch has completed (the batch's stream also waits for it). One launch per object
// at a time.
j40hip_alf *j40hip_alf_create(int device);
void j40hip_alf_free(j40hip_alf *a);
uint32_t j40hip_alf_launch(j40hip_alf *a, j40hip_aframe *const *frames, int n, hipStream_t stream);
int j40hip_alf_done(j40hip_alf *a);
int j40hip_alf_elapsed(j40hip_alf *a, float *ms, int *frames, int *sections, int *waves);   // the finished launch: the kernel's own duration (device-recorded events); 0 = read

// j40hip_shutdown's share of async.hip: the static-table cache and the event pool
void j40hip_async_shutdown(void);
// 0: every batch slot with its own pixel-kernel streams, all of normal priority; 1: one set per device, slot streams at high priority (async.hip)
int j40hip_stream_layout(void);
