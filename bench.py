#!/usr/bin/env python3
"""bench.py -- Mpixels/s RGBA-u8x4 decode of synthetic 8K VarDCT (d1-like) frames on MI355X.

THE CLOCK. A *step* is one pass of the whole decode path over one batch of `--batch` frames per GPU:
codestream bytes -> [host: container / headers / TOC / LfGlobal / HfGlobal / LfGroup parse, plan build, plan upload]
-> [device: entropy decode of every pass-group section (K1), dequantisation + chroma-from-luma + inverse transforms +
XYB->sRGB + RGBA pack (K2 family)] -> RGBA u8x4 in the j40_pixels_u8x4 layout, resident in HBM. Every stage of many frames
is in flight at once (j40hip_pipeline_*: host worker threads, batched entropy launches on alternating streams). Nothing is
parsed, built or uploaded ahead of the timed region: only the codestream BYTES exist when it starts, and `value` stops when the
RGBA of every frame is back in (pinned) HOST memory -- SURVEY.md section 8d / BASELINE.md section 3: "codestream in host memory ->
RGBA in host memory in the j40_pixels_u8x4 layout", the same two end points as the reference's clock and as `cpu_baseline`
beside it. The copy back is 4 B/px over PCIe (133 MB per 8K frame), which is what bounds `value`. The same pass with the pixels
left in HBM is reported as `device_output` (rounds 2 and 3 reported that as `value`), and the part round 1 reported -- frames
parsed and uploaded ahead of time, kernels only -- as `device_resident`.

The host part runs on the CPU time the container is given (cgroup cpu.max; 16 CPUs on the bench boxes although 256 are visible): four
worker threads at half a millisecond a frame. What bounds `value` is the PCIe link: 133 MB of RGBA per frame, copied back on the SDMA
engine the library measured as the device's fastest (j40_amd/csrc/device/hostcopy.hip); DESIGN.md section 5 has the breakdown.

Also on the line: `roofline` (the stage of a batch furthest below the HBM roofline -- the LfGroup launch, k_lf_rows + k_lf_predict -- with the four stages'
own durations inside the timed region and alone, device-recorded); `latency_mode` (one frame alone); BASELINE.json's other configurations (`configs`); `cpu_baseline` = the
unmodified reference on one host core, same stream; `parity_vs_reference` for a frame decoded inside the timed region.

N > 1 (torch.distributed.run, one rank per GPU): frames are independent, every rank runs its own pipeline on its own frames with
its share of the host CPUs (weak scaling, no data-path collective); `value` is the whole-job aggregate. `--shard-groups` is the
north star's single-frame sharding (j40_amd/sharding.py). Prints ONE JSON line on rank 0.
"""
import argparse
import concurrent.futures
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "Mpixels/s RGBA-u8x4 decode, 8K VarDCT d1"


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), else the visible CPU count"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, -(-int(q) // int(p)))   # (rounded up, like j40hip_cpu_quota)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, -(-q // p))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def pin_to_gpu_numa_node(torch, local_rank):
    """binds this rank (and the threads it starts: pipeline workers, the launching thread, the HIP runtime's) to the CPUs of its GPU's
    NUMA node, so that staging buffers and the pinned landing buffers of the pixels are node-local and N ranks' copies back do not
    all cross the sockets' link. Returns what was done for the bench line; J40_BENCH_NUMA=0 leaves the affinity alone."""
    if os.environ.get("J40_BENCH_NUMA") == "0":
        return {"pinned": False, "why": "J40_BENCH_NUMA=0"}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return {"pinned": False, "why": "the device reports no NUMA node", "pci": bdf}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pinned": False, "why": "none of node %d's CPUs is allowed to this process" % node, "pci": bdf}
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "node": node, "cpus": len(cpus), "pci": bdf}
    except (OSError, ValueError, AttributeError) as e:
        return {"pinned": False, "why": "%s: %s" % (type(e).__name__, e)}


def cgroup_cpu_stat():
    """the container's CPU accounting (cgroup v2: cpu.stat; v1: cpu/cpu.stat + cpuacct/cpuacct.usage), or None: what the process group
    used and how often the kernel throttled it -- the copies back are completed by host threads, and a throttled process's copies crawl"""
    def pairs(path):
        out = {}
        for line in open(path):
            k, _, v = line.partition(" ")
            out[k] = int(v)
        return out
    try:
        return pairs("/sys/fs/cgroup/cpu.stat")
    except (OSError, ValueError):
        pass
    try:
        out = pairs("/sys/fs/cgroup/cpu/cpu.stat")
        out["usage_usec"] = int(open("/sys/fs/cgroup/cpuacct/cpuacct.usage").read()) // 1000
        out["throttled_usec"] = out.get("throttled_time", 0) // 1000
        return out
    except (OSError, ValueError):
        return None


def cpu_baseline(data, width, height, budget_s=12.0):
    """the unmodified reference (oracle/_ref) on ONE host core, same codestream, bounded sample"""
    from refdec import Ref, REF_SO
    import numpy as np
    if not os.path.exists(REF_SO):
        return None
    ref = Ref()
    out = np.zeros(width * height * 4, np.uint8)
    buf = C.create_string_buffer(data, len(data))
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 and (not times or time.perf_counter() - t_start + times[-1] < budget_s * 2):
        t0 = time.perf_counter()
        err = ref.lib.ref_decode_into(buf, len(data), out.ctypes.data, out.size)
        times.append(time.perf_counter() - t0)
        if err:
            return None
    best = min(times)
    cpu_baseline.last_pixels = out.reshape(height, width, 4)
    return {"value": round(width * height / best / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
            "sample": "%d full decodes of the same %dx%d stream through the reference's public API, codestream bytes in memory -> RGBA in memory (best of %d, %.2f s each), 1 host core"
                      % (len(times), width, height, len(times), best)}


def _cpu_worker(job):
    """one worker PROCESS of cpu_baseline_many: decodes its share of the streams `rounds` times through the reference (no GPU, no torch)"""
    paths, width, height, rounds = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from refdec import Ref
    import numpy as np
    ref = Ref()
    out = np.zeros(width * height * 4, np.uint8)
    bufs = []
    for p in paths:
        d = open(p, "rb").read()
        bufs.append((C.create_string_buffer(d, len(d)), len(d)))
    if rounds == 0:
        return 0
    n = 0
    for _ in range(rounds):
        for b, size in bufs:
            if ref.lib.ref_decode_into(b, size, out.ctypes.data, out.size):
                return -1
            n += 1
    return n


def cpu_baseline_many(datas, width, height, procs, budget_s=10.0):
    """SURVEY 8d / BASELINE.md 3 for the batch configuration: the unmodified reference in N = #cores worker PROCESSES, each decoding
    its own share of the frames on one core (the reference is single-threaded by design, j40.h:8034); bounded sample"""
    import multiprocessing as mp
    import tempfile
    from refdec import REF_SO
    if not os.path.exists(REF_SO):
        return None
    procs = max(1, procs)
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, d in enumerate(datas):
            paths.append(os.path.join(td, "%d.jxl" % i))
            open(paths[-1], "wb").write(d)
        per = max(1, len(paths) // procs) if len(paths) >= procs else 1
        shares = [[paths[(k * per + j) % len(paths)] for j in range(per)] for k in range(procs)]
        ctx = mp.get_context("spawn")   # (no fork: this process has the HIP runtime loaded)
        with ctx.Pool(procs) as pool:
            pool.map(_cpu_worker, [(sh, width, height, 0) for sh in shares])    # workers up, library loaded, files read
            t0 = time.perf_counter()
            one = pool.map(_cpu_worker, [(sh, width, height, 1) for sh in shares])
            t_one = time.perf_counter() - t0
            if min(one) < 0:
                return None
            rounds = max(1, min(50, int(budget_s / max(t_one, 1e-3))))
            t0 = time.perf_counter()
            done = pool.map(_cpu_worker, [(sh, width, height, rounds) for sh in shares])
            el = time.perf_counter() - t0
    if min(done) < 0:
        return None
    frames = sum(done)
    return {"value": round(width * height * frames / el / 1e6, 2), "unit": "Mpixels/s", "cores": procs, "kind": "reference",
            "sample": "%d worker processes (one per CPU of the container's quota), each decoding its own %d streams %d times through the reference's public API: %d frames of %dx%d in %.2f s"
                      % (procs, per, rounds, frames, width, height, el)}


def _lf_kernel_name():
    if os.environ.get("J40HIP_LF_KERNEL") == "lanes":
        return "k_lf_lanes"
    return "k_lf_rows" if os.environ.get("J40HIP_LF_RAW") == "0" else "k_lf_rows + k_lf_predict (leaf-only channels left as residuals, predicted afterwards)"


def synth_many(specs, workers):
    """generates (or finds in build/streams) the listed streams, `workers` generator processes at a time"""
    from streams import synth
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        return list(ex.map(lambda s: synth(s[0], s[1], s[2], s[3], **s[4]), specs))


def run_pipeline_steps(pipe, bufs, sizes, outs, stride, device_output, steps, torch, dev, dist):
    """`steps` timed steps: every frame of the batch submitted, then drained; returns (elapsed seconds, tickets of the last step)"""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    pipe.reset_stats()
    t0 = time.perf_counter()
    tickets = []
    # the steps are queued back to back and drained once: the pipeline keeps the host threads parsing step k + 1 while the device still
    # decodes the last batch of step k (a step's frames overwrite the previous step's pixels: same streams, same pixels)
    for _ in range(steps):
        tickets = [pipe.submit_raw(bufs[i], sizes[i], outs[i].data_ptr(), stride, device_output) for i in range(len(bufs))]
    pipe.drain()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    run_pipeline_steps.per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)   # (each rank's own clock, for the line's per-rank PCIe figures)
        run_pipeline_steps.per_rank = [float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, tickets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct streams a batch cycles through")
    ap.add_argument("--pipe-batch", type=int, default=256, help="frames per entropy launch inside the pipeline")
    ap.add_argument("--host-threads", type=int, default=0, help="pipeline worker threads per GPU (default: the container's CPU quota / GPUs)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches the pipeline keeps in flight on the device")
    ap.add_argument("--device-output-steps", type=int, default=12, help="steps of the `device_output` section (pixels left in HBM)")
    ap.add_argument("--device-output-batch", type=int, default=256, help="frames per step and per entropy launch of the `device_output` section")
    ap.add_argument("--queued-batch", type=int, default=512, help="frames per entropy launch of the `k_hf_lanes_queued` section (0: skip it)")
    ap.add_argument("--device-output-lf", choices=["auto", "device", "host"], default="device")
    ap.add_argument("--host-buffers", type=int, default=0, help="pinned landing buffers for the pixels (default: one per distinct stream, at most 64; 24 per rank with several ranks)")
    ap.add_argument("--lf-streams", choices=["auto", "device", "host"], default="device",
                    help="who decodes the LfGroup streams of the batched frames: the GPU (k_lf_lanes, a lane per section), the host worker threads, or decided frame by frame (auto: the GPU up to its stage's capacity, the host threads beyond)")
    ap.add_argument("--resident-batch", type=int, default=256, help="frames of the device-resident section (kernels only, as round 1 measured)")
    ap.add_argument("--stream", choices=["forward", "coefficient"], default="forward",
                    help="forward: the generator ENCODES a procedural picture at about distance 1 (tools/jxlsynth forward=1); "
                         "coefficient: the coefficient-domain synthetic stream rounds 1 and 2 were tuned on")
    ap.add_argument("--maxlog", type=int, default=0, help="--stream coefficient: largest transform side (log2) in the mix; 8 brings in the 128/256-sized transforms (k_vardct_large)")
    ap.add_argument("--shard-groups", action="store_true", help="single-frame mode: ONE frame per step, its pass groups split over the ranks (j40_amd.sharding)")
    ap.add_argument("--shard-kind", choices=["vardct", "modular"], default="vardct", help="--shard-groups: a VarDCT frame, or a Modular lossless frame (RCT only; e.g. --width 16384 --height 16384 = BASELINE config 4)")
    ap.add_argument("--config5-batch", type=int, default=512, help="config 5 (1024 x 1920x1080): frames per entropy launch")
    ap.add_argument("--config5-in-flight", type=int, default=2)
    ap.add_argument("--config5-lf", choices=["auto", "device", "host"], default="host")
    ap.add_argument("--sharded-record", action="store_true", help="add the `sharded` record (one frame split by group ranges) also on one rank; with several ranks it is always there")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-sections", action="store_true", help="only the timed pipeline (no device-resident / latency / other-config sections)")
    ap.add_argument("--skip-modular", action="store_true", help="leave BASELINE config 4 (16384 x 16384 Modular) out of `configs`")
    args = ap.parse_args()

    import torch
    import numpy as np
    import j40_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        import __graft_entry__
        if not os.path.exists(j40_amd.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "build", "jxlsynth")):
            __graft_entry__.build()
    if not torch.cuda.is_available() or j40_amd.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # (J40_BENCH_BACKEND=gloo J40_BENCH_SHARE_DEVICE=1: a dry run of the multi-rank code path on a box with fewer GPUs than ranks;
    #  the ranks share devices and the collectives go over gloo with host tensors. Not a measurement.)
    backend = os.environ.get("J40_BENCH_BACKEND", "nccl")
    if os.environ.get("J40_BENCH_SHARE_DEVICE") == "1":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)   # before the first collective: RCCL binds a rank to its current device
    dev = torch.device("cuda", local_rank)
    numa = pin_to_gpu_numa_node(torch, local_rank) if world > 1 or os.environ.get("J40_BENCH_NUMA") == "1" else {"pinned": False, "why": "one rank"}
    dist = None
    if world > 1 or os.environ.get("J40_BENCH_FORCE_DIST") == "1":   # (J40_BENCH_FORCE_DIST: the process group also for one rank -- a dry run of the RCCL calls on a one-GPU box)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        dist.barrier()
    if args.shard_groups:
        from streams import synth
        if args.shard_kind == "modular":
            return bench_sharded(args, torch, j40_amd, dist, dev, rank, local_rank, world, synth("modular", args.width, args.height, args.seed, tree=1, repeat=1 if args.width * args.height <= (1 << 24) else 16))
        return bench_sharded(args, torch, j40_amd, dist, dev, rank, local_rank, world, synth("vardct", args.width, args.height, args.seed, **({"forward": 1} if args.stream == "forward" else {})))

    W, H, B = args.width, args.height, args.batch
    quota = cpu_quota()
    # Worker threads of the pipelines whose LfGroup streams the GPU decodes (the timed region, `device_output`): FOUR, not the quota.
    # A frame's host stage is 1-2 ms, so two threads already feed the 430 frames/s the PCIe link takes and four the 1 450 the
    # device takes -- and sixteen of them, waking together after every batch beside the launching thread and the HIP runtime's own
    # threads, run the container into its CPU quota: the cgroup then throttles every thread of the process (cpu.stat: 221 of 1 282
    # periods throttled in a long run), and the copies back, whose completion the runtime handles on the host, crawl -- 6.6 Gpx/s
    # with 16 threads against 12.5 with 4 and 13.4 with 2 on the same box, same run (gpurun_out/r04n).
    # (several ranks share the quota: quota / world each, at least one -- 8 ranks on a 16-CPU quota get two)
    threads = args.host_threads or (max(2, min(4, quota)) if world == 1 else max(1, min(4, quota // world)))
    threads_host_lf = args.host_threads or (max(2, quota) if world == 1 else max(1, quota // world))   # (pipelines whose worker threads decode the LfGroup streams themselves)
    D = max(1, min(args.distinct, B))
    # every rank decodes its own batch of the same D streams: rank 0 generates them with all the CPUs the container has (an 8K
    # encode takes ~10 s of one core), the other ranks wait and read them from build/streams
    stream_opts = {"forward": 1} if args.stream == "forward" else ({"maxlog": args.maxlog} if args.maxlog else {})
    specs = [("vardct", W, H, args.seed + 1000 * i, stream_opts) for i in range(D)]
    if rank == 0:
        synth_many(specs, max(1, quota))
    if dist is not None:
        dist.barrier()
    datas = synth_many(specs, max(1, quota // world))
    bufs = [C.create_string_buffer(d, len(d)) for d in datas]
    step_bufs = [bufs[i % D] for i in range(B)]
    step_sizes = [len(datas[i % D]) for i in range(B)]
    pipe = j40_amd.Pipeline(local_rank, threads, min(args.pipe_batch, B), args.in_flight, lf_streams=args.lf_streams)
    # the pixels land in pinned host memory, used round-robin: frame i and frame i + D are the same stream, hence the same pixels
    # (letting the pixel kernels store straight into pinned memory instead was measured: 5.1 Gpx/s against 7.5 for the copy)
    nh = args.host_buffers or min(D, B, 64 if world == 1 else 24)
    host_outs = [torch.empty((H, W, 4), dtype=torch.uint8).pin_memory() for _ in range(nh)]
    step_outs = [host_outs[i % nh] for i in range(B)]
    # the link as this process finds it, before anything else runs: one device image copied into every landing buffer in turn (the
    # copy the pipeline issues per frame: hipMemcpyAsync of 133 MB, device -> pinned host), each timed by itself on an idle device
    link_probe = None
    try:
        src = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
        rates = []
        for rep in range(2):
            for hbuf in host_outs:
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                hbuf.copy_(src, non_blocking=True)
                torch.cuda.synchronize(dev)
                if rep:
                    rates.append(hbuf.numel() / (time.perf_counter() - t0) / 1e9)
        rates.sort()
        link_probe = {"d2h_gb_per_s_per_landing_buffer": {"min": round(rates[0], 2), "median": round(rates[len(rates) // 2], 2), "max": round(rates[-1], 2), "buffers": len(rates)},
                      "how": "before the warm-up, idle device: one 133 MB device image copied into each pinned landing buffer by itself (second pass timed)"}
        del src
    except RuntimeError as e:
        link_probe = {"error": str(e)[:200]}

    # the W untimed steps, queued back to back like the timed ones: the pipeline reaches the depth it has in the timed region (frames
    # parsed ahead, three batches' working sets, the device memory cache grown to hold them) before the clock starts -- W steps drained
    # one by one never get there, and a fresh box then spent 2 s of its first timed steps in hipMalloc (call D, run 1: 12.7 Gpx/s
    # against 13.4-13.5 for the five runs after it)
    if args.warmup > 0:
        run_pipeline_steps(pipe, step_bufs, step_sizes, step_outs, W * 4, False, args.warmup, torch, dev, None)
    cg0 = cgroup_cpu_stat()
    elapsed, tickets = run_pipeline_steps(pipe, step_bufs, step_sizes, step_outs, W * 4, False, args.steps, torch, dev, dist)
    per_rank_elapsed = list(run_pipeline_steps.per_rank)
    cg1 = cgroup_cpu_stat()
    st = pipe.stats()
    for t in tickets:
        assert pipe.result(t) == "", "decode error: " + pipe.result(t)
    first_pixels = host_outs[0].clone()
    # ---- the same path with the pixels left in HBM (what rounds 2 and 3 reported as `value`): the device's own pace, its own pipeline,
    # the LfGroup streams on the GPU
    device_output = None
    if not args.skip_sections or world > 1:
        pipe.close()
        torch.cuda.empty_cache()
        Bd = max(1, args.device_output_batch)
        nd = min(Bd, 256)   # device images, shared by frames i and i + 256 k: the same stream (64 distinct ones), hence the same pixels
        outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(nd)]
        dbufs = [bufs[i % D] for i in range(Bd)]; dsizes = [len(datas[i % D]) for i in range(Bd)]; douts = [outs[i % nd] for i in range(Bd)]
        dpipe = j40_amd.Pipeline(local_rank, threads, Bd, args.in_flight, lf_streams=args.device_output_lf)
        # (two untimed steps: a pipeline sizes the device memory cache for its full depth at its SECOND full batch -- 0.9 s of
        # allocations that one warm-up step left inside the timed steps whenever the cache did not already hold the blocks:
        # 258 ms per step instead of 166 in one of round 5's two evidence runs, 318 against 145-149 in four probe runs in a row)
        run_pipeline_steps(dpipe, dbufs, dsizes, douts, W * 4, True, 2, torch, dev, None)
        dsteps = max(1, args.device_output_steps)
        e_dev, tk = run_pipeline_steps(dpipe, dbufs, dsizes, douts, W * 4, True, dsteps, torch, dev, dist)
        sd = dpipe.stats()
        assert all(dpipe.result(t) == "" for t in tk)
        assert torch.equal(outs[0].cpu(), first_pixels)
        # the same pipeline over a third of the steps: a region starts with an empty pipeline (barrier + synchronise on both sides), so
        # its first batch waits for the first LfGroup launch, the plan build and the entropy decode with nothing beside them; what a
        # step costs once the stages overlap is the DIFFERENCE of the two regions over the difference of their step counts
        steady = None
        if dsteps >= 6:
            ssteps = dsteps // 3
            e_short, tk2 = run_pipeline_steps(dpipe, dbufs, dsizes, douts, W * 4, True, ssteps, torch, dev, dist)
            assert all(dpipe.result(t) == "" for t in tk2)
            steady = {"ms_per_step": round((e_dev - e_short) / (dsteps - ssteps) * 1e3, 3), "fill_ms": round((e_short - ssteps * (e_dev - e_short) / (dsteps - ssteps)) * 1e3, 3),
                      "how": "(%d steps: %.1f ms) - (%d steps: %.1f ms) over %d steps; fill_ms = what a region costs beyond its steps at that pace (the empty pipeline's head and tail)" % (dsteps, e_dev * 1e3, ssteps, e_short * 1e3, dsteps - ssteps)}
        dpipe.close()
        dl = max(sd["launches"], 1)
        k1d = (sd["k1_kernel_ms"] / dl) or (sd["k1_ms"] / dl)
        fpl = sd["launch_frames"] / dl
        alg_d = sum(4 * W * H + s for s in dsizes) / Bd   # per frame
        device_output = {"value": round(W * H * Bd * dsteps * world / e_dev / 1e6, 2), "unit": "Mpixels/s", "frames_per_step": Bd, "steps": dsteps,
                         "ms_per_step": round(e_dev / dsteps * 1e3, 3), "ms_per_256_frames": round(e_dev / dsteps * 1e3 * 256 / Bd, 3), "frames_per_launch": round(fpl, 1), "lf_streams": args.device_output_lf,
                         "k_hf_lanes_ms_per_launch": round(k1d, 3), "k_hf_lanes_roofline_frac": round(alg_d * fpl / (k1d / 1e3) / 8e12, 6) if k1d > 0 else None,
                         "pixel_kernels_ms_per_launch": round(sd["k2_ms"] / dl, 3), "lf_streams_plan_tail_ms_per_launch": round(sd["lf_plan_ms"] / dl, 3),
                         "roofline_frac_step": round(alg_d * Bd / (e_dev / dsteps) / 8e12, 6),
                         "steady": steady,
                         "note": "as `value`, but the RGBA stays in device memory (no copy back): the device is the bound here, PCIe is for `value`; ms_per_step includes filling and draining the pipeline once per region, `steady` takes that out"}
        # ---- k_hf_lanes' queued form: launches with more sections than the machine has lanes (512 8K frames = 261 120 sections
        # against 131 072 lanes at the two wavefronts per SIMD the tables' LDS allows): every frame's lanes take its sections from a
        # shared counter, largest first. One batch in flight, so that nothing runs beside the kernel.
        if args.queued_batch > 0 and world == 1:
            Bq = args.queued_batch
            qbufs = [bufs[i % D] for i in range(Bq)]; qsizes = [len(datas[i % D]) for i in range(Bq)]; qouts = [outs[i % nd] for i in range(Bq)]
            qpipe = j40_amd.Pipeline(local_rank, threads_host_lf, Bq, 1, lf_streams="host")
            run_pipeline_steps(qpipe, qbufs, qsizes, qouts, W * 4, True, 1, torch, dev, None)
            eq, tk = run_pipeline_steps(qpipe, qbufs, qsizes, qouts, W * 4, True, 2, torch, dev, None)
            sq = qpipe.stats()
            assert all(qpipe.result(t) == "" for t in tk)
            assert torch.equal(outs[0].cpu(), first_pixels)
            qpipe.close()
            ql = max(sq["launches"], 1)
            if sq["k1_kernel_ms"] > 0 and abs(sq["launch_frames"] / ql - Bq) < 1:
                kq = sq["k1_kernel_ms"] / ql
                device_output["k_hf_lanes_queued"] = {"frames_per_launch": Bq, "kernel_ms": round(kq, 3), "ms_per_256_frames": round(kq * 256 / Bq, 3),
                                                      "achieved": round(alg_d * Bq / (kq / 1e3) / 1e9, 3), "frac": round(alg_d * Bq / (kq / 1e3) / 8e12, 6), "launches": sq["launches"],
                                                      "how": "one batch of %d frames in flight, LfGroup streams on the host threads: k_hf_lanes alone on the device (device-recorded start/end events); "
                                                             "the same launch with one section per lane runs in two rounds (J40HIP_K1_QUEUE_WAVES=0: 103 ms, DESIGN.md section 4)" % Bq}
            del qouts
        del outs, douts
        torch.cuda.empty_cache()
    resident_multi = None
    if world > 1 and not args.skip_sections:
        # every rank's kernels alone on frames prepared ahead (what scales with the GPUs when the host's CPU quota does not):
        # the same figure the single-GPU line carries as `device_resident`, aggregated over the ranks
        pipe.close()
        torch.cuda.empty_cache()
        R = max(1, min(args.resident_batch, B))
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
            frames = list(ex.map(lambda i: j40_amd.Frame(datas[i % len(datas)], threads=1), range(R)))
        for fr in frames:
            fr.upload(local_rank)
        routs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(R)]
        batch = j40_amd.Batch(frames)
        ptrs, strides, main = [o.data_ptr() for o in routs], [W * 4] * R, torch.cuda.current_stream(dev)
        batch.decode_recorded(ptrs, strides, main.cuda_stream, 0)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for k in range(5):
            batch.decode_recorded(ptrs, strides, main.cuda_stream, k)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert all(fr.status() == "" for fr in frames)
        resident_multi = {"value": round(W * H * R * 5 * world / float(t.item()) / 1e6, 2), "unit": "Mpixels/s", "frames_per_step_per_gpu": R, "steps": 5,
                          "note": "kernels only on frames parsed, planned and uploaded ahead, all ranks at once (max over ranks, barriers on both sides)"}
        del routs, frames, batch
    sharded = None
    if (world > 1 or args.sharded_record or dist is not None) and not args.skip_sections:
        if resident_multi is None:
            pipe.close()
            torch.cuda.empty_cache()
        sharded = sharded_records(args, torch, j40_amd, dist, dev, rank, local_rank, world)
    if rank != 0:
        if resident_multi is None:
            pipe.close()
        return

    frames_total = B * args.steps * world
    value = W * H * frames_total / elapsed / 1e6
    alg_step = sum(4 * W * H + s for s in step_sizes)            # algorithmic bytes of one step on one GPU: RGBA written + codestream read
    launches = max(st["launches"], 1)
    k1_stage_ms = st["k1_ms"] / launches                       # HIP events on the batch's stream around the entropy stage (includes queueing behind other kernels)
    k1_launch_ms = (st["k1_kernel_ms"] / launches) or k1_stage_ms   # k_hf_lanes itself: events the device records at the kernel's start and end (hipExtLaunchKernelGGL)
    frames_per_launch = st["launch_frames"] / launches
    alg_launch = alg_step * frames_per_launch / B
    achieved = alg_launch / (k1_launch_ms / 1e3) / 1e9 if k1_launch_ms > 0 else 0.0
    stream_words = ("distance-1 encodes of procedural pictures (tools/jxlsynth forward=1: forward transforms, library matrices" if args.stream == "forward"
                    else "d1-like synthetic frames (tools/jxlsynth, coefficient-domain synthesis")
    result = {
        "metric": METRIC, "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("%d x %dx%d VarDCT frames per GPU per step, " + stream_words + ", %d distinct streams, %.3f bpp, %d pass groups each), whole decode "
                               "path per frame inside the timed region: the host parses what precedes the LfGroup sections (%d worker threads, container CPU quota %d of %d visible CPUs) and copies "
                               "it to the GPU; LfGroup streams (%s), plan build, LfGroup tail, entropy decode and pixel kernels are enqueued per batch of %d frames; "
                               "the RGBA of each frame is copied back into pinned host memory behind its batch's kernels; codestream bytes in host memory in, RGBA u8x4 in host memory out")
                               % (B, W, H, D, 8.0 * sum(step_sizes) / (B * W * H), ((W + 255) // 256) * ((H + 255) // 256), threads, quota, os.cpu_count() or 1,
                                  {"auto": "GPU or host thread, decided per frame", "device": "GPU", "host": "host threads"}[args.lf_streams], min(args.pipe_batch, B)),
                   "clock": "codestream bytes in host memory -> RGBA u8x4 in (pinned) host memory, nothing prepared ahead (SURVEY 8d / BASELINE.md 3: the same two end points as cpu_baseline); bound by the copy back, 4 B/px over PCIe; the same pass with the pixels left in HBM: `device_output`",
                   "stream": args.stream, "frame_pixels": W * H, "frames_per_step": B, "codestream_bytes": step_sizes[0], "parallelism": "frames x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": None,
                     "kernel": "k_hf_lanes", "kernel_ms": round(k1_launch_ms, 4), "launches_in_timed_region": st["launches"], "frames_per_launch": round(frames_per_launch, 2),
                     "algorithmic_bytes_per_launch": int(alg_launch),
                     "step_frac": round(alg_step * args.steps / elapsed / 8e12, 6),
                     "note": "kernel_ms: k_hf_lanes' own duration inside the timed region, from HIP events the device records at the kernel's start and end (hipExtLaunchKernelGGL; what rocprofv3 --kernel-trace reports), averaged over the launches; other batches' stages run beside it (`kernel_alone`: the same launch with the device to itself); step_frac = algorithmic bytes of the steps / wall time / peak -- the wall time of `value` is PCIe time, see device_output for the device's own pace"},
        "pipeline": {"cgroup_cpu_in_region": None if not (cg0 and cg1) else {"cpus_used_on_average": round((cg1["usage_usec"] - cg0["usage_usec"]) / 1e6 / max(elapsed, 1e-9), 2),
                                                                                   "periods": cg1.get("nr_periods", 0) - cg0.get("nr_periods", 0), "periods_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                                                                   "throttled_s": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e6, 3)},
                     "host_stage_ms_per_frame": round(st["parse_thread_ms"] / max(st["completed"] - st["single_frames"], 1), 2),
                     "lf_streams_plan_tail_ms_per_launch": round(st["lf_plan_ms"] / launches, 3), "entropy_ms_per_launch": round(k1_stage_ms, 3), "pixel_kernels_ms_per_launch": round(st["k2_ms"] / launches, 3),
                     "host_threads": threads, "cpu_quota": quota, "cpu_quota_per_rank": round(quota / world, 2), "numa": numa, "lf_streams": args.lf_streams, "lf_streams_on_device_frames": st["lf_device_frames"], "frames": st["completed"],
                     "single_frame_path_frames": st["single_frames"],
                     "note": "host_stage: ms of one worker thread per frame (headers, TOC, LfGlobal, HfGlobal, staging; plus the LfGroup streams for the frames the host kept); "
                             "the per-launch figures are HIP-event times on the batch's stream and overlap with other batches' stages"},
    }
    if device_output is not None:
        result["device_output"] = device_output
    result["pcie"] = {"bytes_back_per_step": 4 * W * H * B, "achieved_gb_per_s": round(4 * W * H * frames_total / world / elapsed / 1e9, 2),
                      "note": "RGBA copied back per GPU during the timed region / wall time; a Gen5 x16 link moved 57 GB/s device-to-host on these boxes (tools/pcie_probe.py): 14.2 Gpx/s is the ceiling of `value` per GPU",
                      "per_rank_gb_per_s": [round(4 * W * H * B * args.steps / e / 1e9, 2) for e in per_rank_elapsed],
                      "host_memory_landing_gb_per_s": round(sum(4 * W * H * B * args.steps / e / 1e9 for e in per_rank_elapsed), 2),
                      "per_rank_note": "each rank's own copy rate over its own clock (barrier to barrier); their sum is what the host's memory takes in -- N ranks land N x 53 GB/s of RGBA in one host's DRAM, beside N x 1.1 GB/s of uploads",
                      "link_probe": link_probe,
                      "copy_engine": j40_amd.copy_engine(local_rank),
                      "copy_engine_note": "the copies back are issued on ONE SDMA engine the library measured as the fastest of the device's sixteen (hsa_amd_memory_async_copy_on_engine; j40_amd/csrc/device/hostcopy.hip): hipMemcpyAsync lets the runtime take whichever engine is free, and they range from 57 to 7 GB/s device to host -- that was rounds 4-5's 'slow runs'"}
    if link_probe and "d2h_gb_per_s_per_landing_buffer" in link_probe:
        med = link_probe["d2h_gb_per_s_per_landing_buffer"]["median"]
        result["pcie"]["slow_run"] = bool(result["pcie"]["achieved_gb_per_s"] < 0.85 * med)
        result["pcie"]["slow_run_rule"] = "the region's copy rate below 0.85 x the idle link's (median of the probe)"
    if resident_multi is not None:
        result["device_resident"] = resident_multi
    if sharded is not None:
        result["sharded"] = sharded
    # ---- the stages of a batch, each with its own duration inside the timed region and its roofline fraction (algorithmic bytes of
    # the frames the launch carries / its duration / 8 TB/s); `kernel` = the stage furthest below the roofline, i.e. the longest per frame
    alg_frame = alg_step / B
    def stage(name, kernel, ms, frames, how):
        if not ms or ms <= 0 or frames <= 0:
            return None
        ach = alg_frame * frames / (ms / 1e3) / 1e9
        return {"stage": name, "kernel": kernel, "kernel_ms": round(ms, 4), "frames_per_launch": round(frames, 2), "ms_per_256_frames": round(ms * 256 / frames, 3),
                "achieved": round(ach, 3), "frac": round(ach / 8000.0, 6), "traffic": None, "how": how}
    lfl = max(st.get("lf_launches", 0), 1)
    stages = [
        stage("LfGroup streams (j40.h:6722-6790)", _lf_kernel_name(), st.get("lf_kernel_ms", 0) / lfl, st.get("lf_launch_frames", 0) / lfl,
              "device-recorded start / end events of each launch (hipExtLaunchKernelGGL), averaged over %d launches; a launch carries what was waiting, up to four batches' worth, on a low-priority stream beside the other stages, and lasts as long as its longest section whatever it carries (a section is one lane): ms_per_256_frames is what its frames' share of it comes to, not a cost per frame" % st.get("lf_launches", 0)),
        stage("plan build + LfGroup tail (j40.h:6585-6720, 6544-6590, 5944)", "k_plan_place / _scan / _emit, k_lf_dequant_smooth_batch, k_llf_small_batch, k_llf_large_batch", st["lf_plan_ms"] / launches, frames_per_launch,
              "HIP events on the batch's stream around the stage (includes what the stage waited for behind other kernels)"),
        stage("entropy decode (j40.h:6888-7005)", "k_hf_lanes", k1_launch_ms, frames_per_launch, "device-recorded start / end events of the kernel"),
        stage("pixels (j40.h:7053-7247, 5690-6246)", "k_vardct_dct<...> x 12, k_vardct_special_{123,halves,afv}, k_vardct_large", st["k2_ms"] / launches, frames_per_launch,
              "HIP events around the stage: four chains of persistent launches on four streams, fork to join"),
    ]
    stages = [x for x in stages if x]
    result["roofline"]["stages"] = stages
    try:
        import glob
        pt = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic.json")))[-1]))   # (the latest round's counter passes: tools/r06_final.sh)
        if (W, H) == (7680, 4320) and pt.get("stream", "coefficient") == args.stream:
            corr = pt.get("fetch_correction", 1.0)
            for x in stages:
                for key, rec in pt.get("stages", {}).items():
                    if x["stage"].startswith(key) and rec.get("frames_per_launch", 0) > 0:   # (scaled to this run's frames per launch: the LfGroup launches carry what was waiting)
                        x["traffic"] = int((rec["fetch_kb"] * corr + rec["write_kb"]) * 1024 * x["frames_per_launch"] / rec["frames_per_launch"])
                        x["traffic_fetch_kb_uncorrected"] = rec["fetch_kb"]; x["traffic_write_kb"] = rec["write_kb"]
            result["roofline"]["traffic_source"] = pt["source"]
    except (OSError, ValueError, KeyError):
        pass
    if stages:
        worst = min(stages, key=lambda x: x["frac"])
        result["roofline"].update({"kernel": worst["kernel"], "stage": worst["stage"], "kernel_ms": worst["kernel_ms"], "frames_per_launch": worst["frames_per_launch"],
                                   "achieved": worst["achieved"], "frac": worst["frac"], "traffic": worst["traffic"], "algorithmic_bytes_per_launch": int(alg_frame * worst["frames_per_launch"])})
        result["roofline"]["note"] = ("`kernel` is the stage of a batch that is furthest below the HBM roofline inside the timed region (the longest per frame): its own duration from events the device "
                                      "records, its algorithmic bytes = (4 B/px RGBA + codestream bytes) of the frames its launch carries; `stages` lists all four with the same arithmetic and, per stage, the PMC traffic "
                                      "(FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE) from the tracked rocprofv3 passes; `stages_alone`: the same launches with the device to themselves; "
                                      "step_frac = algorithmic bytes of the steps / wall time / peak -- the wall time of `value` is PCIe time, see device_output for the device's own pace")

    # parity at the full size: a frame decoded inside the timed region against the reference's pixels (bar: 1 level)
    if not args.no_cpu_baseline and world == 1:
        cb = cpu_baseline(datas[0], W, H)
        if cb:
            result["cpu_baseline"] = cb
            d = np.abs(first_pixels.numpy().astype(np.int16) - cpu_baseline.last_pixels.astype(np.int16))
            result["parity_vs_reference"] = {"max_abs_diff": int(d.max()), "differing_samples": int((d > 0).sum()), "samples": int(d.size)}
            assert d.max() <= 1, "GPU and reference pixels differ by more than one level"

    if not args.skip_sections and world == 1:
        del host_outs, step_outs
        pipe.close()
        torch.cuda.empty_cache()
        # ---- k_hf_lanes with the device to itself: one batch in flight, the LfGroup streams decoded by the host threads (nothing else
        # runs beside the kernel). The timed region above overlaps it with the previous batch's pixel kernels and the lane decoder of
        # the LfGroup streams, which is what makes `value` -- and stretches the kernel. Same frames, same launch geometry.
        outs2 = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(B)]
        alone = j40_amd.Pipeline(local_rank, threads_host_lf, min(args.pipe_batch, B), 1, lf_streams="host")
        run_pipeline_steps(alone, step_bufs, step_sizes, outs2, W * 4, True, 1, torch, dev, None)
        run_pipeline_steps(alone, step_bufs, step_sizes, outs2, W * 4, True, 2, torch, dev, None)
        sa = alone.stats()
        alone.close()
        del outs2
        torch.cuda.empty_cache()
        if sa["launches"] and sa["k1_kernel_ms"] > 0:
            ms = sa["k1_kernel_ms"] / sa["launches"]
            result["roofline"]["kernel_alone"] = {"kernel": "k_hf_lanes", "kernel_ms": round(ms, 4), "achieved": round(alg_launch / (ms / 1e3) / 1e9, 3), "frac": round(alg_launch / (ms / 1e3) / 8e12, 6), "launches": sa["launches"],
                                                  "how": "same frames and launch geometry, one batch in flight, LfGroup streams on the host threads: no other kernel beside k_hf_lanes (device-recorded start/end events)"}
            fa = sa["launch_frames"] / sa["launches"]
            alone_stages = [stage("plan build + LfGroup tail", "k_plan_*, LfGroup tail kernels", sa["lf_plan_ms"] / sa["launches"], fa, "one batch in flight, LfGroup streams on the host threads"),
                            stage("entropy decode", "k_hf_lanes", ms, fa, "one batch in flight, LfGroup streams on the host threads"),
                            stage("pixels", "k_vardct_*", sa["k2_ms"] / sa["launches"], fa, "one batch in flight, LfGroup streams on the host threads")]
            # the lane decoder of the LfGroup streams alone: one batch in flight and one step at a time, so that its launch is the only
            # thing on the device (the batch cannot start before it ends, and the step is drained before the next one)
            outs3 = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(min(B, 256))]
            os.environ["J40HIP_LF_WAIT_MS"] = "500"   # (this pipeline's launches wait for the whole batch: one launch of B frames, as in the steady state)
            lfa = j40_amd.Pipeline(local_rank, threads, min(args.pipe_batch, B), 1, lf_streams="device")
            del os.environ["J40HIP_LF_WAIT_MS"]
            so3 = [outs3[i % len(outs3)] for i in range(B)]
            run_pipeline_steps(lfa, step_bufs, step_sizes, so3, W * 4, True, 1, torch, dev, None)
            acc = {"ms": 0.0, "n": 0, "frames": 0, "waves": 0}
            for _ in range(2):
                run_pipeline_steps(lfa, step_bufs, step_sizes, so3, W * 4, True, 1, torch, dev, None)
                sl = lfa.stats()
                acc["ms"] += sl["lf_kernel_ms"]; acc["n"] += sl["lf_launches"]; acc["frames"] += sl["lf_launch_frames"]; acc["waves"] += sl["lf_launch_waves"]
            lfa.close()
            del outs3, so3
            torch.cuda.empty_cache()
            if acc["n"]:
                x = stage("LfGroup streams", _lf_kernel_name(), acc["ms"] / acc["n"], acc["frames"] / acc["n"],
                          "one batch in flight, one step at a time: the launch has the device to itself (%d launches, %.0f wavefronts each)" % (acc["n"], acc["waves"] / acc["n"]))
                if x:
                    alone_stages.insert(0, x)
            result["roofline"]["stages_alone"] = [x for x in alone_stages if x]
        result.update(sections(args, torch, np, j40_amd, dev, local_rank, datas, quota))
    else:
        pipe.close()
    print(json.dumps(result))


def sections(args, torch, np, j40_amd, dev, local_rank, datas, quota):
    """the figures beside `value`: kernels only on frames prepared ahead (round 1's measure), one frame alone, BASELINE's other configs"""
    from streams import synth
    W, H = args.width, args.height
    out = {}
    main = torch.cuda.current_stream(dev)
    # ---- device-resident: R frames parsed and uploaded ahead of time, one entropy launch per step ----
    R = max(1, args.resident_batch)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, quota)) as ex:
        frames = list(ex.map(lambda i: j40_amd.Frame(datas[i % len(datas)], threads=1), range(R)))
    for fr in frames:
        fr.upload(local_rank)
    routs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(R)]
    batch = j40_amd.Batch(frames)
    ptrs, strides = [o.data_ptr() for o in routs], [W * 4] * R
    batch.decode_recorded(ptrs, strides, main.cuda_stream, 0)
    torch.cuda.synchronize(dev)
    steps = 5
    t0 = time.perf_counter()
    for s in range(steps):
        batch.decode_recorded(ptrs, strides, main.cuda_stream, s)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    for fr in frames:
        assert fr.status() == ""
    k1 = [batch.elapsed(s)[0] for s in range(steps)]; k2 = [batch.elapsed(s)[1] for s in range(steps)]
    alg = sum(4 * W * H + len(datas[i % len(datas)]) for i in range(R))
    k1ms = sum(k1) / steps
    out["device_resident"] = {"value": round(W * H * R * steps / el / 1e6, 2), "unit": "Mpixels/s", "frames_per_step": R, "steps": steps, "ms_per_step": round(el / steps * 1e3, 3),
                              "k_hf_lanes_ms": round(k1ms, 3), "pixel_kernels_ms": round(sum(k2) / steps, 3),
                              "roofline_frac_k_hf_lanes": round(alg / (k1ms / 1e3) / 8e12, 6), "roofline_frac_step": round(alg * steps / el / 8e12, 6),
                              "note": "frames parsed, planned and uploaded before the clock starts; kernels only (what round 1 reported as value)"}
    # ---- one frame alone, latency mode (one section per wavefront) ----
    lat = [frames[0].decode_timed(routs[0].data_ptr(), W * 4, main.cuda_stream) for _ in range(3)]
    out["latency_mode"] = {"frame_ms": round(float(min(sum(map(float, m)) for m in lat)), 3), "k_hf_entropy_ms": round(float(min(float(m[0]) for m in lat)), 3),
                           "mpixels_per_s": round(W * H / min(sum(map(float, m)) for m in lat) / 1e3, 1)}
    # the unchanged public API, one caller: j40_from_memory -> j40_next_frame -> j40_frame_pixels_u8x4 (host pixels) -> j40_free. The first
    # call pays for what a process pays once (pinning the image plane, growing the device memory cache, the latency path's streams);
    # the median of the following calls is what a caller decoding image after image sees
    api_ms = []
    b0 = C.create_string_buffer(datas[0], len(datas[0]))
    for _ in range(7):
        err, ms_call, _ = j40_amd.decode_timed(b0, len(datas[0]))
        api_ms.append(ms_call)
        assert err == ""
    warm = sorted(api_ms[1:])
    out["latency_mode"]["public_api_from_memory_to_host_pixels_ms"] = round(api_ms[0], 2)
    out["latency_mode"]["public_api_cold_ms"] = round(api_ms[0], 2)
    out["latency_mode"]["public_api_warm_median_ms"] = round(warm[len(warm) // 2], 2)
    out["latency_mode"]["public_api_warm_calls_ms"] = [round(v, 2) for v in api_ms[1:]]
    # how the public API's call decodes a lone image: j40hip_frame_decode_to_host in two phases (the longest pass-group sections on a stream of
    # their own, the image over the link while they finish: DESIGN.md section 4 "One image alone") -- how many sections that was for this frame
    frames[0].decode_to_host()
    out["latency_mode"]["two_phase_long_sections"] = int(frames[0].two_phase_sections())
    out["latency_mode"]["note"] = ("frame_ms / k_hf_entropy_ms: the frame alone on the device in ONE phase (device-recorded: entropy launch, pixel kernels; no copy); "
                                   "public_api_*: host bytes -> host pixels through the unchanged API, which decodes in two phases when two_phase_long_sections > 0 (J40HIP_TWO_PHASE=0: never)")
    batch.close()
    for fr in frames:
        fr.close()
    del routs, frames
    torch.cuda.empty_cache()

    # ---- the restoration filters (Gaborish + two steps of the edge-preserving filter), opt-in; one 8K frame that signals them,
    # decoded with and without them on the single-frame path: the filter kernels' own device time (HIP events around them), their
    # algorithmic bytes -- 12 B/px read + 12 B/px written per pass over the three float planes, three passes -- against the HBM peak
    try:
        os.environ["J40HIP_RESTORATION_TIMING"] = "1"
        dr = synth("vardct", W, H, args.seed, fullheader=1, gab=1, epf=2)
        fr = j40_amd.Frame(dr, threads=min(8, quota)); fr.upload(local_rank)
        o = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
        plain = min(float(sum(fr.decode_timed(o.data_ptr(), W * 4, main.cuda_stream))) for _ in range(3))
        fr.set_restoration(1)
        with_f, filt = [], []
        for _ in range(3):
            with_f.append(float(sum(fr.decode_timed(o.data_ptr(), W * 4, main.cuda_stream)))); filt.append(fr.restoration_ms())
        assert fr.status() == ""
        passes = 3
        alg_f = passes * 24 * W * H + 4 * ((W + 7) // 8) * ((H + 7) // 8) * 2
        fms = min(filt)
        out["restoration"] = {"stream": "%dx%d VarDCT, Gaborish + 2 steps of the edge-preserving filter signalled (tools/jxlsynth fullheader=1 gab=1 epf=2)" % (W, H),
                              "frame_ms_filters_off": round(plain, 3), "frame_ms_filters_on": round(min(with_f), 3), "filter_kernels_ms": round(fms, 3),
                              "roofline": {"bound": "hbm", "achieved": round(alg_f / (fms / 1e3) / 1e9, 3) if fms > 0 else None, "peak": 8000.0, "unit": "GB/s", "frac": round(alg_f / (fms / 1e3) / 8e12, 6) if fms > 0 else None,
                                           "algorithmic_bytes": alg_f, "kernels": "k_epf_sigma, k_gaborish, k_epf<1>, k_epf<2> (restore_kernels.h)"},
                              "note": "off by default (j40 ignores the frame header's RestorationFilter bundle); J40HIP_RESTORATION=1 or j40hip_frame_set_restoration; parity: tests/test_restoration.py (the HIP kernels against the reference's own j40__gaborish / j40__epf, bit for bit)"}
        fr.close(); del o
    except Exception as e:   # (a measurement beside the contract's: never the reason the line is missing)
        out["restoration"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    # ---- BASELINE.json's other configurations ----
    cfg = {}
    # config 2: ONE 3840x2160 frame, a distance-1 encode of a procedural picture like the 8K bench stream (forward=1). Device time of
    # the latency path's kernels, and the contract clock of SURVEY 8d for one image: the public API, codestream bytes in host memory
    # -> RGBA in host memory (cold first call, median of the following ones), with the reference on one core beside it
    d4k = synth("vardct", 3840, 2160, 102, forward=1)
    fr = j40_amd.Frame(d4k, threads=min(8, quota)); fr.upload(local_rank)
    o = torch.empty((2160, 3840, 4), dtype=torch.uint8, device=dev)
    ms = min((fr.decode_timed(o.data_ptr(), 3840 * 4, main.cuda_stream) for _ in range(3)), key=lambda m: float(sum(m)))
    cfg["config2_3840x2160_one_gpu"] = {"frame_ms": round(float(sum(ms)), 3), "entropy_ms": round(float(ms[0]), 3), "pixel_kernels_ms": round(float(ms[1]), 3),
                                        "mpixels_per_s": round(3840 * 2160 / float(sum(ms)) / 1e3, 1), "mode": "latency (one frame alone, device time)",
                                        "stream": "forward-encoded at about distance 1 (tools/jxlsynth forward=1), %d bytes, %.3f bpp" % (len(d4k), 8.0 * len(d4k) / (3840 * 2160))}
    fr.close()
    api4k = []
    px4k = None
    b4k = C.create_string_buffer(d4k, len(d4k))
    for k in range(7):
        err, ms_call, got = j40_amd.decode_timed(b4k, len(d4k), want_pixels=(k == 6))
        api4k.append(ms_call)
        assert err == ""
        px4k = got if got is not None else px4k
    warm4k = sorted(api4k[1:])
    cfg["config2_3840x2160_one_gpu"]["host_to_host"] = {"cold_ms": round(api4k[0], 2), "warm_median_ms": round(warm4k[len(warm4k) // 2], 2), "mpixels_per_s": round(3840 * 2160 / warm4k[len(warm4k) // 2] / 1e3, 1),
                                                        "clock": "j40_from_memory -> j40_next_frame -> j40_frame_pixels_u8x4 -> j40_free on one thread: codestream bytes in host memory -> RGBA in host memory (SURVEY 8d)"}
    if not args.no_cpu_baseline:
        cb2 = cpu_baseline(d4k, 3840, 2160, budget_s=4.0)
        if cb2:
            cfg["config2_3840x2160_one_gpu"]["cpu_baseline"] = cb2
            d = np.abs(np.asarray(px4k)[:, :, :].astype(np.int16) - cpu_baseline.last_pixels.astype(np.int16))
            cfg["config2_3840x2160_one_gpu"]["parity_vs_reference"] = {"max_abs_diff": int(d.max()), "differing_samples": int((d > 0).sum()), "samples": int(d.size)}
            assert d.max() <= 1
    # config 5: 1024 independent 1920x1080 frames through the pipeline. Sections are the unit of the entropy launch (40 per frame:
    # 256 frames are 256 wavefronts, one per compute unit), so the frames go in large batches, several in flight; the LfGroup
    # streams of such small frames (one section of 130 k samples each) are decoded by the host threads: 0.8 ms of one core per
    # frame against 0.15 s of latency for the lane decoder's launch
    n5, d5 = 1024, 16
    d1080 = synth_many([("vardct", 1920, 1080, 110 + i, {}) for i in range(d5)], quota)
    b1080 = [C.create_string_buffer(d, len(d)) for d in d1080]
    o5 = [torch.empty((1080, 1920, 4), dtype=torch.uint8, device=dev) for _ in range(n5)]
    pipe = j40_amd.Pipeline(local_rank, max(2, quota), args.config5_batch, args.config5_in_flight, lf_streams=args.config5_lf)
    bb = [b1080[i % d5] for i in range(n5)]; ss = [len(d1080[i % d5]) for i in range(n5)]
    run_pipeline_steps(pipe, bb, ss, o5, 1920 * 4, True, 1, torch, dev, None)
    e5 = None
    for _ in range(3):
        e, tk = run_pipeline_steps(pipe, bb, ss, o5, 1920 * 4, True, 1, torch, dev, None)
        assert all(pipe.result(t) == "" for t in tk)
        if e5 is None or e < e5:
            e5, st = e, pipe.stats()
    for i in range(d5):   # per-stream pixels: frame i and frame i + 16 k decode the same stream
        assert torch.equal(o5[i], o5[i + d5 * (n5 // d5 - 1)])
    cfg["config5_1024x_1920x1080_batch"] = {"mpixels_per_s": round(1920 * 1080 * n5 / e5 / 1e6, 1), "seconds": round(e5, 3), "frames": n5, "distinct_streams": d5,
                                            "entropy_ms_per_launch": round(st["k1_ms"] / max(st["launches"], 1), 3), "frames_per_launch": round(st["launch_frames"] / max(st["launches"], 1), 1),
                                            "launches": st["launches"], "lf_streams": args.config5_lf, "in_flight": args.config5_in_flight,
                                            "mode": "pipeline (whole path per frame, codestream bytes in host memory -> RGBA in HBM; best of 3 passes over the 1024 frames); %d frames = %d sections per entropy launch instead of one hipStream per frame" % (args.config5_batch, 40 * args.config5_batch)}
    pipe.close()
    if not args.no_cpu_baseline:
        cb5 = cpu_baseline_many(d1080, 1920, 1080, quota)
        if cb5:
            cfg["config5_1024x_1920x1080_batch"]["cpu_baseline"] = cb5
    del o5
    torch.cuda.empty_cache()
    # config 1: 256x256 RGBA fjxl-like, single section
    d1 = synth("modular", 256, 256, 101, alpha=1, prefix=1, lz77=1)
    fr = j40_amd.Frame(d1); fr.upload(local_rank)
    o = torch.empty((256, 256, 4), dtype=torch.uint8, device=dev)
    ms = min((fr.decode_timed(o.data_ptr(), 256 * 4, main.cuda_stream) for _ in range(2)), key=lambda m: float(sum(m)))
    cfg["config1_256x256_modular_single_section"] = {"frame_ms": round(float(sum(ms)), 3), "mpixels_per_s": round(256 * 256 / float(sum(ms)) / 1e3, 2),
                                                     "note": "one sequential stream: bound by the serial latency of one wavefront; the reference's single core is faster here"}
    fr.close()
    if not args.skip_modular:
        for key, opts in (("config4_16384x16384_modular_rct_only (reference-pinned)", dict(tree=1, repeat=16)),
                          ("config4_16384x16384_modular_squeeze_and_rct (round-trip-pinned, PARITY UNPINNED vs libjxl)", dict(tree=1, repeat=16, squeeze=1))):
            d = synth("modular", 16384, 16384, 21, **opts)
            fr = j40_amd.Frame(d); fr.upload(local_rank)
            o = torch.empty((16384, 16384, 4), dtype=torch.uint8, device=dev)
            ms = fr.decode_timed(o.data_ptr(), 16384 * 4, main.cuda_stream)
            torch.cuda.synchronize(dev)
            assert fr.status() == ""
            cfg[key] = {"frame_ms": round(float(sum(ms)), 2), "sections_ms": round(float(ms[0]), 2), "inverse_transforms_and_pack_ms": round(float(ms[1]), 2),
                        "mpixels_per_s": round(16384 * 16384 / float(sum(ms)) / 1e3, 1), "codestream_mb": round(len(d) / 1e6, 1)}
            fr.close()
            del o
            torch.cuda.empty_cache()
    out["configs"] = cfg
    return out


def time_sharded(steps, warmup, torch, j40_amd, dist, dev, rank, local_rank, data):
    """seconds per step of one frame decoded by group ranges over the ranks (j40_amd.sharding): codestream broadcast from rank 0,
    per-rank partial decode, pixel rectangles sent to rank 0 (device tensors under nccl); max over the ranks"""
    from j40_amd import sharding
    decode = sharding.hip_range_decoder(local_rank)

    def step():
        if dist is not None:
            return sharding.decode_sharded(data if rank == 0 else b"", dist, decode, dev if dist.get_backend() == "nccl" else "cpu")
        err, full, _, _ = decode(data, 0, 1)
        assert err == "", err
        return full

    for _ in range(warmup):
        step()
    # (outside the clock) the assembled frame against the same stream decoded whole by rank 0 alone
    got = step()
    time_sharded.pixels_equal = None
    if rank == 0:
        err, whole, _, _ = decode(data, 0, 1)
        assert err == "", err
        time_sharded.pixels_equal = bool(torch.equal(got.cpu(), whole.cpu()))
        del whole
    del got
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed / steps


def sharded_records(args, torch, j40_amd, dist, dev, rank, local_rank, world):
    """the north star's split beside the frame-parallel figure (every rank calls this): one 7680x4320 VarDCT frame and, unless
    --skip-modular, one 16384x16384 Modular frame (RCT only) decoded by contiguous group ranges balanced by section bytes"""
    from streams import synth
    out = {}
    d8k = synth("vardct", 7680, 4320, args.seed, **({"forward": 1} if args.stream == "forward" else {})) if rank == 0 else b""
    sec = time_sharded(5, 1, torch, j40_amd, dist, dev, rank, local_rank, d8k)
    out["vardct_7680x4320"] = {"ms_per_frame": round(sec * 1e3, 3), "mpixels_per_s": round(7680 * 4320 / sec / 1e6, 1), "steps": 5, "pixels_equal_single_decode": time_sharded.pixels_equal,
                               "broadcast": "the parsed frame (LF bundle: codestream + LfGroup planes + tables, one parse on rank 0)" if __import__("j40_amd.sharding", fromlist=["x"]).LAST_FORM["lf_bundle"] else "the codestream (every rank parses it itself, concurrently)"}
    if not args.skip_modular:
        dm = synth("modular", 16384, 16384, 21, tree=1, repeat=16) if rank == 0 else b""
        sec = time_sharded(2, 1, torch, j40_amd, dist, dev, rank, local_rank, dm)
        out["modular_16384x16384_rct"] = {"ms_per_frame": round(sec * 1e3, 2), "mpixels_per_s": round(16384 * 16384 / sec / 1e6, 1), "steps": 2, "codestream_mb": round(len(dm) / 1e6, 1) if rank == 0 else None,
                                          "pixels_equal_single_decode": time_sharded.pixels_equal}
    out["note"] = ("one frame per step, its pass groups split over %d ranks in contiguous ranges balanced by section bytes; codestream broadcast from rank 0, every rank parses it, decodes its "
                   "range (j40hip_frame_set_group_range) and sends its pixel rectangles to rank 0 point-to-point (device tensors over RCCL). Strong scaling of a latency-bound step: "
                   "one 8K frame's 510 sections already run concurrently on one GPU and the step is its longest section (DESIGN.md section 6)") % world
    out["backend"] = dist.get_backend() if dist is not None else "none"
    return out


def bench_sharded(args, torch, j40_amd, dist, dev, rank, local_rank, world, data):
    """the north star's single-frame mode as the whole run (--shard-groups)"""
    W, H = args.width, args.height
    sec = time_sharded(args.steps, args.warmup, torch, j40_amd, dist, dev, rank, local_rank, data)
    if rank != 0:
        return
    print(json.dumps({
        "metric": METRIC, "value": round(W * H / sec / 1e6, 2), "unit": "Mpixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "one %dx%d %s synthetic frame per step, pass groups split over %d ranks (contiguous ranges balanced by section bytes), RGBA gathered on rank 0" % (W, H, "Modular lossless (RCT)" if args.shard_kind == "modular" else "VarDCT d1-like", world),
                   "frame_pixels": W * H, "codestream_bytes": len(data), "parallelism": "pass groups x%d" % world}}))


if __name__ == "__main__":
    main()
