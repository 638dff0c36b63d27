#!/usr/bin/env python3
"""bench.py -- Mpixels/s RGBA-u8x4 decode of a synthetic 8K VarDCT (d1-like) frame on MI355X.

A *step* is one pass of the hot path over one batch of frames (`--batch`, default 256 per GPU) whose inputs
(codestreams, code specs, orders, dequant tables, LF bundles) are already resident in HBM: coefficient clear +
entropy decode of every pass-group section (K1) + dequant / chroma-from-luma / inverse transforms /
XYB->sRGB / RGBA pack (K2 family). Outputs stay in HBM. Host parsing and PCIe copies are outside the timed
region. Two launch modes exist: the throughput mode (default; one section per wavefront LANE, every frame of the
batch in one entropy launch -- 288 GB of HBM hold the working sets of hundreds of 8K frames) and the latency
mode (`--batch 1`; one section per wavefront, scalarised decoder; reports `e2e_*` fields for a single frame).
The default run also times one frame in latency mode and reports it as `latency_mode`.

N > 1 (launched by torch.distributed.run, one rank per GPU): frames are independent units, so every
rank decodes its own frame (weak scaling, no data-path collective); `value` is the whole-job
aggregate. `--shard-groups` switches to the north star's single-frame sharding (LF bundle broadcast
from rank 0, pass groups split in row bands, RGBA bands gathered on rank 0 over RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_baseline(data, width, height, budget_s=12.0):
    """the unmodified reference (oracle/_ref) on ONE host core, same codestream, bounded sample"""
    from refdec import Ref, REF_SO
    import ctypes as C
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
            "sample": "%d full decodes of the same %dx%d stream through the reference's public API (best of %d, %.2f s each), 1 of %d host cores" % (len(times), width, height, len(times), best, os.cpu_count() or 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU; > 1 uses the throughput mode (j40hip_batch_*: one section per lane), 1 the latency mode")
    ap.add_argument("--distinct", type=int, default=4, help="number of distinct streams a batch cycles through")
    ap.add_argument("--streams", type=int, default=1, help="throughput mode: sub-batches in flight on separate HIP streams")
    ap.add_argument("--shard-groups", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import numpy as np
    import j40_amd
    from streams import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        import __graft_entry__
        if not os.path.exists(j40_amd.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "build", "jxlsynth")):
            __graft_entry__.build()
    if not torch.cuda.is_available() or j40_amd.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)   # before the first collective: RCCL binds a rank to its current device
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)
        dist.barrier()

    W, H = args.width, args.height
    # every rank decodes its own frame (distinct seed) unless the groups of one frame are sharded
    seed = args.seed if args.shard_groups else args.seed + rank
    data = synth("vardct", W, H, seed)
    t0 = time.perf_counter()
    frame = j40_amd.Frame(data, threads=min(8, os.cpu_count() or 1))
    t_parse = time.perf_counter() - t0
    t0 = time.perf_counter()
    frame.upload(local_rank)
    t_upload = time.perf_counter() - t0
    t_upload_again = None
    if args.batch == 1 and not args.shard_groups:   # what a server pays per new frame: device blocks are recycled (runtime.hip)
        other = j40_amd.Frame(data, threads=min(8, os.cpu_count() or 1)); other.upload(local_rank); other.close()
        t0 = time.perf_counter()
        frame.upload(local_rank)
        t_upload_again = time.perf_counter() - t0
    if args.shard_groups:
        return bench_sharded(args, torch, j40_amd, dist, dev, rank, local_rank, world, data)

    out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    if args.batch > 1:
        return bench_batch(args, torch, j40_amd, synth, dist, dev, rank, local_rank, world, frame, data, out)

    for _ in range(args.warmup):
        frame.decode(out.data_ptr(), W * 4, sptr)
    torch.cuda.synchronize(dev)
    assert frame.status() == "", "decode error: " + frame.status()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    k1_ms, k2_ms, misc_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ms = frame.decode_timed(out.data_ptr(), W * 4, sptr)   # HIP events on the launch stream
        k1_ms.append(float(ms[0])); k2_ms.append(float(ms[1])); misc_ms.append(float(ms[2]))
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert frame.status() == ""

    if rank != 0:
        return
    frames_total = args.steps * world
    value = W * H * frames_total / elapsed / 1e6
    # roofline of the dominant kernel (K1, entropy decode): algorithmic bytes of the whole path per
    # frame = RGBA written + codestream read, over K1's average launch time
    alg_bytes = 4 * W * H + len(data)
    k1 = sum(k1_ms) / len(k1_ms) / 1e3
    achieved = alg_bytes / k1 / 1e9
    # end to end for one frame: host parse + upload + decode + copy back (not the headline)
    t0 = time.perf_counter()
    frame.decode(out.data_ptr(), W * 4, sptr)
    host = out.cpu()
    t_e2e_tail = time.perf_counter() - t0
    result = {
        "metric": "Mpixels/s RGBA-u8x4 decode, 8K VarDCT d1",
        "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d VarDCT d1-like synthetic frame (tools/jxlsynth seed %d, %.3f bpp, %d pass groups), one frame per GPU per step, inputs resident in HBM" % (W, H, args.seed, 8.0 * len(data) / (W * H), frame.info["num_groups"]),
                   "frame_pixels": W * H, "codestream_bytes": len(data), "parallelism": "frames x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": None,
                     "kernel": "k_hf_entropy", "kernel_ms": round(k1 * 1e3, 4), "algorithmic_bytes_per_launch": alg_bytes},
        "kernels_ms": {"k_hf_entropy": round(k1 * 1e3, 4), "vardct_to_rgba_kernels": round(sum(k2_ms) / len(k2_ms), 4), "clear_coefficients": round(sum(misc_ms) / len(misc_ms), 4)},
        "e2e": {"host_parse_ms": round(t_parse * 1e3, 2), "plan_upload_first_ms": round(t_upload * 1e3, 2), "plan_upload_recycled_ms": round(t_upload_again * 1e3, 2),
                "decode_plus_copy_back_ms": round(t_e2e_tail * 1e3, 2), "mpixels_per_s": round(W * H / (t_parse + t_upload_again + t_e2e_tail) / 1e6, 2)},
    }
    if not args.no_cpu_baseline:
        cb = cpu_baseline(data, W, H)
        if cb:
            result["cpu_baseline"] = cb
    del host
    print(json.dumps(result))


def bench_sharded(args, torch, j40_amd, dist, dev, rank, local_rank, world, data):
    """the north star's single-frame mode: every step decodes ONE frame whose pass groups are split in row bands over the
    ranks (j40_amd.sharding): codestream broadcast from rank 0, per-rank partial decode, RGBA bands gathered on rank 0
    over RCCL. Strong scaling of a latency-bound step: see DESIGN.md section 6 for why this does not speed a frame up."""
    from j40_amd import sharding
    W, H = args.width, args.height
    if dist is not None:
        data = sharding.broadcast_bytes(data, dist, dev)
    frame = j40_amd.Frame(data, threads=min(8, os.cpu_count() or 1))
    frame.upload(local_rank)
    first, count, y0, y1 = sharding.rank_share(W, H, frame.info["group_size_shift"], world, rank)
    frame.set_group_range(first, count)
    full = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    sptr = torch.cuda.current_stream(dev).cuda_stream

    def step():
        if count:
            frame.decode(full.data_ptr(), W * 4, sptr)
        return sharding.gather_bands(full[y0:y1], W, H, frame.info["group_size_shift"], dist) if dist is not None else full

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    assert frame.status() == ""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    print(json.dumps({
        "metric": "Mpixels/s RGBA-u8x4 decode, 8K VarDCT d1", "value": round(W * H * args.steps / elapsed / 1e6, 2), "unit": "Mpixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "one %dx%d VarDCT d1-like synthetic frame per step, pass groups split in %d row bands, RGBA gathered on rank 0" % (W, H, world),
                   "frame_pixels": W * H, "codestream_bytes": len(data), "parallelism": "group rows x%d" % world}}))


def bench_batch(args, torch, j40_amd, synth, dist, dev, rank, local_rank, world, frame0, data0, out0):
    """throughput mode. A step decodes `--batch` frames; they are organised as `--streams` sub-batches, each with its
    own HIP stream, launched back to back so that one sub-batch's entropy kernel (latency bound, few issue slots)
    overlaps the pixel kernels (VALU bound) of another. Every launch is measured with HIP events on its own stream."""
    W, H, B, S = args.width, args.height, args.batch, max(1, min(args.streams, args.batch))
    datas = [data0] + [synth("vardct", W, H, args.seed + 1000 * (i + 1) + rank) for i in range(min(args.distinct, B) - 1)]
    frames, outs = [frame0], [out0]
    for i in range(1, B):
        fr = j40_amd.Frame(datas[i % len(datas)], threads=min(8, os.cpu_count() or 1))
        fr.upload(local_rank)
        frames.append(fr)
        outs.append(torch.empty((H, W, 4), dtype=torch.uint8, device=dev))
    subs = []
    for k in range(S):
        idx = list(range(k, B, S))
        subs.append({"batch": j40_amd.Batch([frames[i] for i in idx]), "ptrs": [outs[i].data_ptr() for i in idx], "strides": [W * 4] * len(idx),
                     "stream": torch.cuda.Stream(device=dev), "n": len(idx)})
    main = torch.cuda.current_stream(dev)

    def run_step(slot, stagger=False):
        for k, sb in enumerate(subs):
            if stagger and k > 0:   # start one entropy launch behind the previous sub-batch: from then on the streams stay out of phase
                subs[k - 1]["batch"].wait_stage(slot, 2, sb["stream"].cuda_stream)
            sb["batch"].decode_recorded(sb["ptrs"], sb["strides"], sb["stream"].cuda_stream, slot)

    for w in range(max(args.warmup, 1)):
        run_step(0, stagger=(w == 0))
    torch.cuda.synchronize(dev)
    for fr in frames:
        assert fr.status() == "", "decode error: " + fr.status()
    # the batch path must give the pixels of the single-frame path
    check = torch.empty_like(out0)
    frame0.decode(check.data_ptr(), W * 4, main.cuda_stream)
    torch.cuda.synchronize(dev)
    assert torch.equal(check, out0), "batch and single-frame decodes differ"
    del check
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for step in range(args.steps):
        run_step(step)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for fr in frames:
        assert fr.status() == "", "decode error: " + fr.status()
    if rank != 0:
        return
    # dominant kernel: the entropy launch of a sub-batch; average duration and algorithmic bytes per launch
    k1_ms, k2_ms, misc_ms = [], [], []
    for step in range(args.steps):
        for sb in subs:
            a, b_, c = sb["batch"].elapsed(step)
            k1_ms.append(a); k2_ms.append(b_); misc_ms.append(c)
    value = W * H * B * args.steps * world / elapsed / 1e6
    alg_total = sum(4 * W * H + len(datas[i % len(datas)]) for i in range(B))
    alg_per_launch = alg_total / S
    k1 = sum(k1_ms) / len(k1_ms) / 1e3
    achieved = alg_per_launch / k1 / 1e9
    result = {
        "metric": "Mpixels/s RGBA-u8x4 decode, 8K VarDCT d1",
        "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d x %dx%d VarDCT d1-like synthetic frames per GPU per step (tools/jxlsynth, %d distinct streams, %.3f bpp, %d pass groups each), throughput mode: %d sub-batches on their own HIP streams, inputs resident in HBM"
                               % (B, W, H, len(datas), 8.0 * len(data0) / (W * H), frame0.info["num_groups"], S),
                   "frame_pixels": W * H, "frames_per_step": B, "streams": S, "codestream_bytes": len(data0), "parallelism": "frames x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": None,
                     "kernel": "k_hf_lanes", "kernel_ms": round(k1 * 1e3, 4), "launches_per_step": S, "algorithmic_bytes_per_launch": int(alg_per_launch)},
        "kernels_ms": {"k_hf_lanes (per launch, %d frames)" % subs[0]["n"]: round(k1 * 1e3, 4), "vardct_to_rgba_kernels (per sub-batch)": round(sum(k2_ms) / len(k2_ms), 4),
                       "clear_coefficients (per sub-batch)": round(sum(misc_ms) / len(misc_ms), 4)},
    }
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure comes from the
    # committed rocprofv3 passes of this same command (profiles/pmc_traffic.json) and is only reported for a matching launch
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pt.get("frames_per_launch") == subs[0]["n"] and (W, H) == (7680, 4320):
            result["roofline"]["traffic"] = int((pt["fetch_size_kb"] + pt["write_size_kb"]) * 1000)
            result["roofline"]["traffic_source"] = pt["source"]
    except (OSError, ValueError, KeyError):
        pass
    # the same frame alone, latency mode (one section per wavefront)
    lat = [frame0.decode_timed(out0.data_ptr(), W * 4, main.cuda_stream) for _ in range(3)]
    result["latency_mode"] = {"frame_ms": round(float(min(sum(map(float, m)) for m in lat)), 3), "k_hf_entropy_ms": round(float(min(float(m[0]) for m in lat)), 3),
                              "mpixels_per_s": round(W * H / min(sum(map(float, m)) for m in lat) / 1e3, 1)}
    if not args.no_cpu_baseline:
        cb = cpu_baseline(data0, W, H)
        if cb:
            result["cpu_baseline"] = cb
            # parity at the full size: the frame the batch decoded against the reference's pixels (bar: 1 level)
            import numpy as np
            d = np.abs(out0.cpu().numpy().astype(np.int16) - cpu_baseline.last_pixels.astype(np.int16))
            result["parity_vs_reference"] = {"max_abs_diff": int(d.max()), "differing_samples": int((d > 0).sum()), "samples": int(d.size)}
            assert d.max() <= 1, "GPU and reference pixels differ by more than one level"
    print(json.dumps(result))


if __name__ == "__main__":
    main()
