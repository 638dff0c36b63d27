#!/usr/bin/env python3
"""bench.py -- Mpixels/s RGBA-u8x4 decode of a synthetic 8K VarDCT (d1-like) frame on MI355X.

A *step* is one pass of the hot path over one frame whose inputs (codestream, code specs, orders,
dequant tables, LF bundle) are already resident in HBM: coefficient clear + entropy decode of every
pass-group section (K1) + dequant / chroma-from-luma / inverse transforms / XYB->sRGB / RGBA pack
(K2 family). Output stays in HBM. Host parsing and PCIe copies are outside the timed region and
reported separately (`e2e_*` fields).

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
    return {"value": round(width * height / best / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
            "sample": "%d full decodes of the same %dx%d stream through the reference's public API (best of %d, %.2f s each), 1 of %d host cores" % (len(times), width, height, len(times), best, os.cpu_count() or 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="frames per step per GPU; > 1 uses the throughput mode (j40hip_batch_*: one section per lane)")
    ap.add_argument("--distinct", type=int, default=4, help="number of distinct streams a batch cycles through")
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
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
        dist.barrier()
    if not torch.cuda.is_available() or j40_amd.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

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
    if args.shard_groups and world > 1:
        raise SystemExit("--shard-groups is driven by j40_amd.sharding (see tests/test_sharding.py); not part of the default bench")

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
        "e2e": {"host_parse_ms": round(t_parse * 1e3, 2), "plan_upload_ms": round(t_upload * 1e3, 2), "decode_plus_copy_back_ms": round(t_e2e_tail * 1e3, 2),
                "mpixels_per_s": round(W * H / (t_parse + t_upload + t_e2e_tail) / 1e6, 2)},
    }
    if not args.no_cpu_baseline:
        cb = cpu_baseline(data, W, H)
        if cb:
            result["cpu_baseline"] = cb
    del host
    print(json.dumps(result))


def bench_batch(args, torch, j40_amd, synth, dist, dev, rank, local_rank, world, frame0, data0, out0):
    """throughput mode: `--batch` frames per step, one entropy launch for all of them"""
    W, H, B = args.width, args.height, args.batch
    sptr = torch.cuda.current_stream(dev).cuda_stream
    datas = [data0] + [synth("vardct", W, H, args.seed + 1000 * (i + 1) + rank) for i in range(min(args.distinct, B) - 1)]
    frames, outs = [frame0], [out0]
    for i in range(1, B):
        fr = j40_amd.Frame(datas[i % len(datas)], threads=min(8, os.cpu_count() or 1))
        fr.upload(local_rank)
        frames.append(fr)
        outs.append(torch.empty((H, W, 4), dtype=torch.uint8, device=dev))
    batch = j40_amd.Batch(frames)
    ptrs, strides = [o.data_ptr() for o in outs], [W * 4] * B
    for _ in range(max(args.warmup, 1)):
        batch.decode(ptrs, strides, sptr)
    torch.cuda.synchronize(dev)
    for fr in frames:
        assert fr.status() == "", "decode error: " + fr.status()
    # the batch path must give the pixels of the single-frame path
    check = torch.empty_like(out0)
    frame0.decode(check.data_ptr(), W * 4, sptr)
    torch.cuda.synchronize(dev)
    assert torch.equal(check, out0) or os.environ.get("J40HIP_EXP_SAME_GROUP"), "batch and single-frame decodes differ"
    del check
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    k1_ms, k2_ms, misc_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b, c = batch.decode_timed(ptrs, strides, sptr)
        k1_ms.append(a); k2_ms.append(b); misc_ms.append(c)
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
    value = W * H * B * args.steps * world / elapsed / 1e6
    alg_bytes = sum(4 * W * H + len(datas[i % len(datas)]) for i in range(B))
    k1 = sum(k1_ms) / len(k1_ms) / 1e3
    achieved = alg_bytes / k1 / 1e9
    result = {
        "metric": "Mpixels/s RGBA-u8x4 decode, 8K VarDCT d1",
        "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d x %dx%d VarDCT d1-like synthetic frames per GPU per step (tools/jxlsynth, %d distinct streams, %.3f bpp, %d pass groups each), throughput mode, inputs resident in HBM"
                               % (B, W, H, len(datas), 8.0 * len(data0) / (W * H), frame0.info["num_groups"]),
                   "frame_pixels": W * H, "frames_per_step": B, "codestream_bytes": len(data0), "parallelism": "frames x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": None,
                     "kernel": "k_hf_entropy_lanes", "kernel_ms": round(k1 * 1e3, 4), "algorithmic_bytes_per_launch": alg_bytes},
        "kernels_ms": {"k_hf_entropy_lanes": round(k1 * 1e3, 4), "vardct_to_rgba_kernels": round(sum(k2_ms) / len(k2_ms), 4), "clear_coefficients": round(sum(misc_ms) / len(misc_ms), 4)},
    }
    if not args.no_cpu_baseline:
        cb = cpu_baseline(data0, W, H)
        if cb:
            result["cpu_baseline"] = cb
    print(json.dumps(result))


if __name__ == "__main__":
    main()
