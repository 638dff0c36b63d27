#!/usr/bin/env python3
"""Round 5's A/B probe: one process, one configuration (J40HIP_LIB / J40HIP_LF_KERNEL / J40HIP_LF_ROWS_LDS_KB ... from the environment),
three passes over N forward-encoded 8K frames with the pixels left in HBM, one JSON line:
  alone      one batch in flight, LfGroup streams on the host threads: the entropy kernel's and the pixel stage's own durations
  lf_alone   one batch in flight, LfGroup streams on the device, one step at a time: the lane decoder's launch has the device to itself
  device     two batches in flight, LfGroup streams on the device: the steady step, and each stage's duration inside it
usage: python tools/r05_probe.py [frames = 256] [distinct streams = 16] [steps = 4]      PROBE_ONLY=alone|lf_alone|device limits the passes"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from bench import run_pipeline_steps, synth_many, cpu_quota
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(os.environ.get("PROBE_STEPS", sys.argv[3] if len(sys.argv) > 3 else 4))
only = os.environ.get("PROBE_ONLY", "")
W, H = 7680, 4320
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
datas = synth_many([("vardct", W, H, 3 + 1000 * i, {"forward": 1}) for i in range(D)], cpu_quota())
bufs = [C.create_string_buffer(d, len(d)) for d in datas]
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(min(B, 256))]
sb = [bufs[i % D] for i in range(B)]; ss = [len(datas[i % D]) for i in range(B)]; so = [outs[i % len(outs)] for i in range(B)]
out = {"lib": os.path.basename(j40_amd.LIB_PATH), "env": {k: v for k, v in os.environ.items() if k.startswith("J40HIP_") and k != "J40HIP_LIB"}, "frames": B}


def per_launch(st):
    n = max(st["launches"], 1); nl = max(st["lf_launches"], 1)
    return {"k_hf_lanes_ms": round(st["k1_kernel_ms"] / n, 3), "pixel_stage_ms": round(st["k2_ms"] / n, 3), "plan_tail_ms": round(st["lf_plan_ms"] / n, 3), "frames_per_launch": st["launch_frames"] / n,
            "lf_kernel_ms": round(st["lf_kernel_ms"] / nl, 3), "lf_launches": st["lf_launches"], "lf_frames_per_launch": st["lf_launch_frames"] / nl, "lf_waves_per_launch": st["lf_launch_waves"] / nl}


if only in ("", "alone"):
    pipe = j40_amd.Pipeline(0, max(2, cpu_quota() // 2), B, 1, lf_streams="host")
    run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
    el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 2, torch, dev, None)
    assert all(pipe.result(t) == "" for t in tk)
    out["alone"] = per_launch(pipe.stats())
    pipe.close()
if only in ("", "lf_alone"):
    os.environ["J40HIP_LF_WAIT_MS"] = "500"   # (one launch per step: it waits for the whole batch)
    pipe = j40_amd.Pipeline(0, 4, B, 1, lf_streams="device")
    del os.environ["J40HIP_LF_WAIT_MS"]
    run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
    acc = []
    for _ in range(2):
        el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
        assert all(pipe.result(t) == "" for t in tk)
        acc.append(per_launch(pipe.stats()))
    out["lf_alone"] = {"lf_kernel_ms": [a["lf_kernel_ms"] for a in acc], "lf_waves_per_launch": acc[-1]["lf_waves_per_launch"], "lf_frames_per_launch": acc[-1]["lf_frames_per_launch"], "ms_per_step": round(el * 1e3, 2)}
    pipe.close()
if only in ("", "device"):
    pipe = j40_amd.Pipeline(0, int(os.environ.get("PROBE_THREADS", "4")), B, 2, lf_streams="device")
    run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 2, torch, dev, None)   # (two: the cache is sized at the second full batch)
    el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, steps, torch, dev, None)
    assert all(pipe.result(t) == "" for t in tk)
    out["device"] = dict(per_launch(pipe.stats()), ms_per_step=round(el / steps * 1e3, 2), mpixels_per_s=round(W * H * B * steps / el / 1e6, 1), steps=steps)
    pipe.close()
print(json.dumps(out))
j40_amd.shutdown()
