#!/usr/bin/env python3
"""The batch's stages with the device to themselves: N 8K frames per entropy launch, ONE batch in flight, LfGroup streams on the host
threads, RGBA left in HBM -- the entropy kernel's and the pixel stage's own durations (device events), for A/B runs of variant
libraries (J40HIP_LIB). usage: python tools/stages_alone_probe.py [frames per launch = 256] [steps = 4] [distinct streams = 8]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from bench import run_pipeline_steps, synth_many, cpu_quota
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H = 7680, 4320
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
datas = synth_many([("vardct", W, H, 3 + 1000 * i, {"forward": 1}) for i in range(D)], cpu_quota())
bufs = [C.create_string_buffer(d, len(d)) for d in datas]
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(min(B, 256))]
sb = [bufs[i % D] for i in range(B)]; ss = [len(datas[i % D]) for i in range(B)]; so = [outs[i % len(outs)] for i in range(B)]
pipe = j40_amd.Pipeline(0, max(2, cpu_quota() // 2), B, 1, lf_streams="host")
run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
pipe.reset_stats()
phases = None
if os.environ.get("PROBE_K2_PHASES"):   # (a library built with -DJ40_K2_PHASES: kernels.hip)
    phases = (C.c_ulonglong * (64 * 8))()
    j40_amd.lib().j40hip_debug_k2_phases.argtypes = [C.c_void_p, C.c_int]
    j40_amd.lib().j40hip_debug_k2_phases(phases, 1)
el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, steps, torch, dev, None)
st = pipe.stats()
assert all(pipe.result(t) == "" for t in tk)
n = max(st["launches"], 1)
print(json.dumps({"lib": os.path.basename(j40_amd.LIB_PATH), "frames_per_launch": st["launch_frames"] / n, "launches": st["launches"], "k_hf_lanes_ms_per_launch": round(st["k1_kernel_ms"] / n, 3),
                  "entropy_stage_ms_per_launch": round(st["k1_ms"] / n, 3), "pixel_kernels_ms_per_launch": round(st["k2_ms"] / n, 3), "lf_plan_tail_ms_per_launch": round(st["lf_plan_ms"] / n, 3),
                  "ms_per_step": round(el / steps * 1e3, 2)}))
if phases is not None:
    j40_amd.lib().j40hip_debug_k2_phases(phases, 0)
    names = ["prologue", "zero", "scatter+llf", "pass1", "pass2", "colour", "end barrier"]
    total_all = sum(phases[s * 8 + k] for s in range(64) for k in range(7)) or 1
    for slot in range(64):
        row = [phases[slot * 8 + k] for k in range(8)]
        if row[7]:
            tot = sum(row[:7])
            print(json.dumps({"kernel": "k_vardct_special, DctSelect %s" % ("1-3" if slot == 0 else "12-17") if slot < 2 else "k_vardct_dct %dx%d" % (1 << (slot // 8), 1 << (slot % 8)), "tiles": row[7], "share_of_all_shapes": round(tot / total_all, 4), "clocks_per_tile": round(tot / row[7], 1),
                              "phases": {n: round(v / tot, 4) for n, v in zip(names, row[:7])}}))
pipe.close()
j40_amd.shutdown()
