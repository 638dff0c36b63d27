#!/usr/bin/env python3
"""BASELINE config 5 alone: 1024 independent 1920x1080 VarDCT frames (16 distinct streams) through the pipeline, RGBA left in HBM.
usage: python tools/config5_probe.py [batch = 256] [in_flight = 4] [lf = host] [host threads = quota]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from bench import run_pipeline_steps, synth_many, cpu_quota
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
in_flight = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lf = sys.argv[3] if len(sys.argv) > 3 else "host"
threads = int(sys.argv[4]) if len(sys.argv) > 4 else max(2, cpu_quota())
n5, d5 = 1024, 16
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
d1080 = synth_many([("vardct", 1920, 1080, 110 + i, {}) for i in range(d5)], cpu_quota())
b1080 = [C.create_string_buffer(d, len(d)) for d in d1080]
o5 = [torch.empty((1080, 1920, 4), dtype=torch.uint8, device=dev) for _ in range(n5)]
pipe = j40_amd.Pipeline(0, threads, batch, in_flight, lf_streams=lf)
bb = [b1080[i % d5] for i in range(n5)]; ss = [len(d1080[i % d5]) for i in range(n5)]
run_pipeline_steps(pipe, bb, ss, o5, 1920 * 4, True, 1, torch, dev, None)
best = None
for _ in range(4):
    e, tk = run_pipeline_steps(pipe, bb, ss, o5, 1920 * 4, True, 1, torch, dev, None)
    assert all(pipe.result(t) == "" for t in tk)
    st = pipe.stats()
    if best is None or e < best[0]:
        best = (e, st)
e, st = best
n = max(st["launches"], 1)
print(json.dumps({"batch": batch, "in_flight": in_flight, "lf": lf, "threads": threads, "seconds": round(e, 4), "mpixels_per_s": round(1920 * 1080 * n5 / e / 1e6, 1), "launches": st["launches"],
                  "host_ms_per_frame": round(st["parse_thread_ms"] / n5, 3), "k1_ms": round(st["k1_kernel_ms"] / n, 2), "k2_ms": round(st["k2_ms"] / n, 2), "lf_plan_ms": round(st["lf_plan_ms"] / n, 2)}))
pipe.close(); j40_amd.shutdown()
