#!/usr/bin/env python3
"""The pixel stage on streams full of 128- and 256-sized transforms (k_vardct_large): N 8K coefficient-domain frames with maxlog = 8 per
step and per entropy launch through the pipeline, the RGBA left in HBM. J40HIP_LARGE_IDCT=sweeps selects round 3's kernel.
usage: python tools/large_probe.py [frames per launch = 64] [steps = 3] [distinct streams = 16]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from bench import run_pipeline_steps, synth_many, cpu_quota
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W, H = 7680, 4320
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
datas = synth_many([("vardct", W, H, 3 + 1000 * i, {"maxlog": 8}) for i in range(D)], cpu_quota())
bufs = [C.create_string_buffer(d, len(d)) for d in datas]
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(min(B, 64))]
sb = [bufs[i % D] for i in range(B)]; ss = [len(datas[i % D]) for i in range(B)]; so = [outs[i % len(outs)] for i in range(B)]
pipe = j40_amd.Pipeline(0, 4, B, 2, lf_streams="host")
run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, steps, torch, dev, None)
st = pipe.stats()
assert all(pipe.result(t) == "" for t in tk)
n = max(st["launches"], 1)
print(json.dumps({"large_idct": os.environ.get("J40HIP_LARGE_IDCT", "levels + 64-point registers"), "frames_per_step": B, "steps": steps, "mpixels_per_s": round(W * H * B * steps / el / 1e6, 1),
                  "ms_per_step": round(el / steps * 1e3, 2), "k_hf_lanes_ms_per_launch": round(st["k1_kernel_ms"] / n, 3), "frames_per_launch": st["launch_frames"] / n,
                  "pixel_kernels_ms_per_launch": round(st["k2_ms"] / n, 3)}))
pipe.close()
j40_amd.shutdown()
