#!/usr/bin/env python3
"""bench.py's `device_output` section alone (for rocprofv3): N 8K frames per step and per entropy launch through the pipeline, the RGBA left
in HBM. usage: python tools/device_output_probe.py [frames per launch = 512] [steps = 3] [lf = device] [in flight = 2] [distinct streams = 64]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from bench import run_pipeline_steps, synth_many, cpu_quota
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lf = sys.argv[3] if len(sys.argv) > 3 else "device"
in_flight = int(sys.argv[4]) if len(sys.argv) > 4 else 2
W, H, D = 7680, 4320, int(sys.argv[5]) if len(sys.argv) > 5 else 64
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
datas = synth_many([("vardct", W, H, 3 + 1000 * i, {"forward": 1}) for i in range(D)], cpu_quota())
bufs = [C.create_string_buffer(d, len(d)) for d in datas]
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(min(B, 256))]
sb = [bufs[i % D] for i in range(B)]; ss = [len(datas[i % D]) for i in range(B)]; so = [outs[i % len(outs)] for i in range(B)]
# (four worker threads like bench.py: sixteen run a 16-CPU container into its quota and the whole process is throttled)
pipe = j40_amd.Pipeline(0, int(os.environ.get("PROBE_THREADS", "4")) if lf == "device" else max(2, cpu_quota() // 2), B, in_flight, lf_streams=lf)
run_pipeline_steps(pipe, sb, ss, so, W * 4, True, 1, torch, dev, None)
el, tk = run_pipeline_steps(pipe, sb, ss, so, W * 4, True, steps, torch, dev, None)
st = pipe.stats()
assert all(pipe.result(t) == "" for t in tk)
n = max(st["launches"], 1)
print(json.dumps({"lib": os.path.basename(j40_amd.LIB_PATH), "frames_per_step": B, "in_flight": in_flight, "lf": lf, "steps": steps, "mpixels_per_s": round(W * H * B * steps / el / 1e6, 1), "ms_per_step": round(el / steps * 1e3, 2),
                  "k_hf_lanes_ms_per_launch": round(st["k1_kernel_ms"] / n, 3), "frames_per_launch": st["launch_frames"] / n, "pixel_kernels_ms_per_launch": round(st["k2_ms"] / n, 3),
                  "lf_plan_tail_ms_per_launch": round(st["lf_plan_ms"] / n, 3)}))
pipe.close()
j40_amd.shutdown()
