#!/usr/bin/env python3
"""tools/latency_probe.py [runs = 7] -- one 8K frame of the bench's stream alone on the device (the single-image path's kernels: k_hf_entropy_fast
+ the pixel kernels, device-recorded) and through the unchanged public API (host bytes -> host pixels); J40HIP_LIB picks the library. One JSON line."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from streams import synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 7
out = {"lib": os.path.basename(j40_amd.LIB_PATH)}
for name, (w, h, seed) in {"8k": (7680, 4320, 3), "4k": (3840, 2160, 5)}.items():
    data = synth("vardct", w, h, seed, forward=1)
    fr = j40_amd.Frame(data, threads=8); fr.upload(0)
    dst = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    lat = [fr.decode_timed(dst.data_ptr(), w * 4, s) for _ in range(runs)]
    assert fr.status() == ""
    b0 = C.create_string_buffer(data, len(data))
    api = []
    for _ in range(runs):
        err, ms, _ = j40_amd.decode_timed(b0, len(data))
        assert err == ""
        api.append(ms)
    warm = sorted(api[1:])
    out[name] = {"k_hf_entropy_ms": round(min(float(m[0]) for m in lat), 3), "pixel_kernels_ms": round(min(float(m[1]) for m in lat), 3),
                 "api_cold_ms": round(api[0], 2), "api_warm_median_ms": round(warm[len(warm) // 2], 2), "checksum": int(dst.to(torch.int64).sum().item())}
    fr.close()
print(json.dumps(out))
j40_amd.shutdown()
