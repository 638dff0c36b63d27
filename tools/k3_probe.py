"""times the Modular section kernel (K3) on a few frames; J40HIP_K3_LANES=1 selects the one-section-per-lane form"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, numpy as np, j40_amd
from streams import synth
cases = [("256x256 fjxl-like", 256, 256, dict(alpha=1, prefix=1, lz77=1)), ("2048x2048 tree=1", 2048, 2048, dict(tree=1)), ("2048x2048 wp", 2048, 2048, dict(tree=2)),
         ("16384x16384 rct", 16384, 16384, dict(tree=1, repeat=16))]
if len(sys.argv) > 1: cases = cases[: int(sys.argv[1])]
for name, w, h, o in cases:
    d = synth("modular", w, h, 21, **o)
    fr = j40_amd.Frame(d); print("   sections (cooperative, all):", fr.coop_sections(), "four to a wavefront:", fr.quad_sections()); fr.upload(0)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    for rep in range(2):
        ms = fr.decode_timed(out.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("%s: K3 %.2f ms, transforms+pack %.2f ms, status %r, %.1f Mpx/s" % (name, ms[0], ms[1], fr.status(), w * h / (ms[0] + ms[1]) / 1e3), flush=True)
    fr.close()
