"""tools/restore_probe.py -- one 8K VarDCT frame that signals Gaborish + EPF, decoded a few times with the filters on (for a
rocprofv3 --kernel-trace --stats run: the filter kernels' own durations) and the restoration_ms the library's HIP events report."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["J40HIP_RESTORATION_TIMING"] = "1"
import torch
import j40_amd
sys.path.insert(0, os.path.join(ROOT, "tests"))
from streams import synth

W, H = 7680, 4320
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for epf in (2, 3):
    data = synth("vardct", W, H, 7, fullheader=1, gab=1, epf=epf)
    fr = j40_amd.Frame(data, threads=8); fr.upload(0)
    out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    fr.set_restoration(1)
    ms = []
    for _ in range(iters):
        fr.decode_timed(out.data_ptr(), W * 4, st); ms.append(fr.restoration_ms())
    assert fr.status() == "", fr.status()
    print("epf=%d filter kernels ms: %s" % (epf, ["%.3f" % v for v in ms]), flush=True)
    fr.close()
