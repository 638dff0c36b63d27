"""times the Squeeze frames: 2048^2 and 16384^2 (sections / inverse transforms + pack)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, j40_amd
from streams import synth
for w, h, o in [(2048, 2048, dict(tree=1, squeeze=1)), (16384, 16384, dict(tree=1, repeat=16, squeeze=1))]:
    d = synth("modular", w, h, 21, **o)
    fr = j40_amd.Frame(d); fr.upload(0)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    for rep in range(2): ms = fr.decode_timed(out.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("%dx%d squeeze: sections %.2f ms, inverse transforms + pack %.2f ms, status %r" % (w, h, ms[0], ms[1], fr.status()), flush=True)
    fr.close()
