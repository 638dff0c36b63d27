"""tools/config1_probe.py -- BASELINE config 1 (256 x 256 RGBA Modular, one section, prefix codes + LZ77) five times through Frame.decode_timed, for
rocprofv3 --kernel-trace --stats: k_modular_tokens 24.0 ms, k_modular_predict 1.0 ms (round 6, call O). Needs an MI355X."""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, j40_amd
from streams import synth
data = synth("modular", 256, 256, 101, alpha=1, prefix=1, lz77=1)
fr = j40_amd.Frame(data); fr.upload(0)
o = torch.empty((256, 256, 4), dtype=torch.uint8, device="cuda:0")
for _ in range(5):
    ms = fr.decode_timed(o.data_ptr(), 1024, torch.cuda.current_stream().cuda_stream)
print([round(float(x), 3) for x in ms], fr.status(), fr.split_sections(), len(data))
