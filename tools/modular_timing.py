import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, j40_amd
from streams import synth
from refdec import Ref
ref = Ref()
for (w, h, opts) in [(2048, 2048, dict()), (2048, 2048, dict(tree=2)), (4096, 4096, dict())]:
    data = synth("modular", w, h, 5, **opts)
    fr = j40_amd.Frame(data); fr.upload(0)
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    fr.decode(out.data_ptr(), w * 4, s); torch.cuda.synchronize()
    ms = fr.decode_timed(out.data_ptr(), w * 4, s)
    t0 = time.perf_counter(); err, exp = ref.decode(data); t1 = time.perf_counter() - t0
    print(w, h, opts, "gpu ms", [round(float(x), 2) for x in ms], "ref s", round(t1, 2), "equal", bool(np.array_equal(out.cpu().numpy(), exp)), fr.status())
