"""device-resident probe of the batch decode: R 8K frames (D distinct streams) parsed + uploaded ahead, then timed batch decodes;
prints the entropy-stage and pixel-stage durations (HIP events). usage: k1_probe.py R [D] [steps]"""
import sys, os, time, concurrent.futures
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, j40_amd
from streams import synth
W, H = 7680, 4320
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    datas = list(ex.map(lambda i: synth("vardct", W, H, 1000 + i), range(D)))
    frames = list(ex.map(lambda i: j40_amd.Frame(datas[i % D], threads=1), range(R)))
for fr in frames: fr.upload(0)
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(R)]
b = j40_amd.Batch(frames)
ptrs, strides = [o.data_ptr() for o in outs], [W * 4] * R
s = torch.cuda.current_stream().cuda_stream
b.decode_recorded(ptrs, strides, s, 0); torch.cuda.synchronize()
for fr in frames: assert fr.status() == "", fr.status()
ref = outs[0].clone()
t0 = time.time()
for i in range(steps): b.decode_recorded(ptrs, strides, s, i)
torch.cuda.synchronize(); dt = time.time() - t0
k1 = sum(b.elapsed(i)[0] for i in range(steps)) / steps; k2 = sum(b.elapsed(i)[1] for i in range(steps)) / steps
assert torch.equal(ref, outs[0])
print("R=%d gen=%s wpw=%s: K1 %.2f ms, K2 %.2f ms, step %.2f ms, %.0f Mpx/s" % (R, os.environ.get("J40HIP_LANES_GEN", "2"), os.environ.get("J40HIP_WAVES_PER_WG", "auto"), k1, k2, dt / steps * 1e3, R * W * H * steps / dt / 1e6), flush=True)
