"""quick throughput probe of the pipeline: N 8K frames (D distinct streams) through j40hip_pipeline_*; prints Mpx/s for device and host output"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, j40_amd
from streams import synth
W, H = 7680, 4320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
D = int(sys.argv[4]) if len(sys.argv) > 4 else 8
datas = [synth("vardct", W, H, 1000 + i) for i in range(D)]
import ctypes as C
bufs = [C.create_string_buffer(d, len(d)) for d in datas]
for dev_out in (True, False):
    pipe = j40_amd.Pipeline(0, threads, batch, 2)
    outs = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") if dev_out else torch.empty((H, W, 4), dtype=torch.uint8).pin_memory() for _ in range(min(N, 3 * batch))]
    for rep in range(2):
        pipe.reset_stats()
        t0 = time.time()
        ts = [pipe.submit(bufs[i % D], outs[i % len(outs)].data_ptr(), W * 4, device_output=dev_out) for i in range(N)]
        pipe.drain(); torch.cuda.synchronize()
        dt = time.time() - t0
        st = pipe.stats()
        bad = [pipe.result(t) for t in ts if pipe.result(t)]
        print("device_output=%s rep %d: %d frames in %.3f s = %.0f Mpx/s; parse %.1f ms/frame, plan+upload %.1f ms/frame (thread time); errors %s" % (dev_out, rep, N, dt, N * W * H / dt / 1e6, st["parse_thread_ms"] / N, st["upload_thread_ms"] / N, bad[:3]), flush=True)
    pipe.close()
