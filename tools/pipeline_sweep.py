"""tools/pipeline_sweep.py [n] [seed] -- random VarDCT streams (the generator's four LF trees, sizes with slivers of LfGroups, a bit flipped
in two thirds of them) through the throughput pipeline with the LfGroup streams on the device (k_lf_rows + k_lf_predict, plan build,
entropy lanes, pixel kernels) against the single-image path (host parse, latency kernels): same codes, same pixels. Needs an MI355X."""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import j40_amd
from streams import SYNTH


def synth(mode, w, h, seed, **opts):
    with tempfile.NamedTemporaryFile(suffix=".jxl") as tmp:
        subprocess.run([SYNTH, mode, str(w), str(h), str(seed), tmp.name] + ["%s=%s" % kv for kv in sorted(opts.items())], check=True, stderr=subprocess.DEVNULL)
        return open(tmp.name, "rb").read()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = []
for i in range(n):
    w, h = r.choice([(520, 264), (1300, 776), (2049, 300), (2600, 2100), (300, 2200), (1920, 1080)])
    opts = {}
    t = r.choice([0, 0, 1, 2, 3])
    if t: opts["lftree"] = t
    if r.random() < .3: opts["forward"] = 1
    elif r.random() < .3: opts["cfl"] = 1
    d = bytearray(synth("vardct", w, h, 500 + (i % 9), **opts))
    flips = r.choice([0, 1, 1, 2])
    for _ in range(flips):
        d[r.randrange(120, max(121, len(d) // r.choice([3, 6, 12])))] ^= 1 << r.randrange(8)
    cases.append((w, h, bytes(d), opts, flips))
pipe = j40_amd.Pipeline(device=0, host_threads=6, batch_frames=16, max_in_flight=2, lf_streams="device")
outs, tickets = [], []
for w, h, d, _, _ in cases:
    o = torch.zeros((h, w, 4), dtype=torch.uint8).pin_memory()
    outs.append(o); tickets.append(pipe.submit(d, o.data_ptr(), w * 4))
pipe.drain()
failed = bad = 0
for (w, h, d, opts, flips), o, t in zip(cases, outs, tickets):
    try:
        err, expect = j40_amd.decode(d)
    except Exception as e:   # (a damaged header the front end refuses with an exception: both paths see the same bytes)
        err, expect = str(e)[-4:], None
    got = pipe.result(t)
    if got != err and not (expect is None and got):
        print("MISMATCH codes %r vs %r: %dx%d %s flips=%d" % (got, err, w, h, opts, flips)); bad += 1; continue
    if err == "":
        if not np.array_equal(o.numpy(), expect):
            print("MISMATCH pixels: %dx%d %s flips=%d" % (w, h, opts, flips)); bad += 1
    else:
        failed += 1
st = pipe.stats()
pipe.close()
print("%d streams (%d end with an error code, the same on both paths; LfGroup streams of %d frames on the device), %d mismatches" % (len(cases), failed, st.get("lf_device_frames", -1), bad))
sys.exit(1 if bad else 0)
