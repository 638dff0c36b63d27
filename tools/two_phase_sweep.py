"""tools/two_phase_sweep.py [n] [seed] -- j40hip_frame_decode_to_host in two phases (device/runtime.hip: the longest pass-group sections on a stream
of their own, the image over the link while they finish, their groups' rectangles on top) against the same call in one phase (J40HIP_TWO_PHASE=0):
random VarDCT streams large enough for the two phases (>= 64 groups, >= 16 MB of pixels), the generator's options at random, a bit flipped
somewhere behind the headers in two thirds of them. Same code, same pixels. Needs an MI355X."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import j40_amd
from streams import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
used = errors = refused = 0
for i in range(n):
    w, h = r.choice([(2600, 2100), (4096, 2304), (2304, 2304), (5000, 1100), (3000, 1800)])
    opts = {}
    if r.random() < .5: opts["forward"] = 1
    elif r.random() < .4: opts["cfl"] = 1
    if r.random() < .2: opts["lftree"] = r.choice([1, 2, 3])
    if not opts.get("forward") and r.random() < .2: opts["maxlog"] = r.choice([6, 7, 8])
    d = bytearray(synth("vardct", w, h, 700 + (i % 5), **opts))
    flips = r.choice([0, 1, 1])
    for _ in range(flips):
        d[r.randrange(len(d) // 4, len(d) - 8)] ^= 1 << r.randrange(8)
    d = bytes(d)
    out = {}
    for two in ("1", "0"):
        os.environ["J40HIP_TWO_PHASE"] = two
        try:
            fr = j40_amd.Frame(d, threads=6)
        except j40_amd.J40Error as e:
            out[two] = (e.code, None, -2)
            continue
        fr.upload(0)
        code, px = fr.decode_to_host()
        out[two] = (code, px, fr.two_phase_sections())
        fr.close()
    a, b = out["1"], out["0"]
    if a[2] == -2: refused += 1
    if a[0] != b[0] or (a[0] == "" and not np.array_equal(a[1], b[1])) or b[2] > 0:
        print("MISMATCH case %d %dx%d %s flips=%d: two phases (%d sections) %r, one phase %r" % (i, w, h, opts, flips, a[2], a[0], b[0])); sys.exit(1)
    used += 1 if a[2] > 0 else 0
    errors += 1 if a[0] else 0
print("%d streams (%d decoded in two phases, %d ending with an error code, %d of those refused by the host parse): two phases and one agree on every code and every pixel; 0 mismatches" % (n, used, errors, refused))
j40_amd.shutdown()
