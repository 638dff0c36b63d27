"""tools/lf_rows_sweep.py [n] [seed] -- k_lf_rows' decoder (device/lf_rows_dev.h compiled for the CPU, leaf-only channels left as residuals
and predicted afterwards, as the kernels run) against the host decoder on damaged streams: random sizes, LF trees and single bit flips
inside the LfGroup sections; status codes and planes must agree (hostsim_lf_rows_check). CPU only."""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from streams import SYNTH
import subprocess
import tempfile


def synth(mode, w, h, seed, **opts):
    """like streams.synth, without leaving the stream in the on-disk cache (hundreds of them would travel to the GPU box)"""
    with tempfile.NamedTemporaryFile(suffix=".jxl") as tmp:
        subprocess.run([SYNTH, mode, str(w), str(h), str(seed), tmp.name] + ["%s=%s" % kv for kv in sorted(opts.items())], check=True, stderr=subprocess.DEVNULL)
        return open(tmp.name, "rb").read()


S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
S.hostsim_lf_rows_check.restype = C.c_int32
S.hostsim_lf_rows_check.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
S.hostsim_lf_rows_counts.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.c_int32]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = failed = skipped = 0
for i in range(n):
    w, h = r.choice([(520, 264), (1300, 776), (2049, 300), (2600, 2100), (300, 2200)])
    opts = {}
    t = r.choice([0, 0, 1, 2, 3])
    if t: opts["lftree"] = t
    if r.random() < .3: opts["forward"] = 1
    elif r.random() < .3: opts["cfl"] = 1   # (the generator's forward encoder writes no chroma-from-luma maps)
    d = bytearray(synth("vardct", w, h, 400 + (i % 7), **opts))
    flips = r.choice([0, 1, 1, 1, 2])
    for _ in range(flips):
        d[r.randrange(120, max(121, len(d) // r.choice([3, 6, 12])))] ^= 1 << r.randrange(8)
    S.hostsim_lf_rows_counts(None, None, 1, 0)
    buf = C.create_string_buffer(bytes(d), len(d))
    ns, bad = C.c_int32(), C.c_int32()
    rc = S.hostsim_lf_rows_check(buf, len(d), r.choice([1, 4, 64]), C.byref(ns), C.byref(bad))
    if rc == -1: skipped += 1; continue
    cases += 1; failed += 1 if bad.value else 0
    if rc != 0:
        print("MISMATCH rc=%d case %d %dx%d %s flips=%d" % (rc, i, w, h, opts, flips)); sys.exit(1)
S.hostsim_lf_rows_deferred_sections.restype = C.c_int64
print("%d cases decoded by both (%d with a failing section: same codes, or -- %d sections -- a run of straight-line steps ran into the error, the lane said 'lffb' and the host's decoder did report one), %d streams the front end refused; 0 mismatches"
      % (cases, failed, S.hostsim_lf_rows_deferred_sections(), skipped))
