#!/usr/bin/env python3
"""random VarDCT option mixes x sizes x seeds, a share of them with one flipped bit: the pipeline's device-side stages compiled for
the CPU (build/libhostsim.so) against the host path -- hostsim_device_plan_check (varblock placement, counting, K1 / K2 records,
verdict: plan_dev.h vs plan_build.cpp) and hostsim_lf_lanes_check (the lane decoder of the LfGroup streams vs the host decoder).
CPU only.  python tools/fuzz_device_plan.py [n] [seed]"""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from fuzz_parity import synth, pick_vardct

S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
S.hostsim_device_plan_check.restype = C.c_int32
S.hostsim_device_plan_check.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
S.hostsim_lf_lanes_check.restype = C.c_int32
S.hostsim_lf_lanes_check.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = checked = skipped = sections = failed_sections = 0
for k in range(n):
    opts = pick_vardct(r)
    opts.pop("container", None)
    if r.random() < .3:
        opts = {"forward": 1}
    w, h = r.choice([(520, 264), (776, 520), (1300, 776), (2100, 1100), (2600, 2100)])
    seed = r.randrange(1 << 20)
    try:
        data = synth("vardct", w, h, seed, **opts)
    except Exception:
        continue
    flips = 0
    if r.random() < .4:
        m = bytearray(data)
        for _ in range(r.choice([1, 1, 2])):
            m[r.randrange(40, max(41, len(m) // 4))] ^= 1 << r.randrange(8)
        data, flips = bytes(m), 1
    buf = C.create_string_buffer(data, len(data))
    err = C.c_uint32()
    rc = S.hostsim_device_plan_check(buf, len(data), C.byref(err))
    ns, nf = C.c_int32(), C.c_int32()
    rl = S.hostsim_lf_lanes_check(buf, len(data), C.byref(ns), C.byref(nf))
    if rc == -1 and rl == -1:
        skipped += 1   # frames the pipeline's batched path does not take (or whose front does not parse)
        continue
    checked += 1
    sections += ns.value; failed_sections += nf.value
    if rc not in (0, -1) or rl not in (0, -1):
        bad += 1
        print("MISMATCH plan rc=%d err=%08x lanes rc=%d: %dx%d seed %d %s flips=%d" % (rc, err.value, rl, w, h, seed, opts, flips), flush=True)
print("%d streams checked, %d not taken by the batched path, %d mismatches; %d LfGroup sections through the lane decoder, %d of them failing like the host's" % (checked, skipped, bad, sections, failed_sections))
sys.exit(1 if bad else 0)
