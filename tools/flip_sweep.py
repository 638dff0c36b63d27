#!/usr/bin/env python3
"""every bit of the first N bytes of a few small streams flipped in turn: the CPU build of the device headers (build/libhostsim.so)
against the reference (oracle/_ref), verdict and error code. The reference runs in a forked child with an alarm: some flips crash
it or make it loop.  python tools/flip_sweep.py [nbytes]   (CPU only, about a minute per thousand flips)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from refdec import Ref
from fuzz_parity import synth, decode_in_child

CASES = [("modular", 50, 40, 1, dict(container=1)), ("vardct", 264, 136, 2, dict(container=2)), ("modular", 64, 48, 3, dict(icc=60)),
         ("vardct", 264, 136, 5, dict(icc=50)), ("modular", 70, 50, 6, dict(extra=3)), ("modular", 60, 40, 7, dict(xyb=1)),
         ("modular", 60, 40, 8, dict(ycbcr=1, alpha=1)), ("vardct", 264, 136, 9, dict(bpp=10)), ("modular", 300, 140, 10, dict(groupshift=7, permute=1)),
         ("vardct", 264, 136, 11, dict(alpha=1, fullheader=1, noxyb=1)), ("modular", 300, 260, 5, dict(passes=2)), ("vardct", 520, 264, 3, dict(passes=2, permute=1))]


def main():
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 130
    ref = Ref()
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    total = bad = crashed = 0
    for mode, w, h, seed, o in CASES:
        d = synth(mode, w, h, seed, **o)
        for byte in range(min(len(d), nbytes)):
            for bit in range(8):
                b = bytearray(d); b[byte] ^= 1 << bit; b = bytes(b)
                e, px = decode_in_child(ref, b, 10)
                if e == "CRSH":
                    crashed += 1; print("reference crashed or hung:", mode, seed, o, byte, bit, flush=True)
                    continue
                if e == "" and px.shape[:2] != (h, w): continue   # a flip in the size header
                out = np.zeros((h, w, 4), np.uint8)
                buf = C.create_string_buffer(b, len(b))
                code = S.hostsim_decode(buf, len(b), out.ctypes.data, None, 0)
                mine = "" if code == 0 else code.to_bytes(4, "big").decode("latin1")
                total += 1
                if mine != e or (e == "" and np.abs(px.astype(int) - out).max() > (0 if mode == "modular" else 1)):
                    bad += 1; print("MISMATCH", mode, w, h, seed, o, byte, bit, repr(e), repr(mine), flush=True)
    print("%d flips, %d where the reference crashed or hung, %d mismatches" % (total, crashed, bad))


if __name__ == "__main__":
    main()
