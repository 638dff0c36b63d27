"""tools/lf_rows_schedule.py -- what a wavefront of k_lf_rows goes through for the bench's 8K frame (CPU: tests/hostsim runs the kernel's
schedule with the device functions): plain iterations, general-step rounds, by combination of needs. MEASUREMENT AID."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from streams import synth

S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
S.hostsim_lf_rows_schedule.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.POINTER(C.c_int64)]
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
for copies in (-1, 1):   # every channel predicted by its lane / leaf-only channels left as residuals (the default)
    d = synth("vardct", W, H, 1, forward=1)
    out = (C.c_int64 * 36)()
    n = S.hostsim_lf_rows_schedule(d, len(d), copies, out)
    print("lanes", n, "plain iterations", out[0], "general rounds", out[1], "longest lane's samples", out[2], "all lanes' samples", out[3])
    print("  plain iterations by need:", {i: out[4 + i] for i in range(32) if out[4 + i]})
