"""times the pipeline's host stage on this CPU (tests/hostsim build of the host sources): front parse, front plan, LfGroup streams"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from streams import synth, ROOT
S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
S.hostsim_front_timing.restype = C.c_int32
S.hostsim_front_timing.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_double)]
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
data = synth("vardct", w, h, 3, forward=1)
buf = C.create_string_buffer(data, len(data))
out = (C.c_double * 4)()
assert S.hostsim_front_timing(buf, len(data), 5, out) == 0
print("front parse %.2f ms, key + front plan %.2f ms, LfGroup streams %.2f ms, full parse_frame %.2f ms (%d bytes)" % (out[0], out[1], out[2], out[3], len(data)))
