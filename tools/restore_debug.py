"""debug: where the XYB planes the pixel kernels leave differ from the oracle's samples, by DctSelect (MEASUREMENT / DEBUG TOOL)"""
import ctypes as C, os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import j40_amd
from streams import synth
D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
D.oracle_run_xyb.restype = C.c_uint32; D.oracle_run_xyb.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
w, h = 776, 520
data = synth("vardct", w, h, 31, fullheader=1, gab=2, epf=3, epfw=1, epfs=1, maxlog=8, bctx=1)
f = j40_amd.Frame(data); f.upload(0); f.set_restoration(1)
err, rgba = f.decode_to_host(); print("err", repr(err))
x0 = f.read_xyb(0)
want = np.zeros((3, h, w), np.float32); buf = C.create_string_buffer(data, len(data))
print(D.oracle_run_xyb(buf, len(data), want.ctypes.data))
m = x0.view(np.uint32) != want.view(np.uint32)
print("differing samples per channel", m.sum(axis=(1, 2)))
blocks = f.plane(0, 0)
# dctsel of the varblock covering each cell: walk top-left cells
sel = np.zeros_like(blocks)
DCT = {0:(8,8),1:(8,8),2:(8,8),3:(8,8),4:(16,16),5:(32,32),6:(16,8),7:(8,16),8:(32,8),9:(8,32),10:(32,16),11:(16,32),12:(8,8),13:(8,8),14:(8,8),15:(8,8),16:(8,8),17:(8,8),18:(64,64),19:(64,32),20:(32,64),21:(128,128),22:(128,64),23:(64,128),24:(256,256),25:(256,128),26:(128,256)}
H8, W8 = blocks.shape
for y in range(H8):
    for x in range(W8):
        d = blocks[y, x] >> 20
        if d >= 2:
            r, c = DCT[d - 2]
            sel[y:y + r // 8, x:x + c // 8] = d - 2
cnt = collections.Counter(); tot = collections.Counter()
for y in range(H8):
    for x in range(W8):
        tot[int(sel[y, x])] += 1
        if m[:, y * 8:y * 8 + 8, x * 8:x * 8 + 8].any(): cnt[int(sel[y, x])] += 1
print("cells with a difference, by DctSelect:", dict(cnt)); print("cells by DctSelect:", dict(tot))
ys, xs = np.nonzero(m.any(axis=0)); print("first positions", list(zip(ys[:10].tolist(), xs[:10].tolist())))
if len(ys):
    y, x = ys[0], xs[0]; print(x0[:, y, x], want[:, y, x])
