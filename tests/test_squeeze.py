"""Inverse Squeeze (ISO 18181-1). The reference parses the parameters and stops with "TODO" (j40.h:3794-3812, 4518), so it cannot
be the oracle for the transform itself: PARITY UNPINNED against libjxl. What pins it here:

  * lossless round trip: tools/jxlsynth applies an independently written forward Squeeze to a picture, the decoder must return
    that picture;
  * the picture itself is pinned by the REFERENCE: the same picture coded without Squeeze (RCT only) decodes through the
    unmodified reference to the expected pixels, bit for bit;
  * three implementations of the inverse step -- the HIP kernels, the device functions compiled for the CPU (tests/hostsim) and
    the plain-C oracle (oracle/hotpath_oracle.c) -- agree;
  * the reference does report "TODO" on every one of these streams (so this is a superset of its behaviour, not a change).

Config 4 of BASELINE.json (16384 x 16384 Modular lossless, Squeeze + RCT) is therefore reported twice: RCT-only (reference-pinned,
tests/test_gpu_parity.py) and Squeeze + RCT (round-trip-pinned, here)."""
import ctypes as C
import os

import numpy as np
import pytest

from streams import synth, ROOT

# (width, height, generator options); squeeze=1 default parameter list, 2 the same list written out, 3 a short explicit list with
# appended (not in place) residual channels and partial channel ranges
CASES = [
    (200, 150, dict()),                                   # one group: everything inside LfGlobal
    (200, 150, dict(alpha=1, tree=1)),
    (600, 300, dict()),                                   # six groups: LfGlobal (small channels) + pass groups
    (600, 300, dict(alpha=1, tree=2)),                    # weighted predictor
    (601, 299, dict(tree=1, alpha=1)),                    # odd sizes: averages one longer than residuals, clipped rectangles
    (520, 520, dict(groupshift=7, prefix=1, lz77=1)),     # 128-pixel groups, prefix codes + LZ77
    (2600, 2100, dict(tree=1)),                           # 2 x 2 LfGroups: channels shifted by >= 3 are coded there
    (300, 200, dict(bpp=12, rct=13)),
]


def _hostsim():
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    return S


def _oracle():
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    D.oracle_run.restype = C.c_uint32
    D.oracle_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    return D


@pytest.mark.parametrize("sq", [1, 2, 3])
@pytest.mark.parametrize("w,h,opts", CASES)
def test_squeeze_round_trip_on_the_cpu_checkers(built, ref, w, h, opts, sq):
    plain = synth("modular", w, h, 7, **opts)
    squeezed = synth("modular", w, h, 7, squeeze=sq, **opts)
    rerr, expect = ref.decode(plain)
    assert rerr == "", "the picture coded without Squeeze is what the reference pins"
    assert ref.decode(squeezed)[0] == "TODO", "the reference stops at the Squeeze parameters (j40.h:3812)"
    buf = C.create_string_buffer(squeezed, len(squeezed))
    a = np.zeros((h, w, 4), np.uint8)
    assert _hostsim().hostsim_decode(buf, len(squeezed), a.ctypes.data, None, 0) == 0
    assert np.array_equal(a, expect), "device functions on the CPU: lossless round trip"
    b = np.zeros((h, w, 4), np.uint8)
    assert _oracle().oracle_run(buf, len(squeezed), b.ctypes.data, None) == 0
    assert np.array_equal(b, expect), "plain-C oracle: lossless round trip"


def test_default_parameter_list_equals_its_explicit_form(built):
    """squeeze=1 sends no parameters (the decoder derives the default list), squeeze=2 sends that list explicitly:
    both must decode to the same pixels"""
    S = _hostsim()
    outs = []
    for sq in (1, 2):
        d = synth("modular", 777, 333, 19, squeeze=sq, tree=1)
        o = np.zeros((333, 777, 4), np.uint8)
        assert S.hostsim_decode(C.create_string_buffer(d, len(d)), len(d), o.ctypes.data, None, 0) == 0
        outs.append(o)
    assert np.array_equal(outs[0], outs[1])


def test_truncated_squeeze_parameters_still_report_like_the_reference(built, ref):
    """running out of bytes inside the parameter list wins over everything else, in the reference as here (j40.h:3794-3811)"""
    S = _hostsim()
    d = synth("modular", 200, 150, 7, squeeze=3)
    for cut in range(14, 40):
        b = d[:cut]
        rerr = ref.decode(b)[0]
        o = np.zeros((150, 200, 4), np.uint8)
        code = S.hostsim_decode(C.create_string_buffer(b, len(b)), len(b), o.ctypes.data, None, 0)
        err = "" if code == 0 else code.to_bytes(4, "big").decode("latin1")
        if rerr != "TODO":
            assert err == rerr, (cut, rerr, err)
        else:
            assert err != "", cut   # truncated behind the parameters: some section must come up short


@pytest.mark.gpu
@pytest.mark.parametrize("sq", [1, 3])
@pytest.mark.parametrize("w,h,opts", CASES)
def test_squeeze_round_trip_on_the_gpu(built, ref, w, h, opts, sq):
    import j40_amd
    plain = synth("modular", w, h, 7, **opts)
    squeezed = synth("modular", w, h, 7, squeeze=sq, **opts)
    rerr, expect = ref.decode(plain)
    assert rerr == ""
    err, rgba = j40_amd.decode(squeezed)
    assert err == "" and np.array_equal(rgba, expect)


@pytest.mark.gpu
def test_config4_16384_squeeze_and_rct_round_trip(built, ref):
    """BASELINE.json config 4, variant B: 16384 x 16384 Modular lossless with RCT + Squeeze (default parameter list: 4 + 66
    channels, 64 LfGroup sections, 4096 pass-group sections). The picture is a 1024 x 1024 tile repeated 16 x 16; the tile is
    pinned by the reference's decode of the tile coded RCT-only; Squeeze couples neighbouring groups, so the whole frame went
    through the forward transform (tools/jxlsynth) and must come back as the tiled picture."""
    import torch
    import j40_amd
    base = synth("modular", 1024, 1024, 21, tree=1)
    rerr, tile = ref.decode(base)
    assert rerr == ""
    data = synth("modular", 16384, 16384, 21, tree=1, repeat=16, squeeze=1)
    fr = j40_amd.Frame(data)
    assert (fr.width, fr.height) == (16384, 16384)
    fr.upload(0)
    out = torch.zeros((16384, 16384, 4), dtype=torch.uint8, device="cuda:0")
    ms = fr.decode_timed(out.data_ptr(), 16384 * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert fr.status() == ""
    print("16384x16384 Modular, RCT + Squeeze: sections %.1f ms, inverse transforms + pack %.1f ms" % (ms[0], ms[1]))
    t = torch.from_numpy(tile).to("cuda:0")
    assert bool((out.view(16, 1024, 16, 1024, 4) == t.view(1, 1024, 1, 1024, 4)).all())
    fr.close()


# ---- known answers, worked out by hand from the standard's formulas (ISO 18181-1, Squeeze: tendency and the pair rule) ----
# PARITY UNPINNED against libjxl all the same: these pin the arithmetic to the text of the standard as read here, the round trips
# above pin the transform as a whole to the generator's independent forward step.
TENDENCY_KAT = [
    # (B, a, n) -> X
    ((10, 10, 10), 0),     # (40 - 30 - 10 + 6) / 12 = 0
    ((20, 14, 2), 5),      # 66 / 12 = 5; 5 - 1 = 4 <= 12; 5 + 1 = 6 <= 24
    ((20, 19, 0), 3),      # 67 / 12 = 5; 5 - 1 = 4 > 2 (B - a) = 2 -> 2 * 1 + 1 = 3; 3 + 1 <= 38
    ((40, 12, 11), 2),     # 121 / 12 = 10; 10 <= 56; 10 > 2 (a - n) = 2 -> 2
    ((10, 10, 4), 1),      # 24 / 12 = 2; 2 > 0 -> 2 * 0 + 1 = 1; 1 + 1 <= 12
    ((2, 14, 20), -6),     # rising: (8 - 60 - 14 - 6) / 12 = -6; -6 >= -24; -6 >= -12
    ((0, 1, 20), -3),      # -67 / 12 = -5 (towards zero); -5 + 1 = -4 < 2 (B - a) = -2 -> -3; -3 - 1 = -4 >= -38
    ((0, 19, 20), -2),     # -85 / 12 = -7; -7 + 1 = -6 >= -38; -7 - 1 = -8 < 2 (a - n) = -2 -> -2
    ((10, 10, 14), -1),    # rising with B = a: (40 - 42 - 10 - 6) / 12 = -18 / 12 = -1; -1 + 1 = 0 >= 0; -1 - 1 = -2 >= -8
    ((5, 9, 3), 0),        # not monotone
    ((9, 5, 8), 0),
]
LINE_KAT = [
    # (averages, residuals) -> samples
    (([10], [0]), [10, 10]),
    (([10], [3]), [11, 8]),                         # diff 3: A = 10 + 1; forward check: (11 + 8 + 1) >> 1 = 10, 11 - 8 = 3
    (([10], [-3]), [9, 12]),                        # diff -3: A = 10 + (-3 / 2 = -1)
    (([10, 14, 2], [3, 1, -2]), [11, 9, 14, 13, 1, 3]),
    #   k = 0: left = 10, next = 14: tendency -1, diff 2, A = 11 -> 11, 9
    #   k = 1: left = 9, a = 14, next = 2: not monotone, diff 1, A = 14 -> 14, 13
    #   k = 2: left = 13, a = 2, next = 2 (last): 50 / 12 = 4 > 2 (a - n) = 0 -> 0; diff -2, A = 2 - 1 -> 1, 3
    (([10, 14, 2], [3, 1]), [11, 9, 14, 13, 2]),   # one residual fewer: the last average passes through
    (([32767], [2]), [-32768, 32766]),             # 16-bit buffers wrap (j40.h:3169)
    (([14], [1]), [14, 13]),
]


def test_squeeze_known_answers_from_the_standards_formulas(built):
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    for lib, prefix in ((S, "hostsim_"), (D, "oracle_kat_")):
        tend = getattr(lib, prefix + "squeeze_tendency")
        tend.restype = C.c_int32
        tend.argtypes = [C.c_int32] * 3
        line = getattr(lib, prefix + "unsqueeze_line")
        line.restype = None
        line.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        for (B, a, n), want in TENDENCY_KAT:
            assert tend(B, a, n) == want, (prefix, B, a, n)
        for (avg, res), want in LINE_KAT:
            av, rs = np.array(avg, dtype=np.int16), np.array(res + [0], dtype=np.int16)
            out = np.zeros(len(avg) + len(res), dtype=np.int16)
            line(av.ctypes.data, len(avg), rs.ctypes.data, len(res), out.ctypes.data)
            assert out.tolist() == want, (prefix, avg, res)
    # the pair rule is the forward rule's inverse: avg = (A + B + (A > B)) >> 1, residual = A - B - tendency
    tend = S.hostsim_squeeze_tendency
    line = S.hostsim_unsqueeze_line
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(2, 40))
        px = rng.integers(-300, 300, size=n).astype(np.int64)
        n_avg, n_res = (n + 1) // 2, n // 2
        avg = [int((px[2 * k] + px[2 * k + 1] + (1 if px[2 * k] > px[2 * k + 1] else 0)) >> 1) if 2 * k + 1 < n else int(px[2 * k]) for k in range(n_avg)]
        res = []
        for k in range(n_res):
            left = int(px[2 * k - 1]) if k > 0 else avg[k]
            nxt = avg[k + 1] if k + 1 < n_avg else avg[k]
            res.append(int(px[2 * k] - px[2 * k + 1]) - tend(left, avg[k], nxt))
        av, rs = np.array(avg, dtype=np.int16), np.array(res + [0], dtype=np.int16)
        out = np.zeros(n, dtype=np.int16)
        line(av.ctypes.data, n_avg, rs.ctypes.data, n_res, out.ctypes.data)
        assert out.tolist() == px.tolist()
