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
