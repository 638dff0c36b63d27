"""The host's side of decoding LfGroup streams on the device (frame.cpp: the lf_decoder branch of parse_frame, lf_group_finish),
exercised on the CPU with a stand-in decoder (tests/hostsim: it does on the host what k_lf_groups does on the device and hands back
planes in the kernel's layout). The frame parsed through it must equal the plain parse -- block maps, LF index, LF integers,
chroma-from-luma maps, varblocks -- with every section taken by the stand-in, with every second section sent back to the host
('lffb'), and on damaged streams (same error code). The kernel itself is checked on the GPU (tests/test_pipeline.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from streams import synth, ROOT

CASES = [
    ("vardct", 776, 520, 31, dict()),
    ("vardct", 2600, 2100, 32, dict(bctx=1)),
    ("vardct", 2049, 300, 33, dict(maxlog=8, cfl=1)),
    ("vardct", 1920, 1080, 34, dict(forward=1)),
    ("vardct", 520, 264, 35, dict(alpha=1)),
    ("vardct", 520, 264, 36, dict(passes=3)),
    ("modular", 600, 300, 37, dict(tree=1)),          # never reaches the decoder
]


@pytest.fixture(scope="module")
def sim(built):
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_lf_decoder_glue.restype = C.c_int32
    S.hostsim_lf_decoder_glue.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32)]
    return S


@pytest.mark.parametrize("mode,w,h,seed,opts", CASES)
@pytest.mark.parametrize("fallback", [0, 1])
def test_frame_parsed_through_an_external_lf_decoder_equals_the_plain_parse(sim, mode, w, h, seed, opts, fallback):
    data = synth(mode, w, h, seed, **opts)
    buf = C.create_string_buffer(data, len(data))
    err = C.c_uint32()
    assert sim.hostsim_lf_decoder_glue(buf, len(data), fallback, C.byref(err)) == 0
    assert err.value == 0


def test_damaged_lf_sections_end_the_same_way_through_an_external_decoder(sim):
    data = synth("vardct", 2600, 2100, 41)
    rng = np.random.default_rng(8)
    codes = set()
    for _ in range(60):
        m = bytearray(data)
        m[int(rng.integers(150, len(m) // 6))] ^= 1 << int(rng.integers(0, 8))
        buf = C.create_string_buffer(bytes(m), len(m))
        err = C.c_uint32()
        for fallback in (0, 1):
            assert sim.hostsim_lf_decoder_glue(buf, len(m), fallback, C.byref(err)) == 0
        codes.add(err.value)
    assert len(codes) >= 3, "the flips should have produced several different outcomes"
