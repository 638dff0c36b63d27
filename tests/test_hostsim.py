"""CPU tests of the *device* functions: tests/hostsim compiles j40_amd/csrc/device/*_dev.h for the CPU
and runs them with the kernels' orchestration; results must equal the reference bit for bit
(quantised coefficients) and within 1 u8 level (RGBA; in practice identical)."""
import ctypes as C
import os

import numpy as np
import pytest

from streams import synth, VARDCT_CASES, MODULAR_CASES, ROOT


@pytest.fixture(scope="module")
def sim(built):
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    return S


@pytest.mark.parametrize("name,opts", VARDCT_CASES[:8] + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))])
def test_device_functions_on_cpu_match_reference(ref, sim, name, opts):
    from refdec import RefStage
    w, h = (776, 520) if name == "all_transforms" else (392, 264)
    data = synth("vardct", w, h, 31, **opts)
    rs = RefStage(ref, data)
    ncells = sum(rs.lf_group_info(g)["width8"] * rs.lf_group_info(g)["height8"] for g in range(rs.info["num_lf_groups"]))
    rgba = np.zeros((h, w, 4), np.uint8)
    co = np.zeros((3, ncells * 64), np.float32)
    buf = C.create_string_buffer(data, len(data))
    assert sim.hostsim_decode(buf, len(data), rgba.ctypes.data, co.ctypes.data, 0) == 0
    base = 0
    for g in range(rs.info["num_lf_groups"]):
        n = rs.lf_group_info(g)["width8"] * rs.lf_group_info(g)["height8"] * 64
        for c in range(3):
            assert np.array_equal(rs.coeffs(g, c), co[c, base:base + n]), "quantised HF coefficients differ"
        base += n
    assert rs.combine() == ""
    d = np.abs(rs.rgba().astype(np.int32) - rgba.astype(np.int32))
    assert d.max() <= 1
    rs.close()


@pytest.mark.parametrize("name,w,h,opts", MODULAR_CASES)
def test_modular_device_functions_on_cpu_are_bit_exact(ref, sim, name, w, h, opts):
    data = synth("modular", w, h, 61, **opts)
    err, expect = ref.decode(data)
    assert err == ""
    rgba = np.zeros(expect.shape, np.uint8)
    buf = C.create_string_buffer(data, len(data))
    assert sim.hostsim_decode(buf, len(data), rgba.ctypes.data, None, 0) == 0
    assert np.array_equal(rgba, expect)
