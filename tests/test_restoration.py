"""The restoration filters: Gaborish and the edge-preserving filter.

The reference defines j40__gaborish (j40.h:7271) and j40__epf (j40.h:7578) and never calls them; oracle/ref_harness.c makes them
callable on caller-supplied planes. What is pinned here, and against what:

  * oracle/hotpath_oracle.c's restatement (oracle_gaborish / oracle_epf / oracle_epf_step) == the reference's routines, bit for bit, all
    three channels, every step, sizes from 2x1 up -- with `quirk` = 1, which restates what j40__epf_step actually reads through its
    aliased line buffers. (That routine writes outside its buffer and only runs under an allocator with slack: the zero-filling build
    oracle/_ref/libj40ref_zalloc.so.) The error codes too ("gab0", "epf0", "shrp").
  * the device functions (j40_amd/csrc/device/restore_dev.h) compiled for the CPU (tests/hostsim) == the restatement, both modes.
  * -m gpu: the HIP kernels == the reference's routines (mode 2) and == the restatement (mode 1), bit for bit, on random planes and on the
    XYB planes of decoded streams (8x8-only, mixed transforms, 8K); the frame header's fields and the sharpness map == the reference's
    parse; the XYB planes the pixel kernels leave == the oracle's samples; the final pixels == the oracle's colour conversion of the
    filtered planes; and with the filters off (the default) the decode is the reference's.
"""
import ctypes as C
import os

import numpy as np
import pytest

from streams import synth, ROOT

PARAMS24 = [1, 0.115169525, 0.061248592, 0.2, 0.05, 0.1, 0.08, 2, 0.3, 0.5, 0.7, 1.0, 1.3, 1.6, 2.0, 2.5, 40, 5, 3.5, 0.46, 0.9, 6.5, 2 / 3, 1.0]
SIZES = [(8, 8), (17, 9), (64, 40), (2, 2), (3, 1), (250, 131), (33, 70), (9, 2), (5, 3), (2, 9)]


def e4(c):
    return "".join(chr((c >> s) & 255) for s in (24, 16, 8, 0)) if c else ""


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def libs(built):
    zal = os.path.join(ROOT, "oracle", "_ref", "libj40ref_zalloc.so")
    if not os.path.exists(zal):
        pytest.skip("oracle/_ref/libj40ref_zalloc.so is not available")
    R = C.CDLL(zal); O = C.CDLL(os.path.join(ROOT, "oracle", "libj40oracle.so")); S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so")); D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    vp, i32 = C.c_void_p, C.c_int32
    R.ref_kat_gaborish.restype = C.c_uint32; R.ref_kat_gaborish.argtypes = [vp] * 3 + [i32] * 2 + [vp]
    R.ref_kat_epf.restype = C.c_uint32; R.ref_kat_epf.argtypes = [vp] * 3 + [i32] * 2 + [vp, vp, i32, vp, vp]
    R.ref_kat_epf_step.restype = C.c_uint32; R.ref_kat_epf_step.argtypes = [vp] * 3 + [i32] * 2 + [vp, i32, vp]
    O.oracle_gaborish.restype = C.c_uint32; O.oracle_gaborish.argtypes = [vp] * 3 + [i32] * 2 + [vp]
    O.oracle_epf.restype = C.c_uint32; O.oracle_epf.argtypes = [vp] * 3 + [i32] * 2 + [vp, vp, i32, vp, vp, i32]
    O.oracle_epf_step.restype = C.c_uint32; O.oracle_epf_step.argtypes = [vp] * 3 + [i32] * 2 + [vp, i32, vp, i32]
    S.hostsim_restoration.restype = C.c_uint32; S.hostsim_restoration.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    D.oracle_run_xyb.restype = C.c_uint32; D.oracle_run_xyb.argtypes = [vp, C.c_size_t, vp]
    D.oracle_colour.restype = C.c_uint32; D.oracle_colour.argtypes = [vp, C.c_size_t, vp, vp]
    return R, O, S, D


def random_case(rng, w, h):
    planes = np.stack([(rng.standard_normal((h, w)) * s).astype(np.float32) for s in (0.02, 0.5, 0.3)])
    w8, h8 = (w + 7) // 8, (h + 7) // 8
    sharp = rng.integers(0, 8, (h8, w8)).astype(np.int16)
    hf = (1.0 / rng.integers(1, 12, (h8, w8))).astype(np.float32)
    return planes, sharp, hf


def run3(fn, planes, *args):
    a = np.ascontiguousarray(planes, np.float32).copy()
    code = fn(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a.shape[2], a.shape[1], *args)
    return code, a


def filtered_by(O, planes, sharp, hf, p24, quirk):
    """the restatement: Gaborish when signalled, then the edge-preserving filter; (code, planes, sigma)"""
    p24 = np.asarray(p24, np.float32)
    a = planes
    if p24[0]:
        code, a = run3(O.oracle_gaborish, a, p24[1:7].copy().ctypes.data)
        if code:
            return e4(code), None, None
    sigma = np.zeros(sharp.shape, np.float32)
    if p24[7] > 0:
        p15 = np.concatenate([p24[8:19], p24[19:23]]).astype(np.float32)
        code, a = run3(O.oracle_epf, a, sharp.ctypes.data, hf.ctypes.data, int(p24[7]), p15.ctypes.data, sigma.ctypes.data, quirk)
        if code:
            return e4(code), None, None
    return "", a, sigma


def test_restatement_equals_the_reference_routines_bit_for_bit(libs):
    R, O, _, _ = libs
    rng = np.random.default_rng(1)
    p24 = np.array(PARAMS24, np.float32); p15 = np.concatenate([p24[8:19], p24[19:23]]).astype(np.float32); wts = p24[1:7].copy()
    for (w, h) in SIZES:
        planes, sharp, hf = random_case(rng, w, h)
        if w > 1:
            ca, a = run3(R.ref_kat_gaborish, planes, wts.ctypes.data); cb, b = run3(O.oracle_gaborish, planes, wts.ctypes.data)
            assert ca == cb == 0 and np.array_equal(bits(a), bits(b)), ("gaborish", w, h)
        for iters in (1, 2, 3):
            sa = np.zeros(sharp.shape, np.float32); sb = np.zeros(sharp.shape, np.float32)
            ca, a = run3(R.ref_kat_epf, planes, sharp.ctypes.data, hf.ctypes.data, iters, p15.ctypes.data, sa.ctypes.data)
            cb, b = run3(O.oracle_epf, planes, sharp.ctypes.data, hf.ctypes.data, iters, p15.ctypes.data, sb.ctypes.data, 1)
            assert ca == cb == 0 and np.array_equal(bits(sa), bits(sb)), ("sigma", w, h, iters)
            assert np.array_equal(bits(a), bits(b)), ("epf as the routine stands", w, h, iters)
        # every step by itself, incl. cells the filter skips
        rs = rng.uniform(0.2, 3.0, sharp.shape).astype(np.float32); rs[rng.random(sharp.shape) < 0.2] = -1.0
        for step in (0, 1, 2):
            ca, a = run3(R.ref_kat_epf_step, planes, rs.ctypes.data, step, p15.ctypes.data)
            cb, b = run3(O.oracle_epf_step, planes, rs.ctypes.data, step, p15.ctypes.data, 1)
            assert ca == cb == 0 and np.array_equal(bits(a), bits(b)), ("step", step, w, h)


def test_what_the_routines_aliased_line_buffers_change(libs):
    """quirk off (what the HIP kernels compute by default) against quirk on: channel B only at the picture's corners in rows 0-1 (the
    first buffered rows' unwritten borders), X and Y on the rows y % 4 in {0, 1} (they read the next channel's row there)"""
    _, O, _, _ = libs
    rng = np.random.default_rng(5)
    p24 = np.array(PARAMS24, np.float32); p15 = np.concatenate([p24[8:19], p24[19:23]]).astype(np.float32)
    planes, sharp, hf = random_case(rng, 64, 40)
    rs = rng.uniform(0.2, 3.0, sharp.shape).astype(np.float32)
    for step, cols in ((0, {0, 1, 62, 63}), (1, {0, 63}), (2, {0, 63})):
        _, a = run3(O.oracle_epf_step, planes, rs.ctypes.data, step, p15.ctypes.data, 0)
        _, b = run3(O.oracle_epf_step, planes, rs.ctypes.data, step, p15.ctypes.data, 1)
        d = np.argwhere(bits(a[2]) != bits(b[2]))
        assert len(d) and all(y in (0, 1) and x in cols for y, x in d.tolist()), (step, d.tolist())
        for c in (0, 1):
            rows = set(np.argwhere(bits(a[c]) != bits(b[c]))[:, 0].tolist())
            assert rows and all(y % 4 in (0, 1) for y in rows), (step, c, sorted(rows))


def test_error_codes_are_the_routines_own(libs):
    R, O, S, _ = libs
    w = h = 16
    planes = np.zeros((3, h, w), np.float32); sharp = np.zeros((2, 2), np.int16); hf = np.ones((2, 2), np.float32)
    default_lut = np.array([i / 7 for i in range(8)] + [40, 5, 3.5, 0.46, 0.9, 6.5, 2 / 3], np.float32)   # j40.h:5200: entry 0 is 0
    assert e4(run3(R.ref_kat_epf, planes, sharp.ctypes.data, hf.ctypes.data, 2, default_lut.ctypes.data, None)[0]) == "epf0"
    assert e4(run3(O.oracle_epf, planes, sharp.ctypes.data, hf.ctypes.data, 2, default_lut.ctypes.data, None, 1)[0]) == "epf0"
    lut = default_lut.copy(); lut[0] = 0.1
    bad = sharp.copy(); bad[1, 1] = 9
    assert e4(run3(R.ref_kat_epf, planes, bad.ctypes.data, hf.ctypes.data, 2, lut.ctypes.data, None)[0]) == "shrp"
    assert e4(run3(O.oracle_epf, planes, bad.ctypes.data, hf.ctypes.data, 2, lut.ctypes.data, None, 1)[0]) == "shrp"
    wts = np.array([-0.25, 0, 0.1, 0.1, 0.1, 0.1], np.float32)
    assert e4(run3(R.ref_kat_gaborish, planes, wts.ctypes.data)[0]) == e4(run3(O.oracle_gaborish, planes, wts.ctypes.data)[0]) == "gab0"
    # the device functions' CPU build reports the same three
    for p24, sh, want in ((PARAMS24[:8] + [i / 7 for i in range(8)] + PARAMS24[16:], sharp, "epf0"), (PARAMS24, bad, "shrp"), ([1, -0.25, 0] + PARAMS24[3:], sharp, "gab0")):
        a = planes.copy(); p = np.array(p24, np.float32)
        assert e4(S.hostsim_restoration(a.ctypes.data, w, h, sh.ctypes.data, hf.ctypes.data, p.ctypes.data, 1, None)) == want


@pytest.mark.parametrize("mode", [1, 2])
def test_device_functions_on_the_cpu_equal_the_restatement(libs, mode):
    _, O, S, _ = libs
    rng = np.random.default_rng(7 + mode)
    for (w, h) in SIZES:
        if w < 2:
            continue
        planes, sharp, hf = random_case(rng, w, h)
        for gab, iters in ((1, 0), (0, 1), (0, 2), (1, 3), (1, 2)):
            p24 = np.array(PARAMS24, np.float32); p24[0] = gab; p24[7] = iters
            code, want, sigma = filtered_by(O, planes, sharp, hf, p24, mode == 2)
            assert code == ""
            a = planes.copy(); sg = np.zeros(sharp.shape, np.float32)
            assert S.hostsim_restoration(a.ctypes.data, w, h, sharp.ctypes.data, hf.ctypes.data, p24.ctypes.data, mode, sg.ctypes.data) == 0
            assert np.array_equal(bits(a), bits(want)), (w, h, gab, iters)
            if iters:
                assert np.array_equal(bits(sg), bits(sigma))


# ------------------------------------------------------------------------------------------------------------------------------------
# GPU

STREAMS = [
    ("8x8_only_gab_epf2", 264, 200, dict(fullheader=1, gab=1, epf=2, maxlog=3)),
    ("mixed_transforms_gab2_epf3_custom", 776, 520, dict(fullheader=1, gab=2, epf=3, epfw=1, epfs=1, maxlog=8, bctx=1)),
    ("epf1_two_lf_groups", 2100, 300, dict(fullheader=1, epf=1, cfl=1)),
    ("gab_only_forward_encoded", 520, 392, dict(fullheader=1, gab=1, forward=1)),
]


def frame_cells(f):
    """frame-wide per-cell arrays from the LfGroups' planes: sharpness and the HfMul reciprocal of the covering varblock"""
    W8, H8 = (f.width + 7) // 8, (f.height + 7) // 8
    sharp = np.zeros((H8, W8), np.int16); hf = np.zeros((H8, W8), np.float32)
    for g in range(f.info["num_lf_groups"]):
        gi = f.lf_group_info(g)
        blocks = f.plane(g, 0)
        co = np.zeros(gi["nb_varblocks"], np.int32); inv = np.zeros(gi["nb_varblocks"], np.float32)
        import j40_amd
        j40_amd.lib().j40hip_frame_varblocks(f.h, g, co.ctypes.data, inv.ctypes.data)
        y0, x0 = gi["top"] // 8, gi["left"] // 8
        sharp[y0:y0 + gi["height8"], x0:x0 + gi["width8"]] = f.sharpness(g)
        hf[y0:y0 + gi["height8"], x0:x0 + gi["width8"]] = inv[blocks & 0xfffff]
    return sharp, hf


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_filter_kernels_on_random_planes(libs, mode):
    import j40_amd
    R, O, _, _ = libs
    rng = np.random.default_rng(11 + mode)
    for (w, h) in SIZES + [(1000, 700)]:
        if w < 2:
            continue
        planes, sharp, hf = random_case(rng, w, h)
        for gab, iters in ((1, 0), (0, 1), (0, 2), (0, 3), (1, 3)):
            p24 = np.array(PARAMS24, np.float32); p24[0] = gab; p24[7] = iters
            r = j40_amd.Restoration(); r.gab_enabled = gab; r.epf_iters = iters
            for i in range(6): r.gab_weights[i] = p24[1 + i]
            for i in range(8): r.epf_sharp_lut[i] = p24[8 + i]
            for i in range(3): r.epf_channel_scale[i] = p24[16 + i]
            r.epf_quant_mul, r.epf_pass0_sigma_scale, r.epf_pass2_sigma_scale, r.epf_border_sad_mul = [float(v) for v in p24[19:23]]
            code, got, sigma = j40_amd.kat_device_restoration(planes, sharp, hf, r, mode)
            assert code == ""
            wcode, want, wsigma = filtered_by(O, planes, sharp, hf, p24, mode == 2)
            assert wcode == "" and np.array_equal(bits(got), bits(want)), (w, h, gab, iters)
            if iters:
                assert np.array_equal(bits(sigma), bits(wsigma))
            if mode == 2 and w * h <= 64 * 40:   # and the routines themselves (gaborish, then j40__epf)
                a = planes
                if gab:
                    _, a = run3(R.ref_kat_gaborish, a, p24[1:7].copy().ctypes.data)
                if iters:
                    p15 = np.concatenate([p24[8:19], p24[19:23]]).astype(np.float32)
                    _, a = run3(R.ref_kat_epf, a, sharp.ctypes.data, hf.ctypes.data, iters, p15.ctypes.data, None)
                assert np.array_equal(bits(got), bits(a)), ("against the reference's routines", w, h, gab, iters)


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,opts", STREAMS, ids=[s[0] for s in STREAMS])
@pytest.mark.parametrize("mode", [1, 2])
def test_restored_decode_of_streams(libs, ref, name, w, h, opts, mode):
    import j40_amd
    R, O, _, D = libs
    data = synth("vardct", w, h, 31, **opts)
    buf = C.create_string_buffer(data, len(data))
    f = j40_amd.Frame(data)
    # default: the filters do not run and the pixels are the reference's
    f.upload(0)
    err, plain = f.decode_to_host()
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and np.abs(plain.astype(np.int32) - expect.astype(np.int32)).max() <= 1
    # the filters
    f.set_restoration(mode)
    err, rgba = f.decode_to_host()
    assert err == ""
    xyb0, xyb1 = f.read_xyb(0), f.read_xyb(1)
    want0 = np.zeros((3, h, w), np.float32)
    assert D.oracle_run_xyb(buf, len(data), want0.ctypes.data) == 0
    assert np.array_equal(bits(xyb0), bits(want0)), "the samples the pixel kernels leave in XYB are the oracle's"
    sharp, hf = frame_cells(f)
    p24 = np.array(f.restoration().as_list(), np.float32)
    code, want1, wsigma = filtered_by(O, xyb0, sharp, hf, p24, mode == 2)
    assert code == "" and np.array_equal(bits(xyb1), bits(want1)), "filtered planes"
    if p24[7] > 0:
        assert np.array_equal(bits(f.read_xyb(2)), bits(wsigma))
    if mode == 2:   # the reference's own routines on the same planes
        a = xyb0
        if p24[0]:
            c, a = run3(R.ref_kat_gaborish, a, p24[1:7].copy().ctypes.data); assert c == 0
        if p24[7] > 0:
            c, a = run3(R.ref_kat_epf, a, sharp.ctypes.data, hf.ctypes.data, int(p24[7]), f.restoration().params15().ctypes.data, None); assert c == 0
        assert np.array_equal(bits(xyb1), bits(a)), "against j40__gaborish / j40__epf themselves"
    want_rgba = np.zeros((h, w, 4), np.uint8)
    assert D.oracle_colour(buf, len(data), np.ascontiguousarray(want1).ctypes.data, want_rgba.ctypes.data) == 0
    d = np.abs(rgba.astype(np.int32) - want_rgba.astype(np.int32))
    assert d.max() <= 1, "pixels after the filters"
    assert (rgba != plain).any(), "the filters changed the picture"
    f.close()


@pytest.mark.gpu
def test_default_sharpness_table_is_rejected_like_the_routine_rejects_it(ref):
    """epf with the DEFAULT sharpness table: j40__epf_recip_sigmas raises "epf0" (its first entry is 0, j40.h:5200, 7384), and so does a
    decode that was asked to run the filters -- behind the sections' own codes; without the filters the frame decodes like in j40"""
    import j40_amd
    data = synth("vardct", 264, 200, 9, fullheader=1, epf=2, epflut=0)
    f = j40_amd.Frame(data); f.upload(0); f.set_restoration(1)
    err, _ = f.decode_to_host()
    assert err == "epf0"
    f.set_restoration(0)
    err, plain = f.decode_to_host()
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and np.abs(plain.astype(np.int32) - expect.astype(np.int32)).max() <= 1
    f.close()
    # a damaged section's code comes first
    for at in (len(data) * 3 // 4, len(data) * 7 // 8, len(data) - 40):
        bad = bytearray(data); bad[at] ^= 0x10
        rerr, _ = ref.decode(bytes(bad))
        if not rerr:
            continue
        try:
            g = j40_amd.Frame(bytes(bad)); g.upload(0); g.set_restoration(1)
            code = g.decode_to_host()[0]
            g.close()
        except j40_amd.J40Error as e:
            code = e.code
        assert code == rerr, at


@pytest.mark.gpu
def test_8k_frame_restored(libs):
    """config 3's size: Gaborish + two steps over 7680 x 4320 against the restatement (bit for bit, both modes on the device agree
    with it) -- and through the public API with J40HIP_RESTORATION set, in a process of its own"""
    import j40_amd, subprocess, sys
    _, O, _, _ = libs
    w, h = 7680, 4320
    data = synth("vardct", w, h, 3, fullheader=1, gab=1, epf=2)
    f = j40_amd.Frame(data, threads=8); f.upload(0); f.set_restoration(1)
    err, rgba = f.decode_to_host()
    assert err == ""
    xyb0, xyb1 = f.read_xyb(0), f.read_xyb(1)
    sharp, hf = frame_cells(f)
    p24 = np.array(f.restoration().as_list(), np.float32)
    code, want1, _ = filtered_by(O, xyb0, sharp, hf, p24, 0)
    assert code == "" and np.array_equal(bits(xyb1), bits(want1))
    f.close()
    path = os.path.join(ROOT, "build", "streams", "restored_8k.jxl")
    open(path, "wb").write(data)
    code = ("import sys; sys.path.insert(0, %r); import j40_amd, numpy as np; e, px = j40_amd.decode(open(%r, 'rb').read()); assert e == '', e; "
            "print(int(px.astype(np.uint64).sum()))" % (ROOT, path))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, J40HIP_RESTORATION="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert int(out.stdout.strip().splitlines()[-1]) == int(rgba.astype(np.uint64).sum())
