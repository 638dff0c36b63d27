"""GPU parity tests (run with -m gpu on an MI355X): the HIP hot path behind the C-ABI against the
unmodified reference (oracle/_ref) and the committed golden fixtures.

Bars: quantised HF coefficients bit-exact; RGBA within 1 u8 level per channel (VarDCT float path;
the only non-bit-identical step is powf in the sRGB transfer)."""
import hashlib
import json
import os

import numpy as np
import pytest

from streams import synth, VARDCT_CASES, MODULAR_CASES, ROOT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gpu(built):
    import j40_amd
    assert j40_amd.device_count() > 0, "the gpu tests need a HIP device"
    return j40_amd


def compare(rgba, expect):
    assert rgba is not None and rgba.shape == expect.shape
    d = np.abs(rgba.astype(np.int32) - expect.astype(np.int32))
    return int(d.max()), int((d > 0).sum())


@pytest.mark.parametrize("name,opts", VARDCT_CASES + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))])
def test_vardct_public_api_matches_reference(gpu, ref, name, opts):
    w, h = (776, 520) if name == "all_transforms" else (520, 264)
    data = synth("vardct", w, h, 41, **opts)
    err, rgba = gpu.decode(data)
    assert err == ""
    rerr, expect = ref.decode(data)
    assert rerr == ""
    dmax, ndiff = compare(rgba, expect)
    assert dmax <= 1, "max |delta| %d, %d differing samples" % (dmax, ndiff)
    assert ndiff <= rgba.size // 10000 + 4, "unexpectedly many off-by-one samples: %d" % ndiff
    assert np.all(rgba[..., 3] == 255)


@pytest.mark.parametrize("name,opts", VARDCT_CASES[:8] + VARDCT_CASES[10:] + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))])
def test_hf_coefficients_bit_exact(gpu, ref, name, opts):
    from refdec import RefStage
    w, h = (776, 520) if name == "all_transforms" else (520, 264)
    data = synth("vardct", w, h, 43, **opts)
    rs = RefStage(ref, data)
    fr = gpu.Frame(data)
    fr.upload(0)
    err, first = fr.decode_to_host()
    assert err == ""
    for g in range(rs.info["num_lf_groups"]):
        for c in range(3):
            assert np.array_equal(fr.read_coeffs(g, c), rs.coeffs(g, c)), (g, c)
    err, second = fr.decode_to_host()           # decoding again on the same working set gives the same pixels
    assert err == "" and np.array_equal(first, second)
    if rs.info["num_passes"] == 1:              # single-pass frames use event lists; dense planes (the fallback after "evof") agree
        fr.force_dense(True)
        fr.upload(0)
        err, third = fr.decode_to_host()
        assert err == "" and np.array_equal(first, third)
        for g in range(rs.info["num_lf_groups"]):
            for c in range(3):
                assert np.array_equal(fr.read_coeffs(g, c), rs.coeffs(g, c)), (g, c)
    fr.close()
    rs.close()


@pytest.mark.parametrize("name,w,h,opts", MODULAR_CASES)
def test_modular_bit_exact(gpu, ref, name, w, h, opts):
    data = synth("modular", w, h, 71, **opts)
    err, rgba = gpu.decode(data)
    assert err == ""
    rerr, expect = ref.decode(data)
    assert rerr == "" and np.array_equal(rgba, expect), "Modular output must be bit-exact"


def test_modular_2048_multi_group_bit_exact(gpu, ref):
    data = synth("modular", 2048, 2048, 12, tree=1)
    err, rgba = gpu.decode(data)
    assert err == "" and np.array_equal(rgba, ref.decode(data)[1])


def test_modular_16384_full_size_by_periodicity(gpu, ref):
    """BASELINE.json config 4 at its full size: 16384 x 16384 Modular, 4096 pass groups, global RCT. The stream is the
    1024 x 1024 picture tiled 16 x 16 (jxlsynth repeat=16 reuses the encoded group sections), so two size-independent
    properties pin the result: the 1024 x 1024 stream decodes bit-exactly like the reference, and the large frame is that
    picture repeated -- every one of the 4096 sections landed where it belongs."""
    import torch
    base = synth("modular", 1024, 1024, 21, tree=1)
    rerr, tile = ref.decode(base)
    assert rerr == ""
    data = synth("modular", 16384, 16384, 21, tree=1, repeat=16)
    fr = gpu.Frame(data)
    assert (fr.width, fr.height) == (16384, 16384)
    fr.upload(0)
    out = torch.zeros((16384, 16384, 4), dtype=torch.uint8, device="cuda:0")
    ms = fr.decode_timed(out.data_ptr(), 16384 * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert fr.status() == ""
    print("16384x16384 Modular: sections %.1f ms, transforms + pack %.1f ms" % (ms[0], ms[1]))
    t = torch.from_numpy(tile).to("cuda:0")
    tiled = out.view(16, 1024, 16, 1024, 4)
    assert bool((tiled == t.view(1, 1024, 1, 1024, 4)).all())
    fr.close()


def test_seam_frame_from_view_decodes_on_the_gpu(gpu, ref):
    """the seam for a host with its own parser: parse -> plan view -> j40hip_frame_from_vardct_view -> upload -> decode gives
    the pixels of the direct path (multi-pass, presets, custom orders, prefix codes + LZ77 among the cases)"""
    import ctypes as C
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    D.seam_roundtrip.restype = C.c_uint32
    D.seam_roundtrip.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    for i, (name, opts) in enumerate(VARDCT_CASES + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))]):
        w, h = (776, 520) if name == "all_transforms" else (392, 264)
        data = synth("vardct", w, h, 81, **opts)
        err, direct = gpu.decode(data)
        assert err == ""
        rgba = np.zeros((h, w, 4), np.uint8)
        buf = C.create_string_buffer(data, len(data))
        assert D.seam_roundtrip(buf, len(data), rgba.ctypes.data, 1) == 0, name
        assert np.array_equal(rgba, direct), name
        rgba[:] = 0   # the same through the LF-bundle blob
        assert D.seam_roundtrip(buf, len(data), rgba.ctypes.data, 3) == 0, name
        assert np.array_equal(rgba, direct), name


def test_reference_cli_source_unchanged_runs_on_the_hip_library(gpu, tmp_path):
    """drop-in at the source level: the reference's own command-line decoder (dj40.c, not a line changed) built against
    include/j40.h + libj40hip.so (oracle/Makefile `dropin`) writes the same PNGs as the reference's own build, and reports a
    corrupt file the same way"""
    import subprocess
    exe_ref, exe_hip = os.path.join(ROOT, "oracle", "_ref", "dj40-ref"), os.path.join(ROOT, "oracle", "_ref", "dj40-hip")
    if not (os.path.exists(exe_ref) and os.path.exists(exe_hip)):
        pytest.skip("oracle/_ref/dj40-* not built (needs the reference sources at build time)")
    cases = [("modular", 600, 300, dict(tree=1, alpha=1)), ("modular", 256, 256, dict(alpha=1, prefix=1, lz77=1)),
             ("vardct", 520, 264, dict()), ("vardct", 776, 520, dict(maxlog=8, bctx=1, presets=2, orders=1))]
    for i, (mode, w, h, opts) in enumerate(cases):
        src = tmp_path / ("in%d.jxl" % i)
        src.write_bytes(synth(mode, w, h, 300 + i, **opts))
        outs = []
        for exe in (exe_ref, exe_hip):
            png = tmp_path / ("out%d_%s.png" % (i, os.path.basename(exe)))
            r = subprocess.run([exe, str(src), str(png)], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, (exe, r.stderr)
            assert "%dx%d frame read." % (w, h) in r.stderr
            outs.append(png.read_bytes())
        if outs[0] != outs[1]:   # VarDCT may differ by one level
            from PIL import Image
            import io
            a, b = (np.asarray(Image.open(io.BytesIO(o))).astype(np.int32) for o in outs)
            assert mode == "vardct" and np.abs(a - b).max() <= 1
    bad = bytearray(synth("vardct", 520, 264, 300)); bad[len(bad) // 2] ^= 0x55
    src = tmp_path / "bad.jxl"; src.write_bytes(bytes(bad))
    msgs = [subprocess.run([exe, str(src), str(tmp_path / "bad.png")], capture_output=True, text=True, timeout=120) for exe in (exe_ref, exe_hip)]
    assert msgs[0].returncode == msgs[1].returncode == 1 and msgs[0].stderr == msgs[1].stderr, (msgs[0].stderr, msgs[1].stderr)


def test_modular_corruption_is_reported(gpu, ref):
    data = bytearray(synth("modular", 600, 300, 71, tree=1))
    rng = np.random.default_rng(11)
    rejected = 0
    for _ in range(8):
        m = bytearray(data)
        m[int(rng.integers(len(m) // 2, len(m) - 4))] ^= 0x10
        rerr, rexp = ref.decode(bytes(m))
        err, rgba = gpu.decode(bytes(m))
        if rerr == "":
            assert err == "" and np.array_equal(rgba, rexp)
        else:
            assert err != ""
            rejected += 1
    assert rejected >= 1


@pytest.mark.parametrize("opts", [dict(alpha=1, prefix=1, lz77=1), dict(alpha=1), dict(prefix=1), dict(lz77=1, groupshift=7), dict(palette=2, alpha=1, prefix=1, lz77=1)],
                         ids=["config1_prefix_lz77", "ans", "prefix", "ans_lz77_group128", "palette_prefix_lz77"])
def test_two_pass_modular_sections_pixels_and_codes(gpu, ref, opts):
    """sections with a position-only MA tree go through the two-pass decoder (modular_split.hip: the stream's tokens first -- prefix codes
    sixty-four symbols at a time --, the prediction behind them): the reference's pixels on the clean stream, and on 60 damaged ones
    (a flipped bit, a truncation, bytes appended) the reference's pixels or the reference's 4-char code"""
    import j40_amd
    data = synth("modular", 256, 256, 101, **opts)
    f = j40_amd.Frame(data)
    assert f.split_sections() >= 1
    f.close()
    err, rgba = gpu.decode(data)
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and np.array_equal(rgba, expect)
    rng = np.random.default_rng(5)
    rejected = 0
    for k in range(60):
        m = bytearray(data)
        if k % 10 == 8:
            m = m[:int(rng.integers(len(m) // 3, len(m) - 1))]
        elif k % 10 == 9:
            m += bytes(int(rng.integers(1, 9)))
        else:
            m[int(rng.integers(40, len(m)))] ^= 1 << int(rng.integers(0, 8))
        rerr, rexp = ref.decode(bytes(m))
        err, got = gpu.decode(bytes(m))
        assert err == rerr, (k, err, rerr)
        if rerr == "":
            assert np.array_equal(got, rexp), k
        else:
            rejected += 1
    assert rejected >= 5


def test_lz77_distance_multiplier_of_lf_global_on_the_gpu(gpu, ref):
    """see tests/test_hostsim.py::test_lz77_distance_multiplier_of_lf_global_is_the_whole_images: the same damaged stream through K3"""
    data = bytearray(synth("modular", 645, 28, 75417, palette=3, prefix=1, lz77=1, permute=1))
    data[1249] ^= 0x08
    err, rgba = gpu.decode(bytes(data))
    rerr, expect = ref.decode(bytes(data))
    assert err == "" and rerr == "" and np.array_equal(rgba, expect)


def test_hip_path_against_cpu_oracle(gpu):
    """HIP kernels vs the plain-C restatement (oracle/hotpath_oracle.c), both behind the same plan seam"""
    import ctypes as C
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    D.oracle_run.restype = C.c_uint32
    D.oracle_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    cases = [("vardct", 776, 520, dict(maxlog=8, bctx=1, presets=2, orders=1)), ("vardct", 520, 264, dict(passes=2)),
             ("modular", 600, 300, dict(tree=2)), ("modular", 256, 256, dict(alpha=1, prefix=1, lz77=1)), ("modular", 600, 300, dict(palette=2, alpha=1))]
    for mode, w, h, opts in cases:
        data = synth(mode, w, h, 91, **opts)
        expect = np.zeros((h, w, 4), np.uint8)
        buf = C.create_string_buffer(data, len(data))
        assert D.oracle_run(buf, len(data), expect.ctypes.data, None) == 0
        err, rgba = gpu.decode(data)
        assert err == ""
        if mode == "modular":
            assert np.array_equal(rgba, expect)
        else:
            assert compare(rgba, expect)[0] <= 1


def test_device_srgb_tail_matches_correctly_rounded_powf(gpu):
    """the seeded fp64 root that replaces powf in the pixel kernels, on the device's own log2/exp2/rcp seeds:
    8-bit samples identical to the ones a correctly rounded powf yields (what the reference's libm gives)"""
    rng = np.random.default_rng(5)
    v = np.concatenate([np.linspace(0.0, 1.2, 3000001, dtype=np.float32), rng.uniform(0.9, 70000.0, 2000000).astype(np.float32),
                        np.float32(2.0) ** rng.uniform(-20, 120, 500000).astype(np.float32), np.array([np.inf, np.nan, -1.0, 0.0031308, 0.00313081], np.float32)])
    out = np.zeros(v.size, np.uint8)
    assert gpu.lib().j40hip_kat_device_srgb_u8(v.ctypes.data, v.size, out.ctypes.data) == 0
    P = np.float64(np.float32(1.0) / np.float32(2.4))
    with np.errstate(all="ignore"):
        p = np.power(v.astype(np.float64), P).astype(np.float32)
        t = np.where(v <= np.float32(0.0031308), np.float32(12.92) * v, np.float32(1.055) * p - np.float32(0.055)).astype(np.float32)
        y = (np.float32(255.0) * t + np.float32(0.5)).astype(np.float32)
        # (int16_t) of a float on x86: cvttss2si to int32 (0x80000000 when out of range / NaN), then the low 16 bits
        ok = np.isfinite(y) & (np.abs(y) < 2147483648.0)
        i32 = np.where(ok, np.trunc(np.where(ok, y, 0)).astype(np.int64), -2147483648)
    i16 = ((i32 & 0xFFFF) ^ 0x8000) - 0x8000
    expect = np.clip(i16, 0, 255).astype(np.uint8)
    bad = np.nonzero(out != expect)[0]
    assert bad.size <= 2, (bad.size, v[bad[:5]], out[bad[:5]], expect[bad[:5]])


def test_batch_throughput_mode_matches_latency_mode(gpu, ref):
    """j40hip_batch_decode (one section per wavefront lane, many frames per launch) gives exactly the pixels of
    j40hip_frame_decode (one section per wavefront) and the reference's within 1 level, per frame"""
    import torch
    cases = [(n, o) for n, o in VARDCT_CASES] + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))]
    frames, outs, datas = [], [], []
    for i, (name, opts) in enumerate(cases):
        w, h = [(520, 264), (776, 520), (264, 520), (1032, 300)][i % 4]
        data = synth("vardct", w, h, 77 + i, **opts)
        fr = gpu.Frame(data)
        fr.upload(0)
        frames.append(fr); datas.append(data)
        outs.append(torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0"))
    # the rANS-only members alone take the specialised lane kernel (k_hf_lanes), the full set the generic one
    fast = [i for i, (_, o) in enumerate(cases) if not (o.get("hfprefix") or o.get("hflz77"))]
    b_fast = gpu.Batch([frames[i] for i in fast])
    b_fast.decode([outs[i].data_ptr() for i in fast], [outs[i].shape[1] * 4 for i in fast], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fast_pixels = {i: outs[i].cpu().numpy().copy() for i in fast}
    for o in outs:
        o.zero_()
    b_fast.close()
    batch = gpu.Batch(frames)
    batch.decode([o.data_ptr() for o in outs], [o.shape[1] * 4 for o in outs], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in fast:
        assert np.array_equal(fast_pixels[i], outs[i].cpu().numpy()), cases[i][0]
    for fr, o, data, (name, _) in zip(frames, outs, datas, cases):
        assert fr.status() == "", name
        got = o.cpu().numpy()
        err, single = fr.decode_to_host()
        assert err == "" and np.array_equal(got, single), name
        rerr, expect = ref.decode(data)
        assert rerr == "" and compare(got, expect)[0] <= 1, name
    # the alternative launch form of the pixel kernels: one launch per transform class over all frames (blockIdx.y = frame)
    os.environ["J40HIP_K2_BATCHED"] = "1"
    try:
        bw = gpu.Batch(frames)
    finally:
        del os.environ["J40HIP_K2_BATCHED"]
    again = [torch.zeros_like(o) for o in outs]
    bw.decode([o.data_ptr() for o in again], [o.shape[1] * 4 for o in again], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for o, a, (name, _) in zip(outs, again, cases):
        assert torch.equal(o, a), name
    bw.close()
    # a corrupt member fails alone
    bad = bytearray(datas[0]); bad[len(bad) // 2] ^= 0x55
    try:
        fb = gpu.Frame(bytes(bad)); fb.upload(0)
    except gpu.J40Error:
        fb = None
    if fb is not None:
        b2 = gpu.Batch([frames[1], fb])
        b2.decode([outs[1].data_ptr(), outs[0].data_ptr()], [outs[1].shape[1] * 4, outs[0].shape[1] * 4], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert frames[1].status() == ""
        rerr, _ = ref.decode(bytes(bad))
        assert fb.status() == rerr
        b2.close()
    batch.close()


def test_group_ranges_decode_disjoint_parts_of_the_frame(gpu, ref):
    """j40hip_frame_set_group_range (the per-rank share of a sharded decode): three ranges written into one buffer give
    the pixels of a whole decode, and each range touches only its own groups"""
    import torch
    from j40_amd import sharding
    w, h = 776, 600   # 4 x 3 groups
    data = synth("vardct", w, h, 59, maxlog=8)
    fr = gpu.Frame(data)
    fr.upload(0)
    err, whole = fr.decode_to_host()
    assert err == ""
    out = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    ranges = sharding.plan_ranges(fr, 3)   # contiguous, balanced by section bytes
    assert sum(c for _, c in ranges) == fr.info["num_groups"]
    for first, count in ranges:
        fr.set_group_range(first, count)
        before = out.clone()
        fr.decode(out.data_ptr(), w * 4, stream)
        torch.cuda.synchronize()
        assert fr.status() == ""
        mine = torch.zeros((h, w), dtype=torch.bool, device="cuda:0")
        for x0, y0, x1, y1 in sharding.range_rectangles(first, count, w, h, fr.info["group_size_shift"]):
            mine[y0:y1, x0:x1] = True
        assert torch.equal(out[~mine], before[~mine]), "a range wrote outside its own groups"
    assert np.array_equal(out.cpu().numpy(), whole)
    fr.set_group_range(0, fr.info["num_groups"])
    rerr, expect = ref.decode(data)
    assert rerr == "" and compare(whole, expect)[0] <= 1


def test_baseline_config_sizes(gpu, ref):
    """BASELINE.json's configurations at (or near) their full sizes against the reference: config 1 (256x256 Modular RGBA,
    single section, prefix codes + LZ77), config 2 (3840x2160 VarDCT, latency mode), config 5 (a batch of 1920x1080 VarDCT
    frames, throughput mode: 40 groups per frame, i.e. partially filled wavefronts), config 4's shape at 2048x2048
    (multi-group Modular with a global RCT). Config 3 (7680x4320) is compared with the reference inside bench.py."""
    import torch
    data = synth("modular", 256, 256, 101, alpha=1, prefix=1, lz77=1)
    err, rgba = gpu.decode(data)
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and np.array_equal(rgba, expect)

    data = synth("vardct", 3840, 2160, 102)
    err, rgba = gpu.decode(data)
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and compare(rgba, expect)[0] <= 1

    frames, outs, datas = [], [], []
    for i in range(6):
        d = synth("vardct", 1920, 1080, 110 + i % 3)
        fr = gpu.Frame(d); fr.upload(0)
        frames.append(fr); datas.append(d)
        outs.append(torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda:0"))
    batch = gpu.Batch(frames)
    batch.decode([o.data_ptr() for o in outs], [1920 * 4] * len(outs), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for fr, o, d in zip(frames[:3], outs[:3], datas[:3]):
        assert fr.status() == ""
        rerr, expect = ref.decode(d)
        assert rerr == "" and compare(o.cpu().numpy(), expect)[0] <= 1
    assert torch.equal(outs[0], outs[3]) and torch.equal(outs[2], outs[5])
    batch.close()

    data = synth("modular", 2048, 2048, 103)
    err, rgba = gpu.decode(data)
    rerr, expect = ref.decode(data)
    assert err == rerr == "" and np.array_equal(rgba, expect)


def test_full_event_region_falls_back_to_dense_planes(ref):
    """a section with more non-zero coefficients than its event region holds reports "evof" on the asynchronous path and is
    decoded with dense planes by the synchronous one (run in a subprocess: the region size is fixed when the library loads)"""
    import subprocess, sys
    code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, j40_amd
from streams import synth
data = synth("vardct", 520, 264, 71)
fr = j40_amd.Frame(data); fr.upload(0)
out = torch.zeros((264, 520, 4), dtype=torch.uint8, device="cuda:0")
fr.decode(out.data_ptr(), 520 * 4, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
assert fr.status() == "evof", fr.status()
err, rgba = j40_amd.decode(data)          # public API: falls back by itself
assert err == "", err
np.save(sys.argv[1], rgba)
""" % (ROOT, ROOT)
    out = os.path.join(ROOT, "build", "evof_test.npy")
    env = dict(os.environ, J40HIP_EVENTS_PER_BYTE="0")
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
    data = synth("vardct", 520, 264, 71)
    rerr, expect = ref.decode(data)
    assert rerr == "" and compare(np.load(out), expect)[0] <= 1


def test_golden_fixtures(gpu):
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for name, e in sorted(manifest.items()):
        data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
        fr = gpu.Frame(data)
        fr.upload(0)
        err, rgba = fr.decode_to_host()
        assert err == "", name
        if e["mode"] == "modular":
            assert sha(rgba) == e["rgba_sha256"], name   # integer path: bit-exact
            fr.close()
            continue
        co = [fr.read_coeffs(g, c) for g in range(fr.info["num_lf_groups"]) for c in range(3)]
        assert sha(np.concatenate(co)) == e["coeffs_sha256"], name
        if sha(rgba) != e["rgba_sha256"]:
            # not bit-identical: must then be within one level of the reference's output
            from refdec import Ref
            expect = Ref().decode(data)[1]
            assert compare(rgba, expect)[0] <= 1, name
        fr.close()


def test_4k_frame_matches_reference(gpu, ref):
    data = synth("vardct", 3840, 2160, 13)
    err, rgba = gpu.decode(data)
    assert err == ""
    expect = ref.decode(data)[1]
    dmax, ndiff = compare(rgba, expect)
    assert dmax <= 1 and ndiff <= 3000, (dmax, ndiff)


def test_8k_frame_matches_reference(gpu, ref):
    data = synth("vardct", 7680, 4320, 3)
    err, rgba = gpu.decode(data)
    assert err == ""
    expect = ref.decode(data)[1]
    dmax, ndiff = compare(rgba, expect)
    assert dmax <= 1 and ndiff <= 12000, (dmax, ndiff)


def test_ragged_sizes(gpu, ref):
    for w, h in ((257, 9), (263, 511), (1000, 257), (2049, 300)):
        data = synth("vardct", w, h, w + h)
        err, rgba = gpu.decode(data)
        assert err == "", (w, h, err)
        assert compare(rgba, ref.decode(data)[1])[0] <= 1, (w, h)


def test_output_layout_and_repeat_calls(gpu):
    data = synth("vardct", 520, 264, 41)
    img = gpu.from_memory(data)
    assert img.output_format() == ""
    assert img.next_frame()
    rgba, stride, ptr = img.frame_pixels_u8x4()
    assert stride == 32 * ((4 * 520 + 1 + 31) // 32) and ptr % 32 == 0  # j40.h:1061-1065, 7939
    assert not img.next_frame()           # single frame: second call says "no more" (j40.h:8390)
    assert img.error() == ""
    img.free()
    assert img.error() == "Ufre"   # j40.h:8475-8476: a freed image answers every call with "Ufre"


def test_corrupt_sections_report_reference_errors(gpu, ref):
    data = bytearray(synth("vardct", 520, 264, 41))
    rng = np.random.default_rng(7)
    checked = 0
    for _ in range(12):
        pos = int(rng.integers(len(data) // 2, len(data) - 8))
        mutated = bytearray(data)
        mutated[pos] ^= 1 << int(rng.integers(0, 8))
        rerr, rexp = ref.decode(bytes(mutated))
        err, rgba = gpu.decode(bytes(mutated))
        if rerr == "":
            assert err == "" and compare(rgba, rexp)[0] <= 1
        else:
            assert err != "", "the reference rejects this stream (%s) but the GPU path accepted it" % rerr
            checked += 1
    assert checked >= 1
    err, _ = gpu.decode(bytes(data[: len(data) - 100]))
    assert err == "shrt"


def test_damage_in_extra_channel_sub_images_is_reported(gpu, ref):
    """VarDCT frame with an alpha channel: every pass-group section goes on with the channel's Modular sub-image after the HF
    coefficients. The reference decodes it (and later drops it); damage there must give the reference's error code, not pass
    unnoticed (runtime.hip: validate_trailers)"""
    data = synth("vardct", 520, 264, 33, alpha=1)
    rng = np.random.default_rng(5)
    seen = {}
    for _ in range(40):
        mutated = bytearray(data)
        pos = int(rng.integers(len(data) // 3, len(data)))
        mutated[pos] ^= 1 << int(rng.integers(0, 8))
        rerr, rexp = ref.decode(bytes(mutated))
        err, rgba = gpu.decode(bytes(mutated))
        assert err == rerr, (pos, rerr, err)
        if rerr == "":
            assert compare(rgba, rexp)[0] <= 1
        seen[rerr] = seen.get(rerr, 0) + 1
        # throughput mode (j40hip_batch_decode) must agree with latency mode on every one of these streams: the batch cannot stop
        # for the host in the middle, so the sub-images are validated when the status is asked for (runtime.hip: trailers_pending)
        import torch
        try:
            fr = gpu.Frame(bytes(mutated))
        except gpu.J40Error as e:
            assert e.code == rerr
            continue
        fr.upload(0)
        out = torch.zeros((264, 520, 4), dtype=torch.uint8, device="cuda:0")
        batch = gpu.Batch([fr])
        batch.decode([out.data_ptr()], [520 * 4], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if rerr != "excs":   # (bytes behind the frame are the public API's business: j40hip_frame_after_frame_status)
            assert fr.status() == rerr, ("batch", pos, rerr, fr.status())
        if rerr == "":
            assert compare(out.cpu().numpy(), rexp)[0] <= 1
        batch.close(); fr.close()
    assert sum(v for k, v in seen.items() if k) >= 10, seen


def test_section_ends_and_bytes_behind_the_frame_like_the_reference(gpu, ref):
    """junk behind the sections' data (accepted in multi-section frames, `shrt` in single-section ones), bytes behind the frame
    (`excs` when the reference's main buffer still holds them, j40hip_frame_after_frame_status), damaged prefix-coded sections"""
    for mode, w, h, o in [("modular", 600, 300, dict(slack=2)), ("modular", 256, 256, dict(slack=2)), ("vardct", 520, 264, dict(slack=3)),
                          ("vardct", 520, 264, dict(slack=1, alpha=1)), ("modular", 600, 300, dict(slack=1, prefix=1, lz77=1))]:
        d = synth(mode, w, h, 9, **o)
        rerr, px = ref.decode(d)
        err, out = gpu.decode(d)
        assert err == rerr, (mode, o, rerr, err)
        if rerr == "":
            assert compare(out, px)[0] <= (0 if mode == "modular" else 1)
    for mode, w, h, o in [("modular", 600, 300, dict()), ("modular", 256, 256, dict()), ("modular", 600, 300, dict(palette=1)), ("vardct", 520, 264, dict()), ("vardct", 1100, 700, dict())]:
        d = synth(mode, w, h, 9, **o) + b"\x00\x01\x02"
        rerr, _ = ref.decode(d)
        err, _ = gpu.decode(d)
        assert err == rerr, (mode, w, h, o, rerr, err)
    d = synth("modular", 762, 8, 236738, prefix=1, lz77=1)
    rng = np.random.default_rng(3)
    for _ in range(30):
        b = bytearray(d)
        b[int(rng.integers(len(d) // 4, len(d)))] ^= 1 << int(rng.integers(8))
        rerr, px = ref.decode(bytes(b))
        err, out = gpu.decode(bytes(b))
        assert err == rerr and (rerr != "" or np.array_equal(px, out))


def test_single_image_two_phases_equal_one_phase(gpu, ref):
    """j40hip_frame_decode_to_host in two phases (runtime.hip: the longest sections on a stream of their own, the image on its way over the
    link while they finish, their groups' rectangles on top) against the same call with J40HIP_TWO_PHASE=0, and against the reference:
    same pixels, same codes -- whole frames, a repeated decode of one upload, streams with a flipped bit in the pass-group sections"""
    cases = [("vardct", 2600, 2100, 71, dict(forward=1)), ("vardct", 4096, 2304, 72, dict()), ("vardct", 3000, 2100, 73, dict(forward=1))]
    rng = np.random.default_rng(5)
    used = damaged = 0
    for (mode, w, h, seed, opts) in cases:
        data = synth(mode, w, h, seed, **opts)
        variants = [data]
        for _ in range(3):   # (behind the LfGroup sections: a pass-group section, most of the time)
            m = bytearray(data)
            m[int(rng.integers(len(m) // 3, len(m) - 16))] ^= 1 << int(rng.integers(0, 8))
            variants.append(bytes(m))
        for i, d in enumerate(variants):
            out = {}
            for two in ("1", "0"):
                os.environ["J40HIP_TWO_PHASE"] = two
                try:
                    fr = gpu.Frame(d, threads=4)
                except gpu.J40Error as e:
                    out[two] = (e.code, None, 0)
                    continue
                fr.upload(0)
                code, px = fr.decode_to_host()
                k = fr.two_phase_sections()
                if two == "1" and i == 0:
                    code2, px2 = fr.decode_to_host()   # (the same upload again: the long sections' entries are in the table by now)
                    assert code2 == code and np.array_equal(px, px2)
                out[two] = (code, px, k)
                fr.close()
            os.environ.pop("J40HIP_TWO_PHASE", None)
            assert out["1"][0] == out["0"][0], (w, h, i, out["1"][0], out["0"][0])
            assert out["0"][2] <= 0
            if out["1"][2] > 0:
                used += 1
            if out["1"][0] == "":
                assert np.array_equal(out["1"][1], out["0"][1]), (w, h, i)
            else:
                damaged += 1
            if i == 0:
                rerr, expect = ref.decode(d)
                assert rerr == "" and out["1"][0] == ""
                dmax, ndiff = compare(out["1"][1], expect)
                assert dmax <= 1 and ndiff <= 6000, (dmax, ndiff)
    assert used >= 3, used
