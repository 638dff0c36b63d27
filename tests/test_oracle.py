"""Pins the CPU oracle (oracle/hotpath_oracle.c, the plain-C restatement of the hot path) against the
unmodified reference (oracle/_ref) and against the committed golden fixtures. CPU only."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from streams import synth, VARDCT_CASES, MODULAR_CASES, ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def oracle(built):
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    D.oracle_run.restype = C.c_uint32
    D.oracle_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]

    def run(data, w, h, ncoeff=0):
        rgba = np.zeros((h, w, 4), np.uint8)
        co = np.zeros(ncoeff, np.float32) if ncoeff else None
        buf = C.create_string_buffer(data, len(data))
        err = D.oracle_run(buf, len(data), rgba.ctypes.data, co.ctypes.data if ncoeff else None)
        return err, rgba, co
    return run


@pytest.mark.parametrize("name,opts", VARDCT_CASES + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))])
def test_vardct_restatement_equals_reference(ref, oracle, name, opts):
    from refdec import RefStage
    w, h = (776, 520) if name == "all_transforms" else (392, 264)
    data = synth("vardct", w, h, 81, **opts)
    rs = RefStage(ref, data)
    sizes = [rs.lf_group_info(g)["width8"] * rs.lf_group_info(g)["height8"] * 64 for g in range(rs.info["num_lf_groups"])]
    err, rgba, co = oracle(data, w, h, 3 * sum(sizes))
    assert err == 0
    off = 0
    for g, n in enumerate(sizes):
        for c in range(3):
            assert np.array_equal(co[off:off + n], rs.coeffs(g, c)), "quantised coefficients"
            off += n
    assert rs.combine() == ""
    assert np.array_equal(rgba, rs.rgba()), "the restatement is built like the reference (no FMA, same libm): RGBA must be identical"
    rs.close()


@pytest.mark.parametrize("name,w,h,opts", MODULAR_CASES)
def test_modular_restatement_equals_reference(ref, oracle, name, w, h, opts):
    data = synth("modular", w, h, 83, **opts)
    rerr, expect = ref.decode(data)
    assert rerr == ""
    err, rgba, _ = oracle(data, w, h)
    assert err == 0 and np.array_equal(rgba, expect)


def test_restatement_against_golden_fixtures(oracle):
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for name, e in sorted(manifest.items()):
        data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
        err, rgba, _ = oracle(data, e["width"], e["height"])
        assert err == 0 and sha(rgba) == e["rgba_sha256"], name


def test_restatement_reports_reference_errors(ref, oracle):
    data = bytearray(synth("vardct", 392, 264, 81))
    rng = np.random.default_rng(3)
    seen = 0
    for _ in range(10):
        m = bytearray(data)
        m[int(rng.integers(len(m) // 2, len(m) - 4))] ^= 4
        rerr = ref.decode(bytes(m))[0]
        err, _, _ = oracle(bytes(m), 392, 264)
        assert (err != 0) == (rerr != ""), (rerr, err)
        if rerr:
            from refdec import err4
            assert err4(err) == rerr
            seen += 1
    assert seen >= 1


@pytest.mark.parametrize("name,opts", VARDCT_CASES + [("all_transforms", dict(maxlog=8, bctx=1, presets=2, orders=1))])
def test_seam_frame_from_view_carries_everything(ref, built, name, opts):
    """j40hip_frame_from_vardct_view: a handle built from the plan view alone (what a host with its own parser hands over,
    INTEGRATION.md) yields the same view again -- the CPU checker decodes it to the reference's pixels"""
    D = C.CDLL(os.path.join(ROOT, "build", "liboracle_driver.so"))
    D.seam_roundtrip.restype = C.c_uint32
    D.seam_roundtrip.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    w, h = (776, 520) if name == "all_transforms" else (392, 264)
    data = synth("vardct", w, h, 81, **opts)
    rerr, expect = ref.decode(data)
    assert rerr == ""
    rgba = np.zeros((h, w, 4), np.uint8)
    buf = C.create_string_buffer(data, len(data))
    assert D.seam_roundtrip(buf, len(data), rgba.ctypes.data, 0) == 0
    assert np.array_equal(rgba, expect)
    # ... and so does the LF bundle, the same view flattened into one relocatable blob for another process (a sharded decode's
    # broadcast): j40hip_frame_lf_bundle -> j40hip_frame_from_lf_bundle
    rgba[:] = 0
    assert D.seam_roundtrip(buf, len(data), rgba.ctypes.data, 2) == 0
    assert np.array_equal(rgba, expect)


def test_lf_bundle_blobs_are_checked(built):
    import j40_amd
    data = synth("vardct", 392, 264, 81)
    fr = j40_amd.Frame(data)
    blob = fr.lf_bundle()
    fr.close()
    assert len(blob) > len(data), "the bundle carries the parsed LF data next to the codestream"
    again = j40_amd.Frame.from_lf_bundle(blob)
    assert (again.width, again.height) == (392, 264)
    again.close()
    # (sizes, counts and the offsets inside the blob are checked; it is a message between the ranks of one job, not an input format)
    for damaged in (blob[:100], blob[:-1], b"\0" * len(blob), blob[:8] + (len(blob) + 16).to_bytes(8, "little") + blob[16:], blob[:16] + blob[16:].replace(blob[16:24], b"\xff" * 8, 1)):
        with pytest.raises(j40_amd.J40Error):
            j40_amd.Frame.from_lf_bundle(damaged)
