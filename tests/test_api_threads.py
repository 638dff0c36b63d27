"""The drop-in boundary under load: many threads running the reference's public API sequence unchanged (tests/api_threads.c does
what /root/reference/dj40.c:29-50 does, image after image) against libj40hip.so. With several threads inside the API at once
j40_next_frame hands the images to the device's process-wide pipeline (j40_amd/csrc/api.cpp, device/pipeline.hip: j40hip_pipeline_run)
so that the callers share batches; the pixels and error codes must be the lone call's and the reference's
(j40.h:8377-8401, 8425-8462; layout j40.h:1061-1065, 7939)."""
import json
import os
import subprocess

import numpy as np
import pytest

from streams import synth, CACHE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "api_threads")


def _path(mode, w, h, seed, **opts):
    synth(mode, w, h, seed, **opts)
    key = "%s_%d_%d_%d_%s" % (mode, w, h, seed, "_".join("%s-%s" % kv for kv in sorted(opts.items())))
    return os.path.join(CACHE, key + ".jxl")


def _run(threads, per_thread, paths, dump=None, warm=0, env=None, timeout=900):
    cmd = [EXE, str(threads), str(per_thread)]
    if dump:
        cmd += ["--dump", str(dump)]
    if warm:
        cmd += ["--warm", str(warm)]
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd + list(paths), check=True, capture_output=True, text=True, timeout=timeout, env=e)
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_threads_without_a_gpu_fail_loudly(built):
    """no CPU fallback behind the public API, whichever path a call takes: every image of every thread ends in "!gpu" """
    import j40_amd
    if j40_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    r = _run(6, 3, [_path("vardct", 264, 200, 11), _path("modular", 256, 256, 9)])
    assert r["errors"] == 18 and r["mismatches"] == 0 and r["file_errors"] == ["!gpu", "!gpu"]
    r = _run(1, 2, [_path("vardct", 264, 200, 11)], env={"J40HIP_SERVE": "1"})
    assert r["errors"] == 2 and r["file_errors"] == ["!gpu"]


def _check_against_reference(ref, dump, paths, expect_errors):
    for i, p in enumerate(paths):
        data = open(p, "rb").read()
        rerr, rpx = ref.decode(data)
        if expect_errors[i]:
            assert rerr == expect_errors[i]
            continue
        assert rerr == ""
        h, w = rpx.shape[:2]
        got = np.fromfile(os.path.join(dump, "%d_%dx%d.rgba" % (i, w, h)), np.uint8).reshape(h, w, 4)
        d = np.abs(got.astype(np.int16) - rpx.astype(np.int16))
        assert d.max() <= 1, (p, int(d.max()))   # bar: 1 level for VarDCT (in practice 0), Modular bit-exact
        if "modular" in os.path.basename(p):
            assert d.max() == 0, p


@pytest.mark.gpu
@pytest.mark.parametrize("serve", ["auto", "1", "0"])
def test_threads_on_a_mix_of_images_give_the_reference_pixels_and_codes(built, ref, tmp_path, serve):
    """16 threads x 6 images over VarDCT and Modular frames of several kinds, a damaged and a truncated stream among them: the
    batched path, the single-frame path inside the pipeline (Modular, extra channels, two passes) and the latency path
    (J40HIP_SERVE=0) all give the reference's pixels and 4-char codes"""
    paths = [_path("vardct", 520, 264, 41), _path("vardct", 520, 264, 42, bctx=1), _path("vardct", 1920, 1080, 7), _path("modular", 600, 300, 5, tree=1),
             _path("vardct", 392, 264, 9, passes=2), _path("vardct", 520, 264, 33, alpha=1), _path("modular", 256, 256, 101, alpha=1, prefix=1, lz77=1),
             _path("vardct", 776, 520, 3, maxlog=8, bctx=1, presets=2, orders=1)]
    damaged = bytearray(open(paths[1], "rb").read()); damaged[len(damaged) * 2 // 3] ^= 0x10
    short = open(paths[0], "rb").read()[:-90]
    for name, blob in (("damaged.jxl", bytes(damaged)), ("short.jxl", short)):
        p = tmp_path / name
        p.write_bytes(blob)
        paths.append(str(p))
    expect = [""] * 8 + [ref.decode(bytes(damaged))[0], ref.decode(short)[0]]
    assert expect[8] != "" and expect[9] == "shrt"
    r = _run(16, 6, paths, dump=tmp_path, env={} if serve == "auto" else {"J40HIP_SERVE": serve})
    assert r["mismatches"] == 0, r
    assert r["file_errors"] == expect, r
    _check_against_reference(ref, str(tmp_path), paths, expect)


@pytest.mark.gpu
def test_64_threads_over_8k_streams_through_the_public_api(built, ref, tmp_path):
    """64 caller threads, each decoding 8K VarDCT streams one after the other through the unchanged ten-function
    API; four distinct forward-encoded streams, every decode compared with the first of its stream and the first with the reference"""
    paths = [_path("vardct", 7680, 4320, 3 + 1000 * i, forward=1) for i in range(4)]
    r = _run(64, 3, paths, dump=tmp_path, warm=1, timeout=1800)
    assert r["errors"] == 0 and r["mismatches"] == 0 and r["warm_errors"] == 0 and r["warm_mismatches"] == 0, r
    assert r["images"] == 192
    _check_against_reference(ref, str(tmp_path), paths, [""] * 4)
    print("64 threads x 3 8K images through j40_next_frame: %.0f Mpx/s, latency median %.0f ms" % (r["mpixels_per_s"], r["latency_ms"]["median"]))
    assert r["mpixels_per_s"] > 2000, r   # (a floor far below what is measured: the 0.39 Gpx/s of one call at a time would fail it)


@pytest.mark.gpu
def test_pinned_plane_pool_is_bounded_evicts_the_oldest_and_expires(built):
    """the pool of pinned image planes behind j40_frame_pixels_u8x4 (runtime.hip: j40hip_pinned_acquire / _release; a drop-in caller
    never calls j40hip_shutdown): what sits idle never exceeds the bound, a new plane size displaces the planes idle longest instead
    of being pinned and unpinned per image, a plane of a size in the pool is reused, idle planes are unpinned after the idle time"""
    code = r'''
import ctypes as C, time, sys
sys.path.insert(0, %r)
import j40_amd
L = j40_amd.lib()
L.j40hip_pinned_acquire.restype = C.c_void_p; L.j40hip_pinned_acquire.argtypes = [C.c_size_t]
L.j40hip_pinned_release.argtypes = [C.c_void_p, C.c_size_t]
L.j40hip_pinned_pool_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
def stats():
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.j40hip_pinned_pool_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
MB = 1 << 20
assert stats() == (0, 0, 1 << 30)                       # J40HIP_PINNED_POOL_GB=1
big = [L.j40hip_pinned_acquire(300 * MB) for _ in range(4)]
assert all(big)
for p in big: L.j40hip_pinned_release(p, 300 * MB)
idle, planes, limit = stats()
assert idle <= limit and planes == 3, (idle, planes)    # the fourth did not fit: the OLDEST made room for it
again = L.j40hip_pinned_acquire(300 * MB)
assert again in big[1:] and stats()[1] == 2             # reused, not pinned anew
L.j40hip_pinned_release(again, 300 * MB)
small = [L.j40hip_pinned_acquire(200 * MB) for _ in range(3)]   # another image size: the old size's planes leave as these come back
for p in small: L.j40hip_pinned_release(p, 200 * MB)
idle, planes, limit = stats()
assert idle <= limit and idle >= 600 * MB, (idle, planes)
q = L.j40hip_pinned_acquire(200 * MB)
assert q in small
L.j40hip_pinned_release(q, 200 * MB)
time.sleep(1.3)                                          # J40HIP_PINNED_IDLE_S=1
r = L.j40hip_pinned_acquire(4096); L.j40hip_pinned_release(r, 4096)
idle, planes, limit = stats()
assert planes == 1 and idle == 4096, (idle, planes)     # everything older than a second was unpinned at this call
print("ok")
''' % ROOT
    env = dict(os.environ, J40HIP_PINNED_POOL_GB="1", J40HIP_PINNED_IDLE_S="1")
    import sys
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stderr[-2000:]
