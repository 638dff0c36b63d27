"""Streams made by the generator's forward path (tools/jxlsynth forward=1, tools/jxlsynth_forward.hpp): an encode of a procedural
picture at about distance 1 -- what bench.py decodes.

The analysis transforms are the decoder's synthesis transforms inverted numerically, so the first thing to pin is that the
UNMODIFIED REFERENCE decodes such a stream back to the source picture (PSNR in the range a distance-1 encode gives); then the
usual parity: CPU checkers and the HIP path against the reference's pixels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from streams import synth, ROOT, CACHE

SYNTH = os.path.join(ROOT, "build", "jxlsynth")


def forward_stream(w, h, seed, tmp_path, **opts):
    out, src = str(tmp_path / "f.jxl"), str(tmp_path / "f.rgb")
    r = subprocess.run([SYNTH, "vardct", str(w), str(h), str(seed), out, "forward=1", "dumpsrc=" + src] + ["%s=%s" % kv for kv in sorted(opts.items())],
                       check=True, capture_output=True, text=True)
    bpp = float((r.stdout + r.stderr).split("(")[1].split(" bpp")[0])
    return open(out, "rb").read(), np.fromfile(src, np.uint8).reshape(h, w, 3), bpp


def psnr(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("w,h,opts", [(776, 520, dict()), (520, 264, dict(detail=3, beta=0.15)), (1000, 600, dict(maxlog=5)), (521, 263, dict(detail=1))])
def test_reference_decodes_a_forward_stream_to_its_source_picture(built, ref, tmp_path, w, h, opts):
    data, src, bpp = forward_stream(w, h, 9, tmp_path, **opts)
    err, px = ref.decode(data)
    assert err == ""
    q = psnr(px[..., :3], src)
    assert 34.0 < q < 52.0, "PSNR %.2f dB at %.3f bpp: not what a distance-1 encode of this picture gives" % (q, bpp)
    assert 0.2 < bpp < 6.0
    assert np.all(px[..., 3] == 255)


def test_more_detail_costs_more_bits_and_sections_differ_in_length(built, tmp_path):
    import j40_amd
    calm, _, bpp_calm = forward_stream(1920, 1080, 5, tmp_path, detail=1)
    busy, _, bpp_busy = forward_stream(1920, 1080, 5, tmp_path, detail=3, beta=0.15)
    assert bpp_busy > 1.5 * bpp_calm
    fr = j40_amd.Frame(busy)
    sizes = fr.section_sizes()
    fr.close()
    assert sizes.max() > 1.4 * sizes.mean(), "the picture's calm and busy regions should show in the sections' sizes"


def test_forward_stream_through_the_cpu_checkers(built, ref, tmp_path):
    data, _, _ = forward_stream(776, 520, 4, tmp_path)
    rerr, expect = ref.decode(data)
    assert rerr == ""
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    buf = C.create_string_buffer(data, len(data))
    got = np.zeros((520, 776, 4), np.uint8)
    assert S.hostsim_decode(buf, len(data), got.ctypes.data, None, 0) == 0
    assert np.abs(got.astype(np.int32) - expect.astype(np.int32)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,opts", [(776, 520, dict()), (1920, 1080, dict()), (1920, 1080, dict(detail=3, beta=0.15)), (2600, 2100, dict(maxlog=5)), (521, 263, dict())])
def test_forward_streams_on_the_gpu_match_the_reference(built, ref, w, h, opts):
    import j40_amd
    assert j40_amd.device_count() > 0
    data = synth("vardct", w, h, 17, forward=1, **opts)
    err, rgba = j40_amd.decode(data)
    rerr, expect = ref.decode(data)
    assert err == "" and rerr == ""
    d = np.abs(rgba.astype(np.int32) - expect.astype(np.int32))
    assert d.max() <= 1 and int((d > 0).sum()) <= rgba.size // 10000 + 4
