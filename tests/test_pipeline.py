"""The whole-frame throughput pipeline (j40hip_pipeline_*, j40_amd/csrc/device/pipeline.hip) against the single-image public API
and the reference: same pixels, same error codes, whatever mix of images goes in (GPU only)."""
import numpy as np
import pytest

from streams import synth

pytestmark = pytest.mark.gpu


def _mix():
    items = [("vardct", 520, 264, s, {}) for s in (41, 42, 43, 44, 45)]
    items += [("vardct", 1920, 1080, 7, {}), ("vardct", 392, 264, 9, dict(passes=2)), ("vardct", 520, 264, 33, dict(alpha=1)),
              ("modular", 600, 300, 5, dict(tree=1)), ("modular", 300, 200, 6, dict(squeeze=1)), ("vardct", 776, 520, 3, dict(maxlog=8, bctx=1, presets=2, orders=1))]
    return [(w, h, synth(m, w, h, s, **o)) for m, w, h, s, o in items]


@pytest.mark.parametrize("device_output", [False, True])
def test_pipeline_matches_single_image_decodes(built, ref, device_output):
    import torch
    import j40_amd
    mix = _mix()
    damaged = bytearray(mix[1][2]); damaged[len(damaged) * 2 // 3] ^= 0x10
    mix.append((520, 264, bytes(damaged)))
    mix.append((520, 264, mix[0][2][: len(mix[0][2]) - 90]))   # truncated: "shrt"
    mix = mix * 3                                                # 39 images, several batches of 8
    pipe = j40_amd.Pipeline(device=0, host_threads=6, batch_frames=8, max_in_flight=2)
    outs, tickets = [], []
    for w, h, data in mix:
        o = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") if device_output else torch.zeros((h, w, 4), dtype=torch.uint8).pin_memory()
        outs.append(o)
        tickets.append(pipe.submit(data, o.data_ptr(), w * 4, device_output=device_output))
    pipe.drain()
    torch.cuda.synchronize()
    seen_errors = 0
    for (w, h, data), o, t in zip(mix, outs, tickets):
        err, expect = j40_amd.decode(data)
        assert pipe.result(t) == err, (w, h, err, pipe.result(t))
        if err == "":
            assert np.array_equal(o.cpu().numpy(), expect), (w, h)
            rerr, rexp = ref.decode(data)
            if rerr != "TODO":   # (the Squeeze image: the reference stops at its parameters, tests/test_squeeze.py)
                assert rerr == "" and np.abs(rexp.astype(int) - expect.astype(int)).max() <= 1
        else:
            seen_errors += 1
    assert seen_errors >= 3
    st = pipe.stats()
    assert st["completed"] == len(mix)
    pipe.close()


def test_pipeline_can_be_drained_and_reused(built):
    import torch
    import j40_amd
    pipe = j40_amd.Pipeline(device=0, host_threads=2, batch_frames=4, max_in_flight=1)
    for round_ in range(3):
        datas = [synth("vardct", 520, 264, 60 + round_ * 5 + i) for i in range(5)]
        outs = [torch.zeros((264, 520, 4), dtype=torch.uint8, device="cuda:0") for _ in datas]
        ts = [pipe.submit(d, o.data_ptr(), 520 * 4, device_output=True) for d, o in zip(datas, outs)]
        pipe.drain()
        torch.cuda.synchronize()
        for d, o, t in zip(datas, outs, ts):
            assert pipe.result(t) == ""
            assert np.array_equal(o.cpu().numpy(), j40_amd.decode(d)[1])
    pipe.close()


def test_lf_group_tail_on_the_device_equals_the_host_tail(built):
    """j40hip_frame_parse_ex(flags = 1) leaves dequantisation, adaptive smoothing and the LLF coefficients of the LfGroups to the
    device (lf_tail_kernels.hip, run at upload); the pixels must be those of the frame whose tail the host computed, bit for bit --
    over every transform size (the large ones take the LDS kernel), several LfGroups, and with smoothing off"""
    import ctypes as C
    import torch
    import j40_amd
    L = j40_amd.lib()
    for w, h, opts in [(776, 520, dict(maxlog=8, bctx=1, presets=2, orders=1)), (2600, 2100, dict()), (520, 264, dict(fullheader=1, nosmooth=1)), (7680, 4320, dict())]:
        data = synth("vardct", w, h, 23, **opts)
        err, expect = j40_amd.decode(data)
        assert err == ""
        buf = C.create_string_buffer(data, len(data))
        e = C.c_uint32()
        fh = L.j40hip_frame_parse_ex(buf, len(data), 1, 1, C.byref(e))
        assert fh and e.value == 0
        assert L.j40hip_frame_upload(fh, 0) == 0
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
        assert L.j40hip_frame_decode(fh, out.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert L.j40hip_frame_status(fh) == 0
        assert np.array_equal(out.cpu().numpy(), expect), (w, h, opts)
        L.j40hip_frame_free(fh)


LF_DEVICE_CASES = [
    ("vardct", 776, 520, 31, dict()),
    ("vardct", 2600, 2100, 32, dict(bctx=1)),                  # four LfGroup sections, custom LF thresholds (LF index)
    ("vardct", 2049, 300, 33, dict(maxlog=8, cfl=1)),          # a 1-cell-wide second LfGroup, 256x256 transforms
    ("vardct", 1920, 1080, 34, dict(forward=1)),
    ("vardct", 520, 264, 35, dict(alpha=1)),
    ("vardct", 520, 264, 36, dict(passes=3)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,w,h,seed,opts", LF_DEVICE_CASES)
def test_lf_group_streams_on_the_device_equal_the_host_parse(built, mode, w, h, seed, opts):
    """j40hip_frame_parse_on: LF coefficients and HF metadata of every LfGroup section decoded by k_lf_groups; everything the
    host derives from them (LF index, varblock placement, LLF coefficients) and the decoded pixels must equal the host parse's"""
    import j40_amd
    data = synth(mode, w, h, seed, **opts)
    host, dev = j40_amd.Frame(data), j40_amd.Frame(data, lf_device=0)
    assert dev.lf_on_device() and not host.lf_on_device()
    assert host.info == dev.info
    for gg in range(host.info["num_lf_groups"]):
        assert host.lf_group_info(gg) == dev.lf_group_info(gg)
        for which in range(4):
            assert np.array_equal(host.plane(gg, which), dev.plane(gg, which)), (gg, which)
        for a, b in zip(host.varblocks(gg), dev.varblocks(gg)):
            assert np.array_equal(a, b)
        for c in range(3):
            assert np.array_equal(host.llf(gg, c), dev.llf(gg, c))
    for f in (host, dev):
        f.upload(0)
    (ea, pa), (eb, pb) = host.decode_to_host(), dev.decode_to_host()
    assert ea == "" and eb == "" and np.array_equal(pa, pb)
    host.close(); dev.close()


@pytest.mark.gpu
def test_lf_group_streams_on_the_device_report_damage_like_the_host(built, ref):
    import j40_amd
    data = synth("vardct", 2600, 2100, 41)
    fr = j40_amd.Frame(data)
    # the LfGroup sections lie between LfGlobal / HfGlobal and the pass groups: flip bits in the first fifth of the stream
    sizes = fr.section_sizes()
    fr.close()
    start, end = 200, len(data) - int(sizes.sum())
    rng = np.random.default_rng(3)
    seen = set()
    for _ in range(40):
        m = bytearray(data)
        m[int(rng.integers(start, end))] ^= 1 << int(rng.integers(0, 8))
        outcomes = []
        for lf_device in (None, 0):
            try:
                f = j40_amd.Frame(bytes(m), lf_device=lf_device)
                outcomes.append(("", [f.plane(g, 0).tobytes() for g in range(f.info["num_lf_groups"])]))
                f.close()
            except j40_amd.J40Error as e:
                outcomes.append((e.code, None))
        assert outcomes[0] == outcomes[1]
        assert outcomes[0][0] == ref.decode(bytes(m))[0] or outcomes[0][0] == ""   # (errors behind the LfGroups surface at decode time)
        seen.add(outcomes[0][0])
    assert len(seen) >= 2, "the damage should have hit some LfGroup section"
