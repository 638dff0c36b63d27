"""The plan build that the pipeline runs on the device (j40_amd/csrc/device/plan_dev.h: varblock placement, prefix sums, per-varblock
records for K1 and K2; j40_amd/csrc/plan_front.cpp: everything that does not depend on the LfGroup sections), compiled for the CPU
(tests/hostsim) and compared array by array with the host path (frame.cpp lf_group_finish + plan_build.cpp), whose products
tests/test_host.py pins against the reference's internals (j40.h:6585-6720, 6722-6790). On damaged LfGroup sections the device
path's verdict -- first failing LfGroup section in file order -- must be the host parse's error code. The kernels that run these
functions are checked on the GPU (tests/test_pipeline.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from streams import synth, ROOT, VARDCT_CASES

CASES = [("vardct", 520, 264, 100 + i, opts) for i, (_, opts) in enumerate(VARDCT_CASES)] + [
    ("vardct", 776, 520, 31, dict()),
    ("vardct", 2600, 2100, 32, dict(bctx=1)),                  # four LfGroup sections, custom LF thresholds (LF index)
    ("vardct", 2049, 300, 33, dict(maxlog=8, cfl=1)),          # a 1-cell-wide second LfGroup, 256x256 transforms
    ("vardct", 776, 520, 3, dict(maxlog=8, bctx=1, presets=2, orders=1)),
    ("vardct", 1920, 1080, 34, dict(forward=1)),
    ("vardct", 2600, 2100, 35, dict(forward=1)),
    ("vardct", 520, 264, 36, dict(passes=3)),
    ("vardct", 4100, 2100, 37, dict()),                        # 3 x 2 LfGroups
    ("vardct", 2600, 2100, 51, dict(lftree=1)),                # every LfGroup channel under a subtree of sample properties; predictors reaching NE, NEE, NN, NWW
    ("vardct", 520, 264, 52, dict(lftree=1, cfl=1)),
    ("vardct", 2049, 300, 53, dict(lftree=1, forward=1)),      # (with a 1-cell-wide LfGroup)
    ("vardct", 2600, 2100, 61, dict(lftree=2)),                # every LfGroup channel one test over two leaves that predict alike: the lane decoder's straight-line step
    ("vardct", 2600, 2100, 62, dict(lftree=3)),                # ... with the other properties and predictors it takes
    ("vardct", 520, 264, 63, dict(lftree=2, cfl=1)),
    ("vardct", 2049, 300, 64, dict(lftree=3, forward=1)),
]


@pytest.fixture(scope="module")
def sim(built):
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_device_plan_check.restype = C.c_int32
    S.hostsim_device_plan_check.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    return S


def check(sim, data):
    buf = C.create_string_buffer(data, len(data))
    err = C.c_uint32()
    return sim.hostsim_device_plan_check(buf, len(data), C.byref(err)), err.value


@pytest.mark.parametrize("mode,w,h,seed,opts", CASES)
def test_device_plan_equals_host_plan(sim, mode, w, h, seed, opts):
    data = synth(mode, w, h, seed, **opts)
    rc, err = check(sim, data)
    assert err == 0
    if opts.get("alpha"):
        assert rc == -1          # extra channels: Modular sub-images behind the coefficients, left to the host path
    else:
        assert rc == 0, "check %d failed" % rc


@pytest.mark.parametrize("threads", [2, 5, 12])
def test_plan_built_by_a_team_of_threads_equals_the_plan_of_one(built, threads):
    """build_vardct_plan's frame-wide arrays (LF bundle, K1's block lists, K2's sorted work list) written by a team of threads -- what
    the single-image path of j40_next_frame does with the threads it parsed with -- are byte for byte those of the calling thread
    alone, on frames from one LfGroup to the 8K bench frame's twelve"""
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_plan_threads_check.restype = C.c_int32
    S.hostsim_plan_threads_check.argtypes = [C.c_void_p, C.c_size_t, C.c_int32]
    cases = CASES[-8:] + [("vardct", 7680, 4320, 3, dict(forward=1)), ("vardct", 520, 264, 41, dict(alpha=1))]
    for mode, w, h, seed, opts in cases:
        data = synth(mode, w, h, seed, **opts)
        buf = C.create_string_buffer(data, len(data))
        for _ in range(3 if w < 3000 else 1):   # (a race would not show every time)
            assert S.hostsim_plan_threads_check(buf, len(data), threads) == 0, (w, h, opts)


def test_damaged_lf_sections_get_the_host_parse_verdict(sim):
    data = synth("vardct", 2600, 2100, 41)
    rng = np.random.default_rng(11)
    outcomes = {}
    for _ in range(80):
        m = bytearray(data)
        m[int(rng.integers(150, len(m) // 6))] ^= 1 << int(rng.integers(0, 8))
        rc, err = check(sim, bytes(m))
        assert rc in (0, -1), (rc, err)
        outcomes[err] = outcomes.get(err, 0) + (rc == 0)
    assert len(outcomes) >= 3 and sum(outcomes.values()) >= 40, outcomes


@pytest.fixture(scope="module")
def lanes(built):
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_lf_lanes_check.restype = C.c_int32
    S.hostsim_lf_lanes_check.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    return S


def lanes_check(S, data):
    buf = C.create_string_buffer(data, len(data))
    n, bad = C.c_int32(), C.c_int32()
    return S.hostsim_lf_lanes_check(buf, len(data), C.byref(n), C.byref(bad)), n.value, bad.value


@pytest.mark.parametrize("mode,w,h,seed,opts", CASES)
def test_lf_lane_decoder_equals_the_host_decoder(lanes, mode, w, h, seed, opts):
    """device/lf_lanes_dev.h (one LfGroup section per wavefront lane: k_lf_lanes) compiled for the CPU, every section of every case
    against frame.cpp's read_lf_group_raw (whose products tests/test_host.py pins against the reference's j40__lf_group): LF
    integers, chroma-from-luma maps, varblock count and the two rows of the varblock-info channel"""
    rc, n, bad = lanes_check(lanes, synth(mode, w, h, seed, **opts))
    if opts.get("alpha"):
        assert rc == -1          # extra channels: the front plan leaves the frame to the host path
    else:
        assert rc == 0 and n >= 1 and bad == 0, (rc, n, bad)


def test_lf_lane_decoder_fails_like_the_host_decoder(lanes):
    data = synth("vardct", 2600, 2100, 41)
    rng = np.random.default_rng(12)
    failed = 0
    for _ in range(80):
        m = bytearray(data)
        m[int(rng.integers(150, len(m) // 6))] ^= 1 << int(rng.integers(0, 8))
        rc, n, bad = lanes_check(lanes, bytes(m))
        assert rc in (0, -1), rc
        failed += bad
    assert failed >= 20


def rows_check(S, data, lanes):
    S.hostsim_lf_rows_check.restype = C.c_int32
    S.hostsim_lf_rows_check.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    buf = C.create_string_buffer(data, len(data))
    n, bad = C.c_int32(), C.c_int32()
    return S.hostsim_lf_rows_check(buf, len(data), lanes, C.byref(n), C.byref(bad)), n.value, bad.value


@pytest.mark.parametrize("mode,w,h,seed,opts", CASES)
def test_lf_row_window_decoder_equals_the_host_decoder(lanes, mode, w, h, seed, opts):
    """device/lf_rows_dev.h (k_lf_rows: a section per lane, tree + alias tables + a 256-sample window per lane in LDS, finished rows
    copied out in one piece) compiled for the CPU with the tables staged as the kernel stages them, the frame's sections stepped in
    lockstep as a wavefront's lanes are: same planes, varblock counts and status as frame.cpp's read_lf_group_raw (pinned against the
    reference's j40__lf_group by tests/test_host.py). The varblock-info channel (2 rows of hundreds to thousands of samples) is the
    case of rows wider than the window."""
    lanes.hostsim_lf_rows_counts.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.c_int32]
    # (mode bit 0: every sample through the general step; bit 2: no channel left as residuals for k_lf_predict -- the
    # default leaves every leaf-only channel, predicted afterwards by lf_predict_section_serial, the kernel's arithmetic in stream order)
    lanes.hostsim_lf_rows_raw_channels.restype = C.c_int64
    raw_before = lanes.hostsim_lf_rows_raw_channels()
    for nl, general_only in ((1, 0), (64, 0), (3, 1), (64, 4), (7, 5)):
        lanes.hostsim_lf_rows_counts(None, None, 1, general_only)
        rc, n, bad = rows_check(lanes, synth(mode, w, h, seed, **opts), nl)
        plain, general = C.c_int64(), C.c_int64()
        lanes.hostsim_lf_rows_counts(C.byref(plain), C.byref(general), 1, 0)
        if opts.get("alpha"):
            assert rc == -1
            continue
        assert rc == 0 and n >= 1 and bad == 0, (rc, n, bad, nl)
        # the straight-line step (lf_row_step_plain) takes nearly every sample of the trees it accepts -- the generator's default LF
        # tree and lftree=2/3 --, none of the others (lftree=1); with it switched off the general step alone gives the same planes
        if (general_only & 1) or opts.get("lftree") == 1:
            assert plain.value == 0 and general.value > 0
        else:
            assert plain.value > 20 * general.value, (plain.value, general.value)
    if not opts.get("alpha") and not opts.get("lftree"):   # the default LF tree: X, B and the HF metadata channels hang under single leaves
        assert lanes.hostsim_lf_rows_raw_channels() > raw_before


def test_lf_fast_entries_equal_the_symbol_decoder_on_random_configurations(lanes):
    """lf_rows_fast_entry (the alias entry and the hybrid integer worked out per symbol: what k_lf_rows' straight-line step reads) against
    lane_symbol_in_cluster on random entries, every split_exp / msb_in_token / lsb_in_token combination the format allows, max_token
    anywhere, random states and bit windows: same value, same next state, same bits taken; a token above max_token is the entry's flag"""
    lanes.hostsim_lf_fast_entry_check.restype = C.c_int64
    lanes.hostsim_lf_fast_entry_check.argtypes = [C.c_uint32, C.c_int32]
    total = 0
    for seed in range(8):
        n = lanes.hostsim_lf_fast_entry_check(seed, 200000)
        assert n > 0, (seed, n)
        total += n
    assert total > 400000, total


def test_lf_row_window_decoder_fails_like_the_host_decoder(lanes):
    data = synth("vardct", 2600, 2100, 41)
    rng = np.random.default_rng(12)
    failed = 0
    for _ in range(80):
        m = bytearray(data)
        m[int(rng.integers(150, len(m) // 6))] ^= 1 << int(rng.integers(0, 8))
        rc, n, bad = rows_check(lanes, bytes(m), 4)
        assert rc in (0, -1), rc
        failed += bad
    assert failed >= 20


def test_lanes_take_the_groups_by_decreasing_section_size(built):
    """DevPlan::lane_order (plan_front.cpp): a permutation of the frame's groups, section bytes (summed over the passes) never
    increasing along it, ties in group order -- so that the 64 lanes of a k_hf_lanes wavefront decode sections of about one length"""
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_lane_order.restype = C.c_int32
    S.hostsim_lane_order.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32]
    for (w, h, seed, opts) in [(1920, 1080, 31, dict(forward=1)), (1300, 776, 32, dict(passes=3)), (520, 264, 33, dict())]:
        data = synth("vardct", w, h, seed, **opts)
        buf = C.create_string_buffer(data, len(data))
        order, size = np.zeros(4096, np.uint32), np.zeros(4096, np.uint64)
        n = S.hostsim_lane_order(buf, len(data), order.ctypes.data, size.ctypes.data, 4096)
        assert n == ((w + 255) // 256) * ((h + 255) // 256)
        order, size = order[:n], size[:n]
        assert sorted(order.tolist()) == list(range(n))
        along = size[order]
        assert np.all(along[:-1] >= along[1:])
        for a, b in zip(range(n - 1), range(1, n)):
            if along[a] == along[b]:
                assert order[a] < order[b]


def test_lf_row_window_decoder_specialised_steps_all_ran(lanes):
    """lf_row_step_plain_needs: the plain step instantiated per combination of what a wavefront's lanes need (a test or none, which
    predictions, multipliers); the cases above and below run at least eight different combinations through it, each against the host
    decoder's planes"""
    for (w, h, seed, opts) in [(2600, 2100, 61, dict(lftree=2)), (2600, 2100, 62, dict(lftree=3)), (1920, 1080, 34, dict(forward=1)), (520, 264, 63, dict(lftree=2, cfl=1))]:
        for general_only in (4, 0):   # every channel predicted by its lane (the combinations of predictions), then the default
            lanes.hostsim_lf_rows_counts(None, None, 1, general_only)
            for nl in (1, 2, 64):
                rc, n, bad = rows_check(lanes, synth("vardct", w, h, seed, **opts), nl)
                assert rc == 0 and bad == 0
    seen = (C.c_int64 * 32)()
    lanes.hostsim_lf_rows_needs_seen(seen)
    used = [i for i in range(32) if seen[i]]
    assert len(used) >= 8, used



def test_lf_row_window_decoder_on_damaged_streams_of_every_tree(lanes):
    """tools/lf_rows_sweep.py, a short run: random sizes, the generator's four LF trees, bit flips inside the LfGroup sections; the lane
    decoder with its residual channels predicted afterwards (k_lf_predict's arithmetic in stream order) ends every section with the
    host decoder's code -- also where a sample of a residual channel leaves the range before whatever stopped the lane -- and the same planes"""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lf_rows_sweep.py"), "60", "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-400:] + out.stderr[-400:]
