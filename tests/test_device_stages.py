"""The pipeline's device stages pinned ON THE DEVICE against the reference's internals: one image goes through
the enqueue a batch takes (j40hip_stage_dump_*, j40_amd/csrc/device/async.hip) and what each stage left in HBM is compared with
what the unmodified reference holds after reading the same sections (RefStage = j40__lf_group_st, j40.h:6360-6390):

  k_lf_lanes            chroma-from-luma maps, sharpness map (plain copies of its output planes); the LF integers through the LF index
                        of EVERY cell (j40.h:6566-6570) and through the LLF coefficients
  k_plan_place          varblock placement: the block map in the reference's encoding, varblock count, coeffoff_qfidx, HfMul
                        (j40.h:6634-6701)
  k_plan_scan / _emit   the entropy kernel's block lists per group in j40__hf_coeffs' visiting order with the three block contexts
                        (j40.h:6907-6915, 6951-6953), the pixel kernels' work list (geometry, multipliers: j40.h:7078, 7145-7146)
  LfGroup tail (batch)  LLF coefficients, bit for bit (j40.h:6544-6590, 6492, 5944)
  k_plan_verdict        the first failing section's code on damaged LfGroup sections = the reference's error

GPU only (the CPU builds of the same device functions are compared with the host path in tests/test_device_plan.py)."""
import numpy as np
import pytest

from streams import synth

pytestmark = pytest.mark.gpu

# DctSelect -> (log2 rows, log2 columns, coefficient-order index), written out from the format's table (j40.h:4591-4600), NOT read
# from the product
DCT = {0: (3, 3, 0), 1: (3, 3, 1), 2: (3, 3, 1), 3: (3, 3, 1), 4: (4, 4, 2), 5: (5, 5, 3), 6: (4, 3, 4), 7: (3, 4, 4), 8: (5, 3, 5), 9: (3, 5, 5),
       10: (5, 4, 6), 11: (4, 5, 6), 12: (3, 3, 1), 13: (3, 3, 1), 14: (3, 3, 1), 15: (3, 3, 1), 16: (3, 3, 1), 17: (3, 3, 1), 18: (6, 6, 7),
       19: (6, 5, 8), 20: (5, 6, 8), 21: (7, 7, 9), 22: (7, 6, 10), 23: (6, 7, 10), 24: (8, 8, 11), 25: (8, 7, 12), 26: (7, 8, 12)}

CASES = [
    ("vardct", 776, 520, 31, dict()),
    ("vardct", 2600, 2100, 32, dict(bctx=1)),                  # four LfGroup sections, custom LF thresholds and block-context map
    ("vardct", 2049, 300, 33, dict(maxlog=8, cfl=1)),          # a 1-cell-wide second LfGroup, 256x256 transforms, chroma-from-luma maps
    ("vardct", 1920, 1080, 34, dict(forward=1)),
    ("vardct", 520, 264, 36, dict(passes=3)),
    ("vardct", 4100, 2100, 37, dict(presets=2, orders=1)),     # 3 x 2 LfGroups
    ("vardct", 776, 520, 3, dict(maxlog=8, bctx=1, presets=2, orders=1)),
    ("vardct", 7680, 4320, 3, dict(forward=1)),                # the bench stream
    ("vardct", 2600, 2100, 51, dict(lftree=1)),                # LfGroup channels under subtrees of sample properties, predictors reaching NE, NEE, NN, NWW
    ("vardct", 2600, 2100, 61, dict(lftree=2)),                # one test over two like leaves per channel: k_lf_rows' straight-line step on other properties / predictors than the default tree's
    ("vardct", 2049, 300, 64, dict(lftree=3, forward=1)),
]


def compare(ref, data, lf_on_device):
    import j40_amd
    from refdec import RefStage
    sd = j40_amd.StageDump(data, 0, lf_on_device)
    rs = RefStage(ref, data)
    assert sd.verdict == "" and (sd.flags & 3) == 0
    assert sd.lf_on_device == lf_on_device
    assert sd.num_lf_groups == rs.info["num_lf_groups"] and sd.num_groups == rs.info["num_groups"]
    assert (sd.width, sd.height) == (rs.info["width"], rs.info["height"])
    bmap = rs.block_ctx_map()
    nb_qf1 = rs.info["nb_qf_thr"] + 1
    lfidx_size = (rs.info["nb_lf_thr0"] + 1) * (rs.info["nb_lf_thr1"] + 1) * (rs.info["nb_lf_thr2"] + 1)
    gcolumns, ggcolumns = (sd.width + 255) // 256, (sd.width + 2047) // 2048
    expect_blocks = {}   # frame-wide group index -> [(coeffoff_qfidx, pos_dct, bctx3)] in visiting order
    expect_sorted = []   # (dctsel, lf group, varblock) -> geometry
    total = 0
    for gg in range(sd.num_lf_groups):
        gi, ri = sd.lf_group_info(gg), rs.lf_group_info(gg)
        assert gi["status"] == ""
        assert {k: gi[k] for k in ri} == ri, (gg, gi, ri)
        rblocks, rlfi = rs.plane(gg, 0), rs.plane(gg, 1)
        assert np.array_equal(sd.plane(gg, 0), rblocks), "block map of LfGroup %d" % gg
        assert np.array_equal(sd.plane(gg, 1), rlfi), "LF index of LfGroup %d" % gg
        assert np.array_equal(sd.plane(gg, 2), rs.plane(gg, 2)) and np.array_equal(sd.plane(gg, 3), rs.plane(gg, 3)), "chroma-from-luma maps of LfGroup %d" % gg
        if lf_on_device:
            assert np.array_equal(sd.plane(gg, 4), rs.plane(gg, 4)), "sharpness map of LfGroup %d" % gg
        ca, cb = rs.varblocks(gg)
        da, db, dxyz = sd.varblocks(gg)
        assert np.array_equal(ca, da), "coeffoff_qfidx of LfGroup %d" % gg
        assert np.array_equal(cb.view(np.uint32), db.view(np.uint32)), "hfmul.inv of LfGroup %d" % gg
        for c in range(3):
            assert np.array_equal(rs.llf(gg, c).view(np.uint32), sd.llf(gg, c).view(np.uint32)), "LLF coefficients of LfGroup %d channel %d" % (gg, c)
        # what plan_emit must have written, derived from the REFERENCE's planes: top-left cells in raster order per group
        ggx, ggy = gg % ggcolumns, gg // ggcolumns
        ys, xs = np.nonzero((rblocks >> 20) >= 2)
        order = np.lexsort((xs, ys))
        mult_base = np.float32(65536.0) / np.float32(rs.info["global_scale"])
        for y8, x8 in zip(ys[order].tolist(), xs[order].tolist()):
            cell = int(rblocks[y8, x8])
            dctsel, v = (cell >> 20) - 2, cell & 0xfffff
            lr, lc, oidx = DCT[dctsel]
            gid = (ggy * 8 + (y8 >> 5)) * gcolumns + ggx * 8 + (x8 >> 5)
            cq = int(ca[v]) & 0xffffffff
            b0 = (oidx * nb_qf1 + (cq & 15)) * lfidx_size + int(rlfi[y8, x8])
            b3 = sum((int(bmap[b0 + 13 * nb_qf1 * lfidx_size * k]) & 15) << (4 * k) for k in range(3))
            expect_blocks.setdefault(gid, []).append((cq, ((y8 & 31) * 32 + (x8 & 31)) | (dctsel << 10), b3))
            expect_sorted.append((dctsel, gg, v, ri["left"] + x8 * 8, ri["top"] + y8 * 8, min(1 << lc, ri["width"] - x8 * 8), min(1 << lr, ri["height"] - y8 * 8),
                                  np.float32(mult_base * cb[v])))
        total += ri["nb_varblocks"]
    assert sd.num_varblocks == total
    for g in range(sd.num_groups):
        got = [tuple(int(v) for v in row) for row in sd.group_blocks(g)]
        assert got == expect_blocks.get(g, []), "block list of group %d" % g
    ints, floats, class_start = sd.sorted_varblocks()
    assert class_start[27] == total == len(ints)
    expect_sorted.sort(key=lambda e: e[:3])    # by DctSelect, then LfGroup, then varblock: plan_build.cpp's order
    blk_seen = np.zeros(total, bool)
    for k, e in enumerate(expect_sorted):
        px, py, effw, effh, dctsel, blk = (int(v) for v in ints[k][:6])
        assert (dctsel, px, py, effw, effh) == (e[0], e[3], e[4], e[5], e[6]), (k, e, ints[k])
        assert np.float32(floats[k][0]).view(np.uint32) == e[7].view(np.uint32), (k, floats[k][0], e[7])
        assert not blk_seen[blk]
        blk_seen[blk] = True
    for d in range(27):
        assert class_start[d] == sum(1 for e in expect_sorted if e[0] < d)
    err, expect = ref.decode(data)
    assert err == "" and np.abs(sd.rgba().astype(np.int16) - expect.astype(np.int16)).max() <= 1
    rs.close()
    sd.close()


@pytest.mark.parametrize("mode,w,h,seed,opts", CASES)
@pytest.mark.parametrize("lf_on_device", [True, False])
def test_device_stages_equal_the_reference_internals(built, ref, mode, w, h, seed, opts, lf_on_device):
    compare(ref, synth(mode, w, h, seed, **opts), lf_on_device)


@pytest.mark.parametrize("lf_on_device", [True, False])
def test_damaged_lf_sections_get_the_reference_verdict_from_the_device_stages(built, ref, lf_on_device):
    """one flipped bit inside the LfGroup sections: the LF lane decoder (or the host decoder feeding the same stages), the placement
    and k_plan_verdict must end with the reference's 4-char code -- or leave the frame to the single-frame path (an LfGroup header
    the lane decoder does not take), never with pixels for a stream the reference rejects"""
    import j40_amd
    data = synth("vardct", 2600, 2100, 41)
    fr = j40_amd.Frame(data)
    lf_end = len(data) - sum(fr.section_sizes())
    fr.close()
    rng = np.random.default_rng(21)
    agreed, rejected = 0, 0
    for _ in range(40):
        m = bytearray(data)
        m[int(rng.integers(200, lf_end))] ^= 1 << int(rng.integers(0, 8))
        rerr, _ = ref.decode(bytes(m))
        try:
            sd = j40_amd.StageDump(bytes(m), 0, lf_on_device)
        except j40_amd.J40Error as e:
            assert e.code == "TODO"     # the front parse gave the frame up: the pipeline sends it down the single-frame path
            continue
        if sd.flags & 1:                 # lffb: ditto, decided on the device
            sd.close()
            continue
        assert sd.verdict == rerr, (sd.verdict, rerr)
        agreed += 1
        rejected += rerr != ""
        sd.close()
    assert agreed >= 25 and rejected >= 10, (agreed, rejected)


@pytest.mark.parametrize("env", [{"J40HIP_PLAN_PLACE_FORM": "1"}, {"J40HIP_PLAN_PLACE_LANES": "1"}], ids=["walk", "lanes"])
def test_the_placement_kernels_earlier_forms_give_the_same_products(built, env):
    """k_plan_place_walk (everything inside the serial walk) and k_plan_place_lanes (an LfGroup per lane) stay in the library for
    comparisons; the form is read once per process, so each runs the comparison above -- three streams, both damaged-section
    cases -- in a process of its own"""
    import os, subprocess, sys
    from streams import ROOT
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_device_stages.py"),
                          "-k", "(equal_the_reference_internals and (2600-2100-32 or 2049-300-33 or 776-520-3-)) or damaged"],
                         cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    assert " passed" in run.stdout and "failed" not in run.stdout, run.stdout[-1000:]
