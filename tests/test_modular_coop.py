"""The wave-cooperative form of the Modular section kernel (j40_amd/csrc/device/modular_coop.hip).

On the CPU: the tree layout it decodes with (DevCoopTree, built by plan_build.cpp) selects the leaf a walk of the MA tree reaches,
for random property vectors; which streams it takes. On the GPU: bit-exact against the unmodified reference, block-boundary widths,
damaged streams, and the same answers as k_modular_sections (J40HIP_NO_COOP=1 in a child process)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from streams import synth, ROOT

# (width, height, options): all of them decode through k_modular_coop (rANS, no LZ77, no weighted predictor, <= 64 leaves)
COOP_CASES = [
    (600, 300, dict(tree=5)),                              # 48 leaves, properties 0..14, 13 predictors, offsets, multipliers
    (601, 299, dict(tree=5, alpha=1)),
    (64, 40, dict(tree=5)), (65, 40, dict(tree=5)), (127, 9, dict(tree=5)), (128, 5, dict(tree=5)), (129, 3, dict(tree=5)),
    (1, 1, dict(tree=5)), (2, 7, dict(tree=5)), (3, 2, dict(tree=1)), (1, 200, dict(tree=5)), (300, 1, dict(tree=5)),
    (520, 520, dict(tree=5, groupshift=7)),
    (1100, 700, dict(tree=5, groupshift=10)),              # one 1024-wide group and its neighbours: 17 blocks of 64 columns per row
    (700, 520, dict(tree=5, groupshift=9)),
    (600, 300, dict(tree=1, localtree=1)),                 # every other group: a tree and code spec of its own
    (600, 300, dict(tree=5, passes=2)),
    (600, 300, dict(tree=5, localrct=4, alpha=1)),
    (300, 200, dict(tree=5, bpp=12, rct=13)),
    (600, 300, dict(tree=5, palette=1)),                   # meta channel in LfGlobal
    (256, 256, dict(tree=0)),                              # a single leaf: no branch at all
]


def _hostsim():
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_coop_check.restype = C.c_int32
    S.hostsim_coop_check.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32]
    return S


@pytest.mark.parametrize("w,h,opts", COOP_CASES[:3] + COOP_CASES[12:])
def test_coop_tree_selects_the_leaf_the_walk_reaches(built, w, h, opts):
    import j40_amd
    data = synth("modular", w, h, 7, **opts)
    fr = j40_amd.Frame(data)
    coop, total = fr.coop_sections()
    split = fr.split_sections()
    fr.close()
    # (a tree without a branch looks at nothing: the two-pass decoder takes its sections, modular_split.hip)
    assert coop + split == total and total >= 1 and (split == 0 or opts.get("tree") == 0), "every section of these streams is the cooperative kernel's"
    buf = C.create_string_buffer(data, len(data))
    checked = _hostsim().hostsim_coop_check(buf, len(data), 99, 20000)
    assert checked >= (1 if coop else 0), "mismatches: %d" % (-checked - 1)


def test_what_the_cooperative_kernel_leaves_to_the_general_one(built):
    import j40_amd
    for opts, expect_coop in [(dict(tree=2), False), (dict(tree=5, prefix=1), False), (dict(tree=5, lz77=1), False), (dict(tree=3, alpha=1), False),
                              (dict(tree=5), True), (dict(tree=1, localtree=2), None)]:
        fr = j40_amd.Frame(synth("modular", 600, 300, 7, **opts))
        coop, total = fr.coop_sections()
        fr.close()
        if expect_coop is None:
            assert 0 < coop < total, "local trees with the weighted predictor stay with k_modular_sections, the others do not"
        else:
            assert (coop == total) if expect_coop else (coop == 0), (opts, coop, total)


@pytest.fixture(scope="module")
def gpu(built):
    import j40_amd
    assert j40_amd.device_count() > 0, "the gpu tests need a HIP device"
    return j40_amd


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,opts", COOP_CASES)
def test_coop_bit_exact_with_the_reference(gpu, ref, w, h, opts):
    data = synth("modular", w, h, 71, **opts)
    err, rgba = gpu.decode(data)
    assert err == ""
    rerr, expect = ref.decode(data)
    assert rerr == "" and np.array_equal(rgba, expect), "Modular output must be bit-exact"


@pytest.mark.gpu
def test_coop_squeeze_round_trip(gpu, ref):
    for w, h, opts in [(600, 300, dict(tree=1)), (2600, 2100, dict(tree=1))]:   # (tree=5 quantises: not a round trip)
        plain = synth("modular", w, h, 7, **opts)
        squeezed = synth("modular", w, h, 7, squeeze=1, **opts)
        fr = gpu.Frame(squeezed)
        coop, total = fr.coop_sections()
        fr.close()
        assert coop == total
        rerr, expect = ref.decode(plain)
        err, rgba = gpu.decode(squeezed)
        assert rerr == "" and err == "" and np.array_equal(rgba, expect)


@pytest.mark.gpu
def test_coop_reports_damage_like_the_reference(gpu, ref):
    data = bytearray(synth("modular", 600, 300, 71, tree=5))
    rng = np.random.default_rng(5)
    rejected = 0
    for _ in range(24):
        m = bytearray(data)
        m[int(rng.integers(len(m) // 3, len(m) - 4))] ^= 1 << int(rng.integers(0, 8))
        rerr, rexp = ref.decode(bytes(m))
        err, rgba = gpu.decode(bytes(m))
        assert err == rerr, "reference: %r, here: %r" % (rerr, err)
        if rerr == "":
            assert np.array_equal(rgba, rexp)
        else:
            rejected += 1
    assert rejected >= 1
    for cut in (100, 1000, len(data) // 2):
        assert gpu.decode(bytes(data[: len(data) - cut]))[0] == ref.decode(bytes(data[: len(data) - cut]))[0]


CHILD = r"""
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, j40_amd
from streams import synth
out = []
for w, h, opts in %r:
    data = bytearray(synth("modular", w, h, 71, **opts))
    for damage in (0, 1, 2):
        m = bytearray(data)
        if damage:
            m[len(m) // 2 + 17 * damage] ^= 0x20
        err, rgba = j40_amd.decode(bytes(m))
        out.append((err, hashlib.sha256(rgba.tobytes()).hexdigest() if rgba is not None and err == "" else ""))
print(repr(out))
"""


@pytest.mark.gpu
def test_coop_and_general_kernel_agree(gpu):
    cases = [c for c in COOP_CASES if c[0] * c[1] > 2000] + [(600, 300, dict(tree=5, squeeze=1)), (2600, 2100, dict(tree=5, squeeze=3)), (2100, 1300, dict(tree=1, alpha=1))]
    script = CHILD % (ROOT, os.path.join(ROOT, "tests"), cases)
    runs = []
    # the cooperative kernel (default for these sizes), the general kernel only, and four sections per wavefront (k_modular_quad,
    # normally for frames with thousands of sections) forced onto every section that can take it
    for extra in ({}, {"J40HIP_NO_COOP": "1"}, {"J40HIP_QUAD_MIN": "1"}, {"J40HIP_NO_SPLIT": "1"}, {"J40HIP_SPLIT_NO_FAST": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(eval(r.stdout.strip().splitlines()[-1]))
    assert runs[0] == runs[1]
    assert runs[0] == runs[2]
    assert any(e == "" for e, _ in runs[0]) and any(e != "" for e, _ in runs[0])


def test_which_sections_the_two_pass_decoder_takes(built):
    """modular_split.hip takes the sections whose MA tree looks only at a sample's position (properties 0-3) and predicts without the
    weighted predictor -- none of those with neighbour properties, the weighted predictor or previous-channel properties"""
    import j40_amd
    from streams import synth
    want = {(): True, (("tree", 1),): False, (("tree", 2),): False, (("tree", 3), ("alpha", 1)): False, (("prefix", 1), ("lz77", 1), ("alpha", 1)): True, (("palette", 2), ("alpha", 1)): True}
    for opts, takes in want.items():
        f = j40_amd.Frame(synth("modular", 300, 200, 7, **dict(opts)))
        coop, total = f.coop_sections()
        split = f.split_sections()
        assert (split > 0) == takes, (opts, split, coop, total)
        assert split + coop <= total
        f.close()
