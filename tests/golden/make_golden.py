"""Generates the committed golden fixtures: small streams written by tools/jxlsynth plus what the
*unmodified reference* (oracle/_ref, compiled from /root/reference) decodes them to.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
Outputs: tests/golden/<name>.jxl and tests/golden/manifest.json with, per stream, the sha256 of the
reference's RGBA output, of its quantised HF coefficients and of its LLF coefficients.
"""
import json
import os
import sys
import hashlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from refdec import Ref, RefStage  # noqa: E402
from streams import synth, VARDCT_CASES, MODULAR_CASES  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = Ref()
    manifest = {}
    cases = [("vardct_" + name, "vardct", 264, 200, 11 + i, opts) for i, (name, opts) in enumerate(VARDCT_CASES)]
    cases.append(("vardct_all_transforms", "vardct", 776, 520, 9, dict(maxlog=8, bctx=1, presets=2, orders=1)))
    for i, (name, w, h, opts) in enumerate(MODULAR_CASES):
        if w * h <= 100 * 1000 or name in ("multi_group", "local_tree_wp_prefix_lz77_alpha", "local_rct_per_group"):
            cases.append(("modular_" + name, "modular", min(w, 300), min(h, 200) if name != "fjxl_like_rgba" else h, 51 + i, opts))
    # round 2: encodes of a procedural picture (tools/jxlsynth forward=1), the 48-leaf tree that walks every property and
    # predictor, and Squeeze. The reference stops at Squeeze with "TODO" (j40.h:3812): those fixtures are pinned by what the
    # reference decodes the SAME picture to when it is coded without Squeeze (lossless round trip, tests/test_squeeze.py)
    cases.append(("vardct_forward_encode", "vardct", 264, 200, 61, dict(forward=1)))
    cases.append(("vardct_forward_encode_busy", "vardct", 520, 264, 62, dict(forward=1, detail=3, beta=0.15)))
    cases.append(("modular_wide_tree_48_leaves", "modular", 300, 200, 91, dict(tree=5)))
    cases.append(("modular_squeeze_default_list", "modular", 300, 200, 92, dict(squeeze=1, tree=1)))
    cases.append(("modular_squeeze_explicit_list_alpha", "modular", 300, 200, 93, dict(squeeze=3, alpha=1)))
    for name, mode, w, h, seed, opts in cases:
        data = synth(mode, w, h, seed, **opts)
        with open(os.path.join(HERE, name + ".jxl"), "wb") as fp:
            fp.write(data)
        if "squeeze" in opts:
            assert ref.decode(data)[0] == "TODO", name
            err, rgba = ref.decode(synth(mode, w, h, seed, **{k: v for k, v in opts.items() if k != "squeeze"}))
        else:
            err, rgba = ref.decode(data)
        assert err == "", (name, err)
        entry = dict(mode=mode, width=w, height=h, seed=seed, opts=opts, bytes=len(data), stream_sha256=hashlib.sha256(data).hexdigest(), rgba_sha256=sha(rgba))
        if "squeeze" in opts:
            entry["pinned_by"] = "the reference's decode of the same picture coded without Squeeze"
        if mode == "vardct":
            st = RefStage(ref, data)
            co, ll = [], []
            for g in range(st.info["num_lf_groups"]):
                for c in range(3):
                    co.append(st.coeffs(g, c))
                    ll.append(st.llf(g, c))
            entry["coeffs_sha256"] = sha(np.concatenate(co))
            entry["llf_sha256"] = sha(np.concatenate(ll))
            st.close()
        manifest[name] = entry
        print(name, len(data), "bytes")
    with open(os.path.join(HERE, "manifest.json"), "w") as fp:
        json.dump(manifest, fp, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
