"""synthetic stream helpers shared by the tests, smoke() and bench.py (TEST INFRASTRUCTURE)"""
import os
import subprocess
import hashlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYNTH = os.path.join(ROOT, "build", "jxlsynth")
CACHE = os.path.join(ROOT, "build", "streams")


def synth(mode, w, h, seed, **opts):
    """returns the bytes of a generated stream (cached on disk by its parameters)"""
    os.makedirs(CACHE, exist_ok=True)
    key = "%s_%d_%d_%d_%s" % (mode, w, h, seed, "_".join("%s-%s" % kv for kv in sorted(opts.items())))
    path = os.path.join(CACHE, key + ".jxl")
    if not os.path.exists(path):
        if not os.path.exists(SYNTH):
            raise RuntimeError("build/jxlsynth is missing; run __graft_entry__.build()")
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.run([SYNTH, mode, str(w), str(h), str(seed), tmp] + ["%s=%s" % kv for kv in sorted(opts.items())],
                       check=True, stderr=subprocess.DEVNULL)
        os.replace(tmp, path)
    with open(path, "rb") as fp:
        return fp.read()


def sha(a):
    return hashlib.sha256(a.tobytes() if hasattr(a, "tobytes") else a).hexdigest()


# the VarDCT feature matrix every parity test walks (small frames: the reference decodes them in ms)
VARDCT_CASES = [
    ("default", dict()),
    ("bctx", dict(bctx=1)),
    ("presets", dict(presets=2)),
    ("orders", dict(orders=1)),
    ("passes", dict(passes=2)),
    ("fullheader", dict(fullheader=1, xqm=2, bqm=4, nosmooth=1)),
    ("simpleclusters", dict(simpleclusters=1, logalpha=6)),
    ("cfl", dict(cfl=1)),
    ("container_jxlc", dict(container=1)),
    ("container_jxlp", dict(container=2)),
    ("hf_prefix_codes", dict(hfprefix=1)),                  # HF coefficient streams with prefix codes (fast encoders)
    ("hf_lz77", dict(hflz77=1)),                            # ... with LZ77 copies
    ("hf_prefix_lz77_passes", dict(hfprefix=1, hflz77=1, passes=2)),
    ("icc_profile", dict(icc=700)),
    ("permuted_toc_two_passes", dict(permute=1, passes=2)),  # sections stored in a shuffled order, Lehmer-coded permutation in the TOC
    ("alpha_extra_channel", dict(alpha=1)),                 # Modular sub-image after the HF coefficients of every group; the reference outputs opaque pixels                         # want_icc: the ICC stream is decoded and discarded like in the reference
    ("bit_depth_12", dict(bpp=12, cfl=1)),                  # more than 8 bits: the long way through the transfer curve, scaling to 8 bits at the end
    ("bit_depth_15", dict(bpp=15)),
    ("custom_dequant_matrices", dict(dq=2)),                # HfGlobal codes the 8x8 matrices in the Hornuss / DCT2x2 / DCT4x4 / DCT4x8 / AFV / band forms and two larger ones raw (Modular sub-images)
    ("not_xyb_encoded_decoded_as_xyb", dict(alpha=1, fullheader=1, noxyb=1)),   # reference quirk: the XYB inverse runs on any VarDCT frame (j40.h:7206)
]

# the Modular feature matrix (width, height, options); all decode bit-exactly
MODULAR_CASES = [
    ("single_group_gradient", 256, 256, dict()),
    ("fjxl_like_rgba", 256, 256, dict(alpha=1, prefix=1, lz77=1)),          # config 1 shape: single section, alpha, RCT 6, prefix codes + LZ77
    ("multi_group", 600, 300, dict()),
    ("property_tree", 600, 300, dict(tree=1)),
    ("weighted_predictor", 600, 300, dict(tree=2)),
    ("previous_channel_props_alpha", 600, 300, dict(tree=3, alpha=1)),
    ("palette", 600, 300, dict(palette=1)),
    ("palette_deltas_synthetic_alpha", 600, 300, dict(palette=2, alpha=1)),
    ("palette_delta_prediction", 300, 200, dict(palette=3)),
    ("rct_permuted_prefix", 600, 300, dict(rct=13, prefix=1)),
    ("no_rct_group128_lz77", 200, 100, dict(rct=-1, groupshift=7, lz77=1)),
    ("palette_prediction_wp_tree", 160, 120, dict(palette=3, tree=2)),
    ("container", 300, 200, dict(container=1, tree=1)),
    ("icc_profile", 256, 256, dict(icc=500, alpha=1)),
    ("local_tree_copy", 600, 300, dict(localtree=1, tree=1)),               # every other group: use_global_tree = 0, own code spec
    ("local_tree_wp_prefix_lz77_alpha", 600, 300, dict(localtree=2, prefix=1, lz77=1, alpha=1)),  # local tree needs WP, global does not
    ("local_tree_under_wp_global", 520, 520, dict(localtree=2, tree=2, groupshift=7)),
    ("flagged_xyb_rendered_raw", 300, 200, dict(xyb=1)),                     # the reference applies no colour transform to Modular frames
    ("flagged_ycbcr_rendered_raw_alpha", 300, 200, dict(ycbcr=1, alpha=1)),
    ("two_passes_last_one_stays", 600, 300, dict(passes=2, tree=1)),         # every pass codes the groups again (j40.h:7025-7033)
    ("three_passes_local_rct_local_tree_alpha", 520, 300, dict(passes=3, localrct=4, localtree=2, alpha=1)),
    ("permuted_toc", 600, 300, dict(permute=1, tree=1, alpha=1)),
    ("local_rct_per_group", 600, 300, dict(localrct=4, alpha=1)),            # every group lists RCTs of its own (one or two)
    ("local_rct_local_tree_no_global_rct", 520, 520, dict(localrct=13, localtree=2, rct=-1, groupshift=7)),
    ("bit_depth_10", 600, 300, dict(bpp=10, tree=1)),
    ("bit_depth_14_wp_rct", 300, 200, dict(bpp=14, tree=2, rct=13)),
    ("local_palette", 600, 300, dict(localpalette=1, tree=1)),                 # every other group: a palette of its own -> decoded in a sub-image, pasted
    ("local_palette_deltas_alpha_prefix_lz77", 520, 300, dict(localpalette=2, alpha=1, prefix=1, lz77=1, groupshift=7)),
    ("local_palette_predicted_two_passes", 600, 300, dict(localpalette=3, passes=2, tree=2, rct=-1)),
    ("local_palette_beside_local_rct_local_tree", 300, 200, dict(localpalette=1, localrct=5, localtree=2, groupshift=7)),
    ("four_extra_channels_alpha_last", 600, 300, dict(extra=3, alpha=1, tree=3, localrct=7)),   # depth channels ahead of the alpha channel
]
