"""config 1's frame (256 x 256 RGBA, one section, prefix codes + LZ77) and a few more Modular frames with position-only trees, decoded by the
two-pass kernels (modular_split.hip) and -- J40HIP_NO_SPLIT=1 in a process of its own -- by the one-pass kernels: device ms, equal pixels,
the reference's pixels (MEASUREMENT TOOL)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = [("config1_256x256_rgba_prefix_lz77", 256, 256, dict(alpha=1, prefix=1, lz77=1), 101), ("256x256_rgba_ans", 256, 256, dict(alpha=1), 101),
         ("1024x1024_rgb_prefix_lz77_one_group", 1024, 1024, dict(prefix=1, lz77=1, groupshift=10), 5), ("2048x2048_rgb_ans_64_groups", 2048, 2048, dict(), 5),
         ("4096x4096_rgb_prefix_lz77_256_groups", 4096, 4096, dict(prefix=1, lz77=1), 5)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, j40_amd, hashlib
    from streams import synth
    from refdec import Ref
    ref = Ref()
    out = {}
    for name, w, h, opts, seed in CASES:
        data = synth("modular", w, h, seed, **opts)
        fr = j40_amd.Frame(data); fr.upload(0)
        o = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
        ms = min((fr.decode_timed(o.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream) for _ in range(3)), key=lambda m: float(sum(m)))
        st = fr.status()
        rerr, expect = ref.decode(data) if w * h <= 2048 * 2048 else ("", None)
        px = o.cpu().numpy()
        out[name] = {"ms": round(float(sum(ms)), 3), "status": st, "split_sections": fr.split_sections(), "equals_reference": None if expect is None else bool(np.array_equal(px, expect)), "sha": hashlib.sha256(px.tobytes()).hexdigest()[:16]}
        fr.close()
    print(json.dumps(out))
else:
    res = {}
    for label, env in (("two_pass", {}), ("one_pass", {"J40HIP_NO_SPLIT": "1"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        res[label] = json.loads(line[-1]) if line else {"error": p.stderr[-1500:]}
    print(json.dumps(res, indent=1))
