"""config 4 (16384 x 16384 Modular, 4096 sections, tree = 1) through the section kernels' forms: the wave-cooperative default, the general
one-wavefront kernel, the lane-per-section form (J40HIP_K3_LANES=1) -- device ms each, equal pixels (MEASUREMENT TOOL)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, j40_amd, hashlib
    from streams import synth
    d = synth("modular", 16384, 16384, 21, tree=1, repeat=16)
    fr = j40_amd.Frame(d); fr.upload(0)
    o = torch.empty((16384, 16384, 4), dtype=torch.uint8, device="cuda:0")
    ms = [fr.decode_timed(o.data_ptr(), 16384 * 4, torch.cuda.current_stream().cuda_stream) for _ in range(2)]
    torch.cuda.synchronize()
    print(json.dumps({"ms": [[round(float(v), 2) for v in m] for m in ms], "status": fr.status(), "coop": fr.coop_sections(), "sha": hashlib.sha256(o[:2048].cpu().numpy().tobytes()).hexdigest()[:16]}))
else:
    for label, env in (("coop_default", {}), ("general_one_wavefront", {"J40HIP_NO_COOP": "1"}), ("lane_per_section", {"J40HIP_NO_COOP": "1", "J40HIP_K3_LANES": "1"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(label, line[-1] if line else p.stderr[-800:])
