"""bench.py as the driver launches it for N > 1: `python -m torch.distributed.run --nproc-per-node 2 bench.py
--gpus 2 ...`, here with both ranks on the one GPU of the test box (J40_BENCH_SHARE_DEVICE=1) and the collectives over gloo
(J40_BENCH_BACKEND=gloo) -- the multi-rank code path of the bench (process group, barriers, max-over-ranks clock, per-rank share of
the CPU quota, the frame-parallel aggregate, `device_resident` over the ranks, the `sharded` record: one frame split by group ranges
and gathered on rank 0), not a measurement. What an 8-GPU node adds is the RCCL transport (tools/rccl_dry_run.py runs it with one
rank) and eight devices."""
import json
import os
import socket
import subprocess
import sys

import pytest

from streams import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_through_bench_py_print_one_line_with_the_sharded_record(built):
    env = dict(os.environ, J40_BENCH_BACKEND="gloo", J40_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "2048", "--height", "1280", "--batch", "8", "--distinct", "2", "--pipe-batch", "8",
           "--device-output-batch", "8", "--device-output-steps", "2", "--resident-batch", "4", "--skip-modular"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]          # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["unit"] == "Mpixels/s" and r["value"] > 0 and r["vs_baseline"] is None and r["data"] == "synthetic"
    # the aggregate is both ranks' frames over the slowest rank's time
    frames = 8 * 2 * 2
    assert abs(r["value"] - 2048 * 1280 * frames / (r["ms_per_step"] * 2 / 1e3) / 1e6) <= 0.02 * r["value"]
    assert r["config"]["parallelism"] == "frames x2"
    assert r["pipeline"]["host_threads"] >= 1 and r["pipeline"]["cpu_quota_per_rank"] * 2 == pytest.approx(r["pipeline"]["cpu_quota"], abs=0.02)
    assert "pinned" in r["pipeline"]["numa"]
    assert "cpu_baseline" not in r                      # (rank 0 at N = 1 only)
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["stages"] and r["roofline"]["kernel"]
    assert r["device_output"]["value"] > 0 and r["device_resident"]["value"] > 0
    sh = r["sharded"]
    assert sh["backend"] == "gloo"
    v = sh["vardct_7680x4320"]
    assert v["ms_per_frame"] > 0 and v["pixels_equal_single_decode"] is True   # the two ranks' halves, gathered, are the frame one rank decodes
