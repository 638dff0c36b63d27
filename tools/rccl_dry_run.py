#!/usr/bin/env python3
"""Dry run of the RCCL calls of the sharded decode (j40_amd/sharding.py) on a box with ONE GPU: a process group of one rank with
backend "nccl" (= RCCL), the codestream broadcast, the error agreement (all_reduce MAX), a point-to-point transfer posted with
batch_isend_irecv (to the rank itself: the only peer there is) and a whole decode_sharded of an 8K frame. Not a measurement of
scaling: it makes sure the first 8-GPU run is not the first execution of these calls. Prints one JSON line.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_dry_run.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
import j40_amd
from j40_amd import sharding
from streams import synth

out = {}
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
out["backend"], out["world"] = dist.get_backend(), dist.get_world_size()
data = synth("vardct", 7680, 4320, 3, forward=1)
t0 = time.perf_counter(); got = sharding.broadcast_bytes(data, dist, dev); out["broadcast_bytes_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
assert got == data
assert sharding.agree_on_errors("", dist, dev) == "" and sharding.agree_on_errors("shrt", dist, dev) == "shrt"
out["all_reduce"] = "ok"
a = torch.arange(1 << 20, dtype=torch.uint8, device=dev).reshape(256, 1024, 4); b = torch.zeros_like(a)
try:
    t0 = time.perf_counter()
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
        req.wait()
    torch.cuda.synchronize()
    out["p2p_to_self"] = "ok" if torch.equal(a, b) else "wrong data"
    out["p2p_to_self_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
except Exception as e:   # (RCCL builds differ in whether a rank may send to itself)
    out["p2p_to_self"] = "refused: %s" % str(e).splitlines()[0][:200]
decode = sharding.hip_range_decoder(dev.index)
full = sharding.decode_sharded(data, dist, decode, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    full = sharding.decode_sharded(data, dist, decode, dev)
torch.cuda.synchronize()
out["decode_sharded_8k_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
err, px = j40_amd.decode(data)
out["pixels_equal_single_decode"] = bool(err == "" and np.array_equal(full.cpu().numpy(), px))
dist.destroy_process_group()
print(json.dumps(out))
