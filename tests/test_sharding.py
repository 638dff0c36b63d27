"""multi-GPU decode of one frame, exercised with world_size-2 and -3 gloo jobs on the CPU:
band arithmetic, codestream broadcast, per-rank partial decode (group ranges), padded gather and reassembly."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from streams import synth, ROOT, CACHE as STREAMS


def test_row_bands_cover_every_row_once():
    from j40_amd import sharding
    for rows in range(1, 40):
        for world in (1, 2, 3, 4, 8):
            bands = sharding.row_bands(rows, world)
            assert len(bands) == world and bands[0][0] == 0 and sum(n for _, n in bands) == rows
            assert all(a[0] + a[1] == b[0] for a, b in zip(bands, bands[1:]))
            assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    # the north star's case: 4320 rows of pixels = 17 group rows over 8 GPUs
    assert [n for _, n in sharding.row_bands(17, 8)] == [3, 2, 2, 2, 2, 2, 2, 2]
    first, count, y0, y1 = sharding.rank_share(7680, 4320, 8, 8, 7)
    assert (first, count, y0, y1) == (15 * 30, 2 * 30, 3840, 4320)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,w,h", [(2, 600, 520), (3, 520, 776)])
def test_sharded_decode_over_gloo_matches_reference(built, ref, world, w, h):
    data = synth("vardct", w, h, 57, bctx=1)
    path = os.path.join(STREAMS, "shard_%d_%d.jxl" % (w, h))
    open(path, "wb").write(data)
    out = os.path.join(STREAMS, "shard_%d_%d_w%d.npy" % (w, h, world))
    if os.path.exists(out):
        os.remove(out)
    port = free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharding_worker.py"), str(r), str(world), str(port), path, out], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    rerr, expect = ref.decode(data)
    assert rerr == "" and got.shape == expect.shape
    assert np.abs(got.astype(np.int32) - expect.astype(np.int32)).max() <= 1
