"""worker of tests/test_sharding.py: one rank of a world_size-N gloo job on the CPU. The per-band decode is done by the
CPU checker (tests/hostsim: the device functions compiled for the host) instead of the HIP kernels; everything around it
-- range arithmetic, codestream broadcast, error agreement, point-to-point gather, reassembly -- is the code the GPU path runs (j40_amd.sharding)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def hostsim_range_decoder():
    import torch
    import j40_amd
    from j40_amd import sharding
    S = C.CDLL(os.path.join(ROOT, "build", "libhostsim.so"))
    S.hostsim_decode.restype = C.c_uint32
    S.hostsim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    S.hostsim_set_group_range.argtypes = [C.c_int64, C.c_int64]

    def decode_range(data, rank, world):
        fr = j40_amd.Frame(data)   # host parse only (no device involved)
        w, h, shift = fr.width, fr.height, fr.info["group_size_shift"]
        ranges = sharding.plan_ranges(fr, world)
        fr.close()
        first, count = ranges[rank]
        full = np.full((h, w, 4), 7, np.uint8)   # pixels outside this rank's groups must stay untouched
        if count:
            buf = C.create_string_buffer(data, len(data))
            S.hostsim_set_group_range(first, count)
            err = S.hostsim_decode(buf, len(data), full.ctypes.data, None, 0)
            S.hostsim_set_group_range(0, -1)
            assert err == 0, hex(err)
            mine = np.zeros((h, w), bool)
            for x0, y0, x1, y1 in sharding.range_rectangles(first, count, w, h, shift):
                mine[y0:y1, x0:x1] = True
            assert np.all(full[~mine] == 7), "a range decode wrote outside its groups"
        return "", torch.from_numpy(full), ranges, (w, h, shift)

    return decode_range


def run(rank, world, port, stream_path, out_path):
    import torch
    import torch.distributed as dist
    from j40_amd import sharding
    rccl = len(sys.argv) > 6 and sys.argv[6] == "rccl"   # one device per rank, device tensors over RCCL (backend "nccl"), the LF-bundle choice left to decode_sharded
    if rccl:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    data = open(stream_path, "rb").read() if rank == 0 else b""
    if rccl:
        frame = sharding.decode_sharded(data, dist, sharding.hip_range_decoder(rank), torch.device("cuda", rank))
        if rank == 0:
            np.save(out_path, frame.cpu().numpy())
        dist.barrier()
        dist.destroy_process_group()
        return
    # sys.argv[6] == "hip": the ranks decode their ranges on the GPU (every rank on device 0 of a one-GPU box, else its own) through
    # libj40hip.so; the transport stays gloo with host tensors
    if len(sys.argv) > 6 and sys.argv[6] in ("hip", "hipbundle"):   # "hipbundle": rank 0 parses alone and broadcasts the LF bundle
        ndev = torch.cuda.device_count()
        frame = sharding.decode_sharded(data, dist, sharding.hip_range_decoder(rank % max(ndev, 1)), lf_bundle=sys.argv[6] == "hipbundle")
    else:
        frame = sharding.decode_sharded(data, dist, hostsim_range_decoder())
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
