"""j40_amd.sharding -- one frame decoded by several GPUs (BASELINE.json config 2).

The path shards by pass-group section: sections are independent once the LF data is known, so rank r takes a band of
group rows. What has to move between ranks is small and sits at the two ends of the path:

  * in:  the codestream (a few MB; every rank parses headers, TOC and LF sections itself -- a broadcast of the parsed
         LF bundle would be ~2.5x larger than the codestream it is derived from, and the host parse is ~10 ms);
  * out: each rank's RGBA band (4 B/pixel), gathered on rank 0.

There is no collective inside the hot path. `torch.distributed` is the transport: backend "nccl" (= RCCL over xGMI)
with device tensors on the GPUs, "gloo" with host tensors in the CPU tests (tests/test_sharding.py), where the
per-band decode is done by the CPU checker instead of the HIP kernels.
"""
import numpy as np


def row_bands(num_group_rows, world):
    """splits the rows of pass groups into `world` contiguous bands, sizes differing by at most one row;
    returns [(first_row, rows)] per rank (rows may be 0 when there are more ranks than rows)"""
    base, extra = divmod(num_group_rows, world)
    bands, row = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        bands.append((row, n))
        row += n
    return bands


def frame_geometry(width, height, group_size_shift):
    dim = 1 << group_size_shift
    return (width + dim - 1) // dim, (height + dim - 1) // dim, dim   # group columns, group rows, group size in pixels


def rank_share(width, height, group_size_shift, world, rank):
    """what `rank` decodes: (first_group, num_groups, y0, y1) with [y0, y1) the pixel rows of its band"""
    gcols, grows, dim = frame_geometry(width, height, group_size_shift)
    row0, rows = row_bands(grows, world)[rank]
    return row0 * gcols, rows * gcols, min(height, row0 * dim), min(height, (row0 + rows) * dim)


def broadcast_bytes(data, dist, device="cpu", src=0):
    """the codestream from rank `src` to every rank; `data` is ignored on the other ranks"""
    import torch
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_bands(band, width, height, group_size_shift, dist, dst=0):
    """band: this rank's rows [y1 - y0, width, 4] uint8 (torch tensor on the transport's device). Returns the whole
    frame [height, width, 4] on rank `dst`, None elsewhere. Bands are padded to the tallest one so that a single
    gather moves them (row counts differ by at most one group row)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    shares = [rank_share(width, height, group_size_shift, world, r) for r in range(world)]
    tallest = max(y1 - y0 for _, _, y0, y1 in shares)
    padded = torch.zeros((tallest, width, 4), dtype=torch.uint8, device=band.device)
    padded[: band.shape[0]] = band
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((height, width, 4), dtype=torch.uint8, device=band.device)
    for (_, _, y0, y1), part in zip(shares, parts):
        out[y0:y1] = part[: y1 - y0]
    return out


def decode_sharded(data, dist, decode_band, device="cpu"):
    """data: codestream on rank 0. decode_band(data, first_group, num_groups, y0, y1) -> uint8 tensor [y1 - y0, width, 4]
    on `device` plus (width, height, group_size_shift). Returns the frame on rank 0."""
    data = broadcast_bytes(data, dist, device)
    band, (width, height, shift) = decode_band(data, dist.get_rank(), dist.get_world_size())
    return gather_bands(band, width, height, shift, dist)


def hip_band_decoder(local_device):
    """decode_band for decode_sharded on a GPU: the frame's sections of this rank through libj40hip.so"""
    import torch
    import j40_amd

    def decode_band(data, rank, world):
        fr = j40_amd.Frame(data)
        fr.upload(local_device)
        w, h, shift = fr.width, fr.height, fr.info["group_size_shift"]
        first, count, y0, y1 = rank_share(w, h, shift, world, rank)
        fr.set_group_range(first, count)
        full = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:%d" % local_device)
        if count:
            fr.decode(full.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            err = fr.status()
            if err:
                raise j40_amd.J40Error(err, "in a sharded decode")
        band = full[y0:y1].contiguous()
        fr.close()
        return band, (w, h, shift)

    return decode_band
