"""j40_amd.sharding -- one frame decoded by several GPUs (BASELINE.json config 3).

The path shards by pass-group section: sections are independent once the LF data is known. Rank r takes a CONTIGUOUS RANGE of
groups (raster order) whose sections add up to about 1 / world of the frame's section BYTES (the TOC gives every section's size,
j40.h:5529) -- entropy decode time follows the bytes, and a band of whole group rows can be 40 % off the mean (17 rows over 8 ranks).
What moves between ranks is small and sits at the two ends of the path:

  * in:  the codestream (a few MB), broadcast from rank 0; every rank parses headers, TOC and LF sections itself (the parsed LF
         bundle is ~3x larger than the codestream it is derived from). decode_sharded(..., lf_bundle=True) is the other form:
         rank 0 alone parses and broadcasts the parsed frame as one blob (Frame.lf_bundle / Frame.from_lf_bundle);
  * out: the pixels of each rank's groups (4 B/pixel). A contiguous range is at most three rectangles (the tail of its first group
         row, whole group rows, the head of its last group row); every rank sends its rectangles straight to rank 0 with
         point-to-point sends -- on xGMI every peer has its own link to rank 0, so the transfers run link-parallel -- and rank 0
         copies them into place.

Errors: before anything is gathered the ranks agree (all_reduce MAX) on whether every rank's decode succeeded, so that a failing
rank cannot leave the others waiting in a receive; a section whose event region overflowed ("evof") is decoded again with dense
planes first (j40hip_frame_force_dense), which is what the single-GPU public API does.

There is no collective inside the hot path. `torch.distributed` is the transport: backend "nccl" (= RCCL over xGMI) with device
tensors on the GPUs, "gloo" with host tensors in the CPU tests (tests/test_sharding.py).
"""
import numpy as np


def frame_geometry(width, height, group_size_shift):
    dim = 1 << group_size_shift
    return (width + dim - 1) // dim, (height + dim - 1) // dim, dim   # group columns, group rows, group size in pixels


def balanced_ranges(section_bytes, world):
    """contiguous group ranges [(first, count)] per rank, balanced by the bytes of the groups' sections.
    section_bytes: per group (summed over the passes). Greedy on the prefix sums: rank r ends at the group where the running
    total first reaches (r + 1) / world of the whole; ranks may end up empty when there are more ranks than groups."""
    sizes = np.maximum(np.asarray(section_bytes, dtype=np.float64), 1.0)   # (empty sections still cost a lane)
    n = len(sizes)
    prefix = np.concatenate([[0.0], np.cumsum(sizes)])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        g = int(np.searchsorted(prefix, target, side="left"))
        if g > 0 and abs(prefix[g - 1] - target) <= abs(prefix[min(g, n)] - target):
            g -= 1
        cuts.append(min(max(g, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def range_rectangles(first, count, width, height, group_size_shift):
    """the pixel rectangles (x0, y0, x1, y1) a contiguous range of groups covers: at most three"""
    gcols, grows, dim = frame_geometry(width, height, group_size_shift)
    rects = []
    g, end = first, first + count
    while g < end:
        row, col = divmod(g, gcols)
        if col == 0 and end - g >= gcols:                      # whole group rows
            rows = (end - g) // gcols
            rects.append((0, row * dim, width, min(height, (row + rows) * dim)))
            g += rows * gcols
        else:                                                  # part of one group row
            n = min(gcols - col, end - g)
            rects.append((col * dim, row * dim, min(width, (col + n) * dim), min(height, (row + 1) * dim)))
            g += n
    return rects


def broadcast_bytes(data, dist, device="cpu", src=0, tag=None):
    """(tag: an integer that travels with the length -- the source's `tag` is returned beside the bytes on every rank as (bytes, tag))
    the codestream from rank `src` to every rank; `data` is ignored on the other ranks. The source keeps the bytes it has (nothing
    comes back from the device for it); a receiving rank needs them in HOST memory -- its parser reads headers, TOC and the LF sections
    there -- so what arrived in a device tensor (RCCL moves device memory) is copied down once, through a pinned buffer"""
    import torch
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == src else 0, int(tag or 0) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    size, got_tag = int(n[0].item()), int(n[1].item())
    done = (lambda b: (b, got_tag)) if tag is not None else (lambda b: b)
    if size == 0:
        return done(b"")
    on_device = str(device) != "cpu"
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        if on_device:
            buf = buf.to(device, non_blocking=True)
        dist.broadcast(buf, src)
        return done(data if isinstance(data, bytes) else bytes(data))
    buf = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    if on_device:
        host = torch.empty(size, dtype=torch.uint8, pin_memory=True)
        host.copy_(buf, non_blocking=False)
        buf = host
    return done(buf.numpy().tobytes())


def agree_on_errors(code, dist, device="cpu"):
    """every rank passes its own 4-char code ('' = fine); returns the first failing rank's code on every rank"""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.zeros(world, dtype=torch.int64, device=device)
    mine[rank] = int.from_bytes(code.encode("latin1"), "big") if code else 0
    dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    for v in mine.cpu().tolist():
        if v:
            return int(v).to_bytes(4, "big").decode("latin1")
    return ""


def gather_rectangles(full, ranges, width, height, group_size_shift, dist, dst=0):
    """full: [height, width, 4] uint8 on the transport's device, holding this rank's groups. Every rank sends the rectangles of its
    range to rank `dst` (point-to-point, all posted at once); returns the assembled frame there, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ops, landing = [], []
    for r in range(world):
        if r == dst:
            continue
        for (x0, y0, x1, y1) in range_rectangles(ranges[r][0], ranges[r][1], width, height, group_size_shift):
            if rank == r:
                ops.append(dist.P2POp(dist.isend, full[y0:y1, x0:x1].contiguous(), dst))
            elif rank == dst:
                buf = torch.empty((y1 - y0, x1 - x0, 4), dtype=torch.uint8, device=full.device)
                ops.append(dist.P2POp(dist.irecv, buf, r))
                landing.append((buf, x0, y0, x1, y1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    for buf, x0, y0, x1, y1 in landing:
        full[y0:y1, x0:x1] = buf
    return full


LAST_FORM = {"lf_bundle": None}   # which form the last decode_sharded of this process took (bench.py puts it on the `sharded` record)


def decode_sharded(data, dist, decode_range, device="cpu", lf_bundle=None):
    """data: codestream on rank 0. decode_range(data, rank, world) -> (error code, full-frame uint8 tensor [height, width, 4] on
    `device` with this rank's groups decoded, ranges, (width, height, group_size_shift)). Returns the frame on rank 0; raises
    the same J40Error on every rank when any rank failed.
    lf_bundle: rank 0 alone parses the stream and broadcasts the parsed frame -- codestream, LF bundle and tables as one blob
    (Frame.lf_bundle, SURVEY.md 8e's wording) -- and decode_range is called as decode_range(blob, rank, world, from_bundle=True).
    The blob is ~3x the codestream; what it saves is the other ranks' host parse (they would run concurrently anyway).
    lf_bundle=None (default): the bundle from four ranks up -- one parse instead of N, and N - 1 ranks' host threads left alone, which is
    what counts when eight ranks share one container's CPU quota; below that every rank parses the 4-5 MB codestream itself, concurrently,
    which is as fast and moves a third of the bytes."""
    import j40_amd
    if lf_bundle is None:
        lf_bundle = dist.get_world_size() >= 4
    # one message from rank 0, tagged: 1 = the parsed frame's blob, 0 = the codestream (asked for, or because the frame has no bundle
    # form -- Modular frames -- or rank 0's parse failed: every rank then parses the codestream and reports what it finds)
    body, is_bundle = data, 0
    if dist.get_rank() == 0 and lf_bundle:
        try:
            fr = j40_amd.Frame(data)
            body, is_bundle = fr.lf_bundle(), 1
            fr.close()
        except j40_amd.J40Error:
            body, is_bundle = data, 0
    body, is_bundle = broadcast_bytes(body, dist, device, tag=is_bundle)
    from_bundle = bool(is_bundle)
    LAST_FORM["lf_bundle"] = from_bundle
    if from_bundle:
        err, full, ranges, (width, height, shift) = decode_range(body, dist.get_rank(), dist.get_world_size(), from_bundle=True)
    else:
        err, full, ranges, (width, height, shift) = decode_range(body, dist.get_rank(), dist.get_world_size())
    err = agree_on_errors(err, dist, device)
    if err:
        raise j40_amd.J40Error(err, "in a sharded decode")
    if str(full.device) != str(device):   # (CPU transport under a GPU decode: the gloo tests)
        full = full.to(device)
    return gather_rectangles(full, ranges, width, height, shift, dist)


def plan_ranges(frame, world):
    """the byte-balanced group ranges of a parsed frame"""
    sizes = frame.section_sizes()
    ng = frame.info["num_groups"]
    per_group = sizes.reshape(-1, ng).sum(axis=0) if len(sizes) >= ng and len(sizes) % ng == 0 else np.ones(ng)
    return balanced_ranges(per_group, world)


def hip_range_decoder(local_device):
    """decode_range for decode_sharded on a GPU: this rank's sections through libj40hip.so (j40hip_frame_set_group_range)"""
    import torch
    import j40_amd

    def decode_range(data, rank, world, from_bundle=False):
        try:
            fr = j40_amd.Frame.from_lf_bundle(data) if from_bundle else j40_amd.Frame(data)
        except j40_amd.J40Error as e:
            return e.code, None, None, (0, 0, 8)
        w, h, shift = fr.width, fr.height, fr.info["group_size_shift"]
        ranges = plan_ranges(fr, world)
        first, count = ranges[rank]
        full = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:%d" % local_device)
        err = ""
        if count:
            for attempt in range(2):
                try:
                    fr.upload(local_device)
                    fr.set_group_range(first, count)
                    fr.decode(full.data_ptr(), w * 4, torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    err = fr.status()
                except j40_amd.J40Error as e:
                    err = e.code
                if err != "evof":
                    break
                fr.force_dense(True)   # a section with more non-zero coefficients than its event region holds: dense planes
        fr.close()
        return err, full, ranges, (w, h, shift)

    return decode_range
