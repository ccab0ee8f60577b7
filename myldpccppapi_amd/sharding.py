"""Frame sharding across the GPUs of a node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

Frames are independent, so decoding needs no exchange at all.  The only
collective is the gather of the packed output bytes at the end -- (B/G)*K/8 bytes
per rank (16.6 MB at 4096 frames of the DVB-S2 rate-1/2 code), negligible next to
>= 100 ms of decoding.  The reference has nothing comparable (single device,
MyLdpc.cpp:235).
"""
import torch
import torch.distributed as dist


def shard_range(total_frames, rank, world):
    """Contiguous, balanced frame range [lo, hi) of `rank` (earlier ranks take the
    remainder), so concatenating the ranks' outputs in rank order restores frame order."""
    base, rem = divmod(int(total_frames), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_decoded(local_bytes, total_frames, K, group=None, dst=None):
    """Gather per-rank packed outputs (uint8 tensors, K/8 bytes per frame, frames in
    rank order as given by shard_range) into one tensor of total_frames*K/8 bytes.

    dst=None: every rank gets the result (all_gather); else only rank `dst`
    (gather) and the others get None.  Ragged shards are padded to the largest."""
    if K % 8:
        raise ValueError("sharded gather needs byte-aligned frames (K % 8 == 0)")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    kb = K // 8
    sizes = [(shard_range(total_frames, r, world)[1] - shard_range(total_frames, r, world)[0]) * kb
             for r in range(world)]
    if local_bytes.numel() != sizes[rank]:
        raise ValueError("rank %d holds %d bytes, expected %d" % (rank, local_bytes.numel(), sizes[rank]))
    mx = max(sizes)
    buf = local_bytes
    if buf.numel() != mx:
        buf = torch.zeros(mx, dtype=torch.uint8, device=local_bytes.device)
        buf[:sizes[rank]] = local_bytes
    if dst is None:
        out = torch.empty(world * mx, dtype=torch.uint8, device=local_bytes.device)
        dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
        parts = [out[r * mx:r * mx + sizes[r]] for r in range(world)]
        return torch.cat(parts)
    glist = [torch.empty(mx, dtype=torch.uint8, device=local_bytes.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(buf.contiguous(), glist, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([glist[r][:sizes[r]] for r in range(world)])


def decode_sharded(decode_fn, total_frames, K, group=None, dst=None):
    """decode_fn(lo, hi) -> uint8 tensor with this rank's packed bytes for frames
    [lo, hi).  Returns the gathered bytes (see gather_decoded)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(total_frames, rank, world)
    local = decode_fn(lo, hi)
    return gather_decoded(local, total_frames, K, group=group, dst=dst)
