"""Multi-GPU plumbing: views are independent, so they are sharded over ranks (one process per GPU)
and never exchange data on the data path; the only collective is the gather of the per-view
keypoint lists (RCCL all-gather on GPU tensors, gloo on CPU tensors in the tests).

The reference has no counterpart (single process); this implements SURVEY.md 8(e).
"""
import torch
import torch.distributed as dist


def shard(n_items, world, rank):
    """Round-robin assignment of `n_items` independent views to `world` ranks."""
    return list(range(rank, n_items, world))


def pack_keypoints(kp_idx, kp_count, cap):
    """[cap + 1] int32: element 0 = count (clamped to cap), then the indices, zero padded.
    `kp_idx` is the buffer kpl_detect_device wrote (length >= count), `kp_count` a 1-element
    tensor or an int.  Stays on the device of `kp_idx`; no host sync."""
    out = torch.zeros(cap + 1, dtype=torch.int32, device=kp_idx.device)
    if torch.is_tensor(kp_count):
        out[0:1] = torch.clamp(kp_count.to(torch.int32).reshape(1), max=cap)
    else:
        out[0] = min(int(kp_count), cap)
    n = min(cap, kp_idx.numel())
    out[1:1 + n] = kp_idx[:n]
    return out


def gather_keypoints(packed, group=None):
    """All-gathers one packed list per rank.  Returns a [world, cap + 1] tensor on every rank."""
    world = dist.get_world_size(group)
    if packed.is_cuda:
        out = torch.empty(world * packed.numel(), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(out, packed, group=group)
        return out.view(world, -1)
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed, group=group)
    return torch.stack(parts)


def unpack_keypoints(gathered):
    """[world, cap + 1] -> list of 1-D index tensors, one per rank (a count above cap means the
    list was truncated to cap entries; a negative count means that rank's call failed)."""
    g = gathered.cpu()
    cap = g.shape[1] - 1
    return [g[r, 1:1 + max(0, min(int(g[r, 0]), cap))].clone() for r in range(g.shape[0])]
