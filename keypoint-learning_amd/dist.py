"""Multi-GPU plumbing: views are independent, so they are sharded over ranks (one process per GPU)
and never exchange data on the data path; the only collective is the gather of the per-view
keypoint lists (RCCL all-gather on GPU tensors, gloo on CPU tensors in the tests).

The reference has no counterpart (single process); this implements SURVEY.md 8(e).
"""
import torch
import torch.distributed as dist


def pin_to_gpu_numa(local_rank):
    """Binds the calling process to the CPUs of the NUMA node its GPU hangs off -- BEFORE anything touches the GPU (no HIP
    call, no torch.cuda call: the rank processes of an 8-GPU node otherwise start on whatever cores the launcher left them
    and enqueue across the socket interconnect).  The GPU's PCI address comes from the KFD topology in sysfs (GPU nodes in
    enumeration order = HIP device order; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES given as a list of integers are
    honoured), its CPUs from the PCI device's local_cpulist.  Best effort: returns a dict for the bench line
    ({"numa_node": k, "cpus": n, "pci": "0000:..."}) or {"skipped": reason}; never raises."""
    import glob
    import os
    try:
        nodes = []
        for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda x: int(os.path.basename(x))):
            props = dict(line.split()[:2] for line in open(os.path.join(d, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0 and int(props.get("vendor_id", "0")) != 0:      # a GPU node
                nodes.append(props)
        order = list(range(len(nodes)))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            val = os.environ.get(var)
            if val:
                try:
                    order = [order[int(x)] for x in val.split(",") if x.strip() != ""]
                except (ValueError, IndexError):
                    return {"skipped": "%s=%s is not a list of device ordinals" % (var, val)}
        if not order:
            return {"skipped": "no GPU node in the KFD topology"}
        props = nodes[order[local_rank % len(order)]]
        loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        pci = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        base = "/sys/bus/pci/devices/" + pci
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)             # never outside what the launcher / cgroup allows
        if not cpus:
            return {"skipped": "no usable CPU local to %s" % pci}
        os.sched_setaffinity(0, cpus)
        numa = int(open(base + "/numa_node").read().strip())
        return {"numa_node": numa, "cpus": len(cpus), "pci": pci}
    except (OSError, KeyError, ValueError) as e:
        return {"skipped": "%s: %s" % (type(e).__name__, e)}


def shard(n_items, world, rank):
    """Round-robin assignment of `n_items` independent views to `world` ranks."""
    return list(range(rank, n_items, world))


def pack_keypoints(kp_idx, kp_count, cap):
    """[cap + 1] int32: element 0 = the TRUE count (not clamped: a count above cap tells the receiver that
    the list was cut, a negative one that the call failed), then the first min(count, cap) indices, zero
    padded.  `kp_idx` is the buffer kpl_detect_device wrote (length >= count), `kp_count` a 1-element
    tensor or an int.  Stays on the device of `kp_idx`; no host sync."""
    out = torch.zeros(cap + 1, dtype=torch.int32, device=kp_idx.device)
    if torch.is_tensor(kp_count):
        out[0:1] = kp_count.to(torch.int32).reshape(1)
    else:
        out[0] = int(kp_count)
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


class KeypointListError(ValueError):
    """A gathered keypoint list is unusable: cut at the packing capacity, or its call failed (count < 0)."""


def unpack_keypoints(gathered, strict=True):
    """[world, cap + 1] -> list of 1-D index tensors, one per row.  A count above cap means the list was cut to cap
    entries, a negative count that the row's call failed: both raise KeypointListError unless strict=False (then the
    cut list / an empty list is returned, as before)."""
    g = gathered.cpu()
    cap = g.shape[1] - 1
    counts = g[:, 0].tolist()
    if strict:
        bad = [(r, c) for r, c in enumerate(counts) if c < 0 or c > cap]
        if bad:
            raise KeypointListError("keypoint lists (row, count) %s do not fit the packing capacity %d or come from a failed call"
                                    % (bad[:8], cap))
    return [g[r, 1:1 + max(0, min(int(counts[r]), cap))].clone() for r in range(g.shape[0])]


# ---- one large cloud over several GPUs ("single huge cloud"): slabs with a halo ----

def slab_plan(xyz, parts, halo, axis=None):
    """Splits a cloud into `parts` slabs along `axis` (default: the longest extent) at point-count
    quantiles.  Slab k gets the points whose coordinate lies in [lo_k - halo, hi_k + halo); its
    INTERIOR is [lo_k, hi_k).  With halo >= r_feat + r_nms every interior point sees, inside the slab,
    all its NMS neighbors and all THEIR feature neighbors, so -- run with the whole cloud's grid
    origin (kpl_set_grid_origin) -- the slab reproduces the whole-cloud scores and keypoint decisions
    of its interior bit for bit.  Points with a non-finite coordinate belong to no slab (their score
    is NaN anyway).

    Returns (origin, plans): origin = float32[3] minimum of the finite points; plans[k] = dict with
    'idx' (ascending global indices of the slab's points) and 'interior' (bool mask over idx)."""
    import numpy as np
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    finite = np.isfinite(xyz).all(axis=1)
    good = np.nonzero(finite)[0]
    if len(good) == 0:
        return np.zeros(3, np.float32), [dict(idx=np.zeros(0, np.int64), interior=np.zeros(0, bool)) for _ in range(parts)]
    pts = xyz[good].astype(np.float64)
    origin = xyz[good].min(axis=0)
    if axis is None:
        axis = int(np.argmax(pts.max(axis=0) - pts.min(axis=0)))
    c = pts[:, axis]
    cuts = np.quantile(c, np.linspace(0.0, 1.0, parts + 1)[1:-1]) if parts > 1 else np.zeros(0)
    lo = np.concatenate([[-np.inf], cuts])
    hi = np.concatenate([cuts, [np.inf]])
    plans = []
    for k in range(parts):
        inside = (c >= lo[k] - halo) & (c < hi[k] + halo)
        idx = good[inside]                       # ascending global index: keeps the canonical order
        interior = (c[inside] >= lo[k]) & (c[inside] < hi[k])
        plans.append(dict(idx=idx, interior=interior))
    return origin, plans


def merge_slabs(n, plans, slab_scores, slab_keypoints):
    """Puts the interiors back together: scores[n] (NaN where no slab owns the point) and the sorted
    global keypoint indices.  slab_scores[k] / slab_keypoints[k] are slab k's outputs (keypoints as
    indices INTO the slab)."""
    import numpy as np
    scores = np.full(n, np.nan, dtype=np.float32)
    kps = []
    for plan, sc, kp in zip(plans, slab_scores, slab_keypoints):
        idx, interior = plan["idx"], plan["interior"]
        scores[idx[interior]] = np.asarray(sc, dtype=np.float32)[interior]
        kp = np.asarray(kp, dtype=np.int64)
        kps.append(idx[kp[interior[kp]]])
    kp_all = np.sort(np.concatenate(kps)) if kps else np.zeros(0, np.int64)
    return scores, kp_all.astype(np.int32)
