"""world_size-2 gloo run of the multi-GPU plumbing (view sharding + keypoint-list gather)."""
import importlib
import json
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kd = importlib.import_module("keypoint-learning_amd.dist")
    views = kd.shard(5, world, rank)
    cap = 8
    # each rank "detects" a different, ragged keypoint list for its first view
    kp = torch.arange(100 * rank, 100 * rank + 3 + 9 * rank, dtype=torch.int32)   # 3 or 12 (> cap)
    packed = kd.pack_keypoints(kp, torch.tensor([kp.numel()], dtype=torch.int32), cap)
    g = kd.gather_keypoints(packed)
    # rank 1's list (12 entries) does not fit cap = 8: the packed count stays 12 and a strict unpack refuses the row
    try:
        kd.unpack_keypoints(g)
        refused = None
    except kd.KeypointListError as e:
        refused = str(e)
    lists = kd.unpack_keypoints(g, strict=False)
    with open(os.path.join(out_dir, "r%d.json" % rank), "w") as f:
        json.dump([views, [x.tolist() for x in lists], refused, g[:, 0].tolist()], f)
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "r0.json"))
    r1 = json.load(open(tmp_path / "r1.json"))
    assert r0[0] == [0, 2, 4] and r1[0] == [1, 3]
    want = [[0, 1, 2], list(range(100, 108))]          # second list cut at cap = 8 (strict=False)
    assert r0[1] == want and r1[1] == want
    assert r0[3] == [3, 12] and r1[3] == [3, 12]       # the TRUE counts travel
    assert r0[2] and "(1, 12)" in r0[2] and r1[2]      # and a strict unpack names the cut row


def test_pack_is_padded_and_counts():
    kd = importlib.import_module("keypoint-learning_amd.dist")
    p = kd.pack_keypoints(torch.tensor([5, 6, 7], dtype=torch.int32), 2, 4)
    assert p.tolist() == [2, 5, 6, 7, 0]               # count says 2, buffer had 3: extra is padding
    assert kd.shard(10, 4, 3) == [3, 7]


def test_unpack_refuses_cut_and_failed_lists():
    import pytest
    kd = importlib.import_module("keypoint-learning_amd.dist")
    ok = kd.pack_keypoints(torch.tensor([1, 2, 3], dtype=torch.int32), 3, 4)
    cut = kd.pack_keypoints(torch.arange(9, dtype=torch.int32), 9, 4)
    failed = kd.pack_keypoints(torch.zeros(4, dtype=torch.int32), torch.tensor([-1], dtype=torch.int32), 4)
    assert cut[0].item() == 9 and failed[0].item() == -1
    assert [x.tolist() for x in kd.unpack_keypoints(torch.stack([ok, ok]))] == [[1, 2, 3], [1, 2, 3]]
    with pytest.raises(kd.KeypointListError):
        kd.unpack_keypoints(torch.stack([ok, cut]))
    with pytest.raises(kd.KeypointListError):
        kd.unpack_keypoints(torch.stack([failed, ok]))
    assert [x.tolist() for x in kd.unpack_keypoints(torch.stack([cut, failed]), strict=False)] == [[0, 1, 2, 3], []]


def test_slab_plan_covers_every_finite_point_once_and_keeps_halos():
    import importlib
    import numpy as np
    kd = importlib.import_module("keypoint-learning_amd.dist")
    rng = np.random.default_rng(2)
    xyz = rng.uniform(0, [100, 30, 5], size=(5000, 3)).astype(np.float32)
    xyz[17] = np.nan
    halo = 7.5
    origin, plans = kd.slab_plan(xyz, 4, halo)
    assert np.array_equal(origin, np.nanmin(xyz, axis=0))
    owner = np.zeros(len(xyz), int)
    for p in plans:
        assert np.all(np.diff(p["idx"]) > 0)                    # ascending global index
        owner[p["idx"][p["interior"]]] += 1
        x = xyz[p["idx"], 0].astype(np.float64)
        xin = x[p["interior"]]
        # everything within the halo of the interior's extent along the split axis is in the slab
        near = np.nonzero((xyz[:, 0] >= xin.min() - halo + 1e-6) & (xyz[:, 0] < xin.max() + halo - 1e-6))[0]
        assert np.isin(near, p["idx"]).all()
    assert owner[17] == 0 and np.all(np.delete(owner, 17) == 1)
    # merge: interiors only
    scores = [np.arange(len(p["idx"]), dtype=np.float32) for p in plans]
    kps = [np.arange(0, len(p["idx"]), 3) for p in plans]
    sc, kp = kd.merge_slabs(len(xyz), plans, scores, kps)
    assert np.isnan(sc[17]) and np.isfinite(np.delete(sc, 17)).all() and np.all(np.diff(kp) > 0)
    for p, k in zip(plans, kps):
        mine = p["idx"][k[p["interior"][k]]]
        assert np.isin(mine, kp).all()


def test_pin_to_gpu_numa_is_best_effort():
    """no GPU here: the helper reports why it did nothing, leaves the affinity alone and never raises"""
    kd = importlib.import_module("keypoint-learning_amd.dist")
    before = os.sched_getaffinity(0)
    out = kd.pin_to_gpu_numa(0)
    assert isinstance(out, dict) and (("skipped" in out) != ("numa_node" in out))
    if "skipped" in out:
        assert os.sched_getaffinity(0) == before
    os.sched_setaffinity(0, before)


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_more_ranks_than_devices():
    """Without --share-devices a run with more ranks than visible GPUs is refused with a non-zero exit code and a message,
    not silently wrapped onto one device (which would report n_gpus = N for one GPU's work)."""
    import subprocess
    import sys
    import torch
    want = max(2, torch.cuda.device_count() + 1)
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(want), "--steps", "2", "--warmup", "1"], cwd=_ROOT,
                         env=env_clean, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "visible" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_scale_report_arithmetic():
    """tools/scale_report.py: efficiency / rank spread / collective share from the N = 1, 2, 4, 8 bench lines"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "scale_report.py"), "--self-test"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
