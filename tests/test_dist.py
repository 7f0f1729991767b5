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
    lists = kd.unpack_keypoints(g)
    with open(os.path.join(out_dir, "r%d.json" % rank), "w") as f:
        json.dump([views, [x.tolist() for x in lists]], f)
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "r0.json"))
    r1 = json.load(open(tmp_path / "r1.json"))
    assert r0[0] == [0, 2, 4] and r1[0] == [1, 3]
    want = [[0, 1, 2], list(range(100, 108))]          # second list clamped to cap = 8
    assert r0[1] == want and r1[1] == want


def test_pack_is_padded_and_counts():
    kd = importlib.import_module("keypoint-learning_amd.dist")
    p = kd.pack_keypoints(torch.tensor([5, 6, 7], dtype=torch.int32), 2, 4)
    assert p.tolist() == [2, 5, 6, 7, 0]               # count says 2, buffer had 3: extra is padding
    assert kd.shard(10, 4, 3) == [3, 7]
