"""BASELINE.json configs[2], [3], [4] at their FULL sizes, each compared with the oracle (scores bit for
bit, keypoint index lists exactly), plus the one exchange step of the multi-GPU path executed over RCCL
on the one GPU of this box (a process group of one rank).

The oracle needs a few seconds per case on the host cores (500 k points at r = 10 mr: ~2 s on 8 cores;
1 M points, 100 trees: ~5 s), so these are ordinary members of the -m gpu suite."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")


def _detector(kpl, A, B, r, rn, thr):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A)
    det.setNBins(B)
    det.setNonMaxima(True)
    det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(thr)
    det.setRadiusSearch(r)
    return det


def _compute_resident(det, xyz, nrm):
    """compute() on a device-resident view (the bench's path); returns (scores, keypoint indices, stats)."""
    import torch
    import importlib
    kpl = importlib.import_module("keypoint-learning_amd")
    dev = torch.device("cuda", 0)
    n = len(xyz)
    dx, dn = torch.from_numpy(np.array(xyz)).to(dev), torch.from_numpy(np.array(nrm)).to(dev)
    ds = torch.empty(n, dtype=torch.float32, device=dev)
    dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
    st = torch.cuda.current_stream().cuda_stream
    det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
    while det.syncStatus(st) == kpl.ERR_RETRY:
        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
    torch.cuda.synchronize()
    cnt = int(dk[0].item())
    assert 0 <= cnt <= n
    stats = det.collectStats(st)
    return ds.cpu().numpy(), dk[1:1 + cnt].cpu().numpy(), stats


@pytest.fixture(scope="module")
def dense_cloud(oracle):
    """configs[3]: the 500 k-point dense cloud (707 x 707 jittered grid, tools/run_configs.py cfg4)."""
    from tools import synth
    xyz, nrm = synth.make_cloud(707, 707, seed=4)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1004)
    assert len(xyz) == 499849
    return xyz, nrm, oracle.cloud_resolution(xyz)


@pytest.mark.parametrize("rmul", [4.0, 6.0, 8.0, 10.0])
def test_config4_full_size_radius_sweep(kpl, oracle, cases, dense_cloud, rmul):
    """BASELINE.json configs[3]: r_feat sweep 4 / 6 / 8 / 10 mr on the 500 k-point cloud (K_f up to ~270: the
    accept-word lists of the feature kernel are drained and refilled several times per point)."""
    from tools import forest_yaml
    xyz, nrm, mr = dense_cloud
    r, rn, thr = float(np.float32(rmul * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
    det = _detector(kpl, 5, 6, r, rn, thr)
    assert det.loadForest(CFG_FOREST), det.lastError()
    scores, kp, stats = _compute_resident(det, xyz, nrm)
    fa = forest_yaml.load_forest(CFG_FOREST)
    o_scores, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, thr, cases.oracle_forest(fa), threads=cases.usable_cores())
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(kp, o_kp)
    assert len(kp) > 1000 and np.all(np.diff(kp) > 0)
    kf = stats["sum_kf"] / stats["n_scored"]
    assert {4.0: 25, 6.0: 60, 8.0: 110, 10.0: 180}[rmul] < kf < {4.0: 60, 6.0: 130, 8.0: 220, 10.0: 330}[rmul], kf


def test_config5_full_size_deep_forest(kpl, oracle, cases):
    """BASELINE.json configs[4]: 1 M-point fused cloud, annuli = 8, bins = 10, 100 trees of ~20 k nodes (2 M nodes
    = 16 MB: the blocked layout below the LDS-resident top, 4 lanes per point, trees out of step)."""
    from tools import synth
    xyz, nrm = synth.make_cloud(500, 500, seed=5, overlap_layers=4)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1005)
    assert len(xyz) == 1000000
    mr = oracle.cloud_resolution(xyz)
    A, B = 8, 10
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), 0.3
    g = oracle.Grid(xyz, r)
    feat = g.features(nrm, A, B, r, g.sorted_indices()[::97].astype(np.int32))
    fa = synth.random_forest(A * B, ntrees=100, max_depth=28, seed=3, target_nodes_per_tree=20000, feat=feat)
    assert fa.nnodes > 1900000
    det = _detector(kpl, A, B, r, rn, thr)
    cases.load_arrays(det, fa)
    scores, kp, stats = _compute_resident(det, xyz, nrm)
    of = cases.oracle_forest(fa)
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, of, threads=cases.usable_cores())
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(kp, o_kp)
    assert len(np.unique(o_scores[np.isfinite(o_scores)])) > 20        # 1 - k/100: many different vote counts
    # the walk really went below the top part: visited nodes per point as the oracle counts them on a sample
    sample = np.arange(0, len(xyz), 997, dtype=np.int32)
    fs = g.features(nrm, A, B, r, sample)
    depth = sum(of.predict_sum(row)[1] for row in fs)
    assert abs(stats["sum_depth"] / stats["n_scored"] - depth / len(sample)) < 0.05 * depth / len(sample)


def _child(cmd, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[-1])


def test_config3_full_size_64_views_on_one_gpu_with_rccl_gather():
    """BASELINE.json configs[2] as far as one GPU goes: all 64 views of 63 k points (batches of 8, two in
    flight), EVERY view checked against the oracle, and the job's one exchange step -- the all-gather of the
    packed keypoint lists -- executed over RCCL in a process group of one rank."""
    j = _child(["tools/run_cfg3.py", "--rounds", "3", "--parity-all", "--force-dist", "--backend", "nccl"])
    assert j["views"] == 64 and j["points_per_view"] == 63000 and j["views_per_rank"] == 64
    assert j["parity_first_4_views_rank0"] is True and j["views_checked_against_oracle"] == 64
    assert j["backend"] == "nccl" and j["Mpoints_per_s"] > 0


def test_bench_forced_through_the_rccl_collective():
    """bench.py --force-dist: with one rank every step still ends with the RCCL all-gather of full_step."""
    j = _child(["bench.py", "--force-dist", "--steps", "5", "--warmup", "2", "--repeats", "2", "--lean",
                "--no-cpu-baseline"])
    assert j["n_gpus"] == 1 and "RCCL" in j["config"]["exchange"]
    assert j["parity"]["scores_bit_exact"] and j["parity"]["keypoints_identical"]


def test_gather_keypoints_over_rccl_world_of_one(kpl, oracle, cases):
    """keypoint-learning_amd/dist.py gather_keypoints on DEVICE tensors: init_process_group("nccl") with one rank
    on this box's GPU, all_gather_into_tensor of the packed lists of two views, unpacked and compared with the
    oracle's keypoint lists.  In this process, so that librccl is loaded next to libkpl."""
    import importlib
    import socket
    import torch
    import torch.distributed as dist
    from tools import forest_yaml
    kd = importlib.import_module("keypoint-learning_amd.dist")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=dev)
        created = True
    try:
        assert dist.get_backend() == "nccl"
        fa = forest_yaml.load_forest(CFG_FOREST)
        of = cases.oracle_forest(fa)
        cap = 6300                     # = points per view: nothing is truncated on the way
        packed, expect = [], []
        for seed in (31, 32):
            xyz, nrm = cases.cloud(90, 70, seed=seed)
            mr = oracle.cloud_resolution(xyz)
            r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
            det = _detector(kpl, 5, 6, r, rn, thr)
            assert det.loadForest(CFG_FOREST)
            n = len(xyz)
            dx, dn = torch.from_numpy(xyz.copy()).to(dev), torch.from_numpy(nrm.copy()).to(dev)
            dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
            det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
            st = torch.cuda.current_stream().cuda_stream
            det.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
            while det.syncStatus(st) == kpl.ERR_RETRY:
                det.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
            packed.append(kd.pack_keypoints(dk[1:], dk[0:1], cap))
            expect.append(oracle.detect(xyz, nrm, 5, 6, r, rn, thr, of)[1])
        send = torch.stack(packed).view(-1)
        assert send.is_cuda
        out = kd.gather_keypoints(send)                   # all_gather_into_tensor over RCCL
        assert out.is_cuda and out.shape == (1, 2 * (cap + 1))
        lists = kd.unpack_keypoints(out.view(2, cap + 1))
        for got, want in zip(lists, expect):
            assert len(want) > 0 and np.array_equal(got.numpy(), want)
        tt = torch.tensor([1.5], dtype=torch.float64, device=dev)      # the bench's max-over-ranks reduction
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(tt.item()) == 1.5
    finally:
        if created:
            dist.destroy_process_group()
