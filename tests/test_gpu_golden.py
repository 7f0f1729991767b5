"""GPU path against the committed golden fixtures (no oracle in the loop for the expected values)
and, at BASELINE.json's full sizes, against the oracle + size-independent properties."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(__file__))
CFG_FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")


def detector(kpl, A, B, r, rn, thr, forest_path):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A)
    det.setNBins(B)
    det.setNonMaxima(True)
    det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(thr)
    det.setRadiusSearch(r)
    assert det.loadForest(forest_path), det.lastError()
    return det


def test_small_case_fixture(kpl, cases):
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    det = detector(kpl, 5, 6, float(z["r_feat"]), float(z["r_nms"]), 0.5, os.path.join(GOLD, "small_forest.yaml.gz"))
    det.setInputCloud(z["xyz"])
    det.setNormals(z["nrm"])
    for A, B in ((5, 6), (5, 10), (8, 10)):
        det.setNAnnulus(A)
        det.setNBins(B)
        assert cases.same_bits(det.computePointsForTrainingFeatures(z["query"]), z["feat_%dx%d" % (A, B)])
    det.setNAnnulus(5)
    det.setNBins(6)
    for thr in (0.0, 0.5, 0.85):
        det.setPredictionThreshold(float(np.float32(thr)))
        _, scores = det.compute()
        assert cases.same_bits(scores, z["scores"])
        assert np.array_equal(det.getKeypointsIndices(), z["kp_thr%03d_dr0" % int(thr * 100)])
        det.setNonMaximaDrawsRemove(True)
        det.setNonMaximaDrawsThreshold(float(z["draws_threshold"]))
        det.compute()
        assert np.array_equal(det.getKeypointsIndices(), z["kp_thr%03d_dr1" % int(thr * 100)])
        det.setNonMaximaDrawsRemove(False)


def test_config1_cheff_view(kpl, cases):
    """BASELINE.json configs[0]: a real view of the reference's data set."""
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    det = detector(kpl, 5, 6, float(z["r_feat"]), float(z["r_nms"]), float(z["thr"]), CFG_FOREST)
    det.setInputCloud(z["xyz"])
    det.setNormals(z["nrm"])
    kp, scores = det.compute()
    assert np.array_equal(det.getKeypointsIndices(), z["kp"])
    assert cases.same_bits(scores, z["scores"])          # tolerance allowed: 1e-5; achieved: 0
    assert np.array_equal(kp[:, :3], z["xyz"][z["kp"]])


def test_sorted_mode_fixture(kpl, cases):
    """tests/golden/sorted_case.npz: sorted-search mode on the committed clouds, no oracle in the loop"""
    z, c, s = (np.load(os.path.join(GOLD, f)) for f in ("small_case.npz", "cheff000.npz", "sorted_case.npz"))
    det = detector(kpl, 5, 6, float(z["r_feat"]), float(z["r_nms"]), float(np.float32(0.5)), os.path.join(GOLD, "small_forest.yaml.gz"))
    det.setSortedSearch(True)
    det.setInputCloud(z["xyz"])
    det.setNormals(z["nrm"])
    for A, B in ((5, 6), (8, 10)):
        det.setNAnnulus(A)
        det.setNBins(B)
        assert cases.same_bits(det.computePointsForTrainingFeatures(z["query"]), s["small_feat_%dx%d" % (A, B)])
    det.setNAnnulus(5)
    det.setNBins(6)
    _, scores = det.compute()
    assert cases.same_bits(scores, s["small_scores"]) and np.array_equal(det.getKeypointsIndices(), s["small_kp_thr050"])
    det = detector(kpl, 5, 6, float(c["r_feat"]), float(c["r_nms"]), float(c["thr"]), CFG_FOREST)
    det.setSortedSearch(True)
    det.setInputCloud(c["xyz"])
    det.setNormals(c["nrm"])
    _, scores = det.compute()
    assert cases.same_bits(scores, s["cheff_scores"]) and np.array_equal(det.getKeypointsIndices(), s["cheff_kp"])


def test_config2_full_size_vs_oracle_and_properties(kpl, oracle, cases):
    """BASELINE.json configs[1] at full size (200k points)."""
    from tools import forest_yaml, synth
    xyz, nrm = synth.make_cloud(500, 400, seed=1)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
    mr = oracle.cloud_resolution(xyz)
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
    det = detector(kpl, 5, 6, r, rn, thr, CFG_FOREST)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    kp = det.getKeypointsIndices().copy()
    fa = forest_yaml.load_forest(CFG_FOREST)
    o_scores, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, thr, cases.oracle_forest(fa), threads=cases.usable_cores())
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(kp, o_kp)
    # properties that need no oracle
    assert np.all(np.diff(kp) > 0)                                         # ascending, unique
    assert np.all(scores[kp].astype(np.float64) >= thr)                     # thresholded
    assert set(np.unique(np.round(scores * 10)).astype(int)) <= set(range(11))   # 1 - k/10
    # idempotence / determinism: same call twice, and a permuted copy gives the permuted answer
    _, scores2 = det.compute()
    assert cases.same_bits(scores2, scores) and np.array_equal(det.getKeypointsIndices(), kp)
    # NMS predicate re-checked by brute force on a sample of keypoints and non-keypoints
    from scipy.spatial import cKDTree
    tree = cKDTree(xyz.astype(np.float64))
    iskp = np.zeros(len(xyz), bool)
    iskp[kp] = True
    rng = np.random.RandomState(0)
    cand = np.flatnonzero(scores.astype(np.float64) >= thr)
    for i in rng.choice(cand, 400, replace=False):
        nb = [j for j in tree.query_ball_point(xyz[i].astype(np.float64), rn * 0.999)]
        assert iskp[i] == bool(np.all(scores[nb] <= scores[i])) or abs(len(nb) - len(
            tree.query_ball_point(xyz[i].astype(np.float64), rn * 1.001))) > 0


@pytest.mark.parametrize("rmul", [4.0, 10.0])
def test_config4_radius_sweep_dense(kpl, oracle, cases, rmul):
    """BASELINE.json configs[3] (neighbor-count stress) at reduced N: denser neighborhoods."""
    from tools import forest_yaml, synth
    xyz, nrm = synth.make_cloud(160, 120, seed=4)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1004)
    mr = oracle.cloud_resolution(xyz)
    r, rn = float(np.float32(rmul * mr)), float(np.float32(4 * mr))
    det = detector(kpl, 5, 6, r, rn, 0.6, CFG_FOREST)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    fa = forest_yaml.load_forest(CFG_FOREST)
    o_scores, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, 0.6, cases.oracle_forest(fa), threads=cases.usable_cores())
    assert cases.same_bits(scores, o_scores) and np.array_equal(det.getKeypointsIndices(), o_kp)


def test_config5_deep_forest_80_features(kpl, oracle, cases):
    """BASELINE.json configs[4] at reduced N: annuli=8 bins=10, 100 deep seeded trees, fused
    (locally denser) cloud."""
    from tools import synth
    xyz, nrm = synth.make_cloud(100, 80, seed=5, overlap_layers=2)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1005)
    mr = oracle.cloud_resolution(xyz)
    A, B = 8, 10
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    g = oracle.Grid(xyz, r)
    feat = g.features(nrm, A, B, r, np.arange(0, len(xyz), 7, dtype=np.int32))
    fa = synth.random_forest(A * B, ntrees=100, max_depth=25, seed=3, target_nodes_per_tree=600, feat=feat)
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(0.3); det.setRadiusSearch(r)
    cases.load_arrays(det, fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, 0.3, cases.oracle_forest(fa), threads=cases.usable_cores())
    assert cases.same_bits(scores, o_scores) and np.array_equal(det.getKeypointsIndices(), o_kp)
    assert len(np.unique(o_scores)) > 10


def test_device_resident_entry_points(kpl, oracle, cases):
    """kpl_bind_cloud_device / kpl_compute_device with torch-owned HBM buffers and a torch stream."""
    import torch
    from tools import forest_yaml
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn, thr = 6 * mr, 4 * mr, 0.85
    det = detector(kpl, 5, 6, r, rn, thr, CFG_FOREST)
    dev = torch.device("cuda", 0)
    n = len(xyz)
    dx, dn = torch.from_numpy(np.array(xyz)).to(dev), torch.from_numpy(np.array(nrm)).to(dev)
    ds = torch.empty(n, dtype=torch.float32, device=dev)
    dk = torch.empty(n, dtype=torch.int32, device=dev)
    dc = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        det.computeDevice(ds.data_ptr(), dk.data_ptr(), n, dc.data_ptr(), st.cuda_stream)
    st.synchronize()
    fa = forest_yaml.load_forest(CFG_FOREST)
    o_scores, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, thr, cases.oracle_forest(fa))
    assert cases.same_bits(ds.cpu().numpy(), o_scores)
    assert np.array_equal(dk[:int(dc.item())].cpu().numpy(), o_kp)
    stats = det.collectStats(st.cuda_stream)
    c = oracle.Grid(xyz, r).alg_counters(nrm, 5, 6, r, rn, thr, cases.oracle_forest(fa))
    for k in ("sum_kf", "sum_kn", "sum_depth", "n_scored", "n_thresholded"):
        assert stats[k] == c[k], k
    # capacity too small: count still reported
    dk2 = torch.empty(4, dtype=torch.int32, device=dev)
    det.detectDevice(None, dk2.data_ptr(), 4, dc.data_ptr(), None)
    torch.cuda.synchronize()
    assert int(dc.item()) == len(o_kp) and np.array_equal(dk2.cpu().numpy(), o_kp[:4])


def test_batched_compute_equals_single_views(kpl, oracle, cases):
    """kpl_compute_batch_device: one scoring launch for several independent views (different
    sizes, radii and thresholds) gives exactly the per-view results."""
    import torch
    from tools import forest_yaml, synth
    fa = forest_yaml.load_forest(CFG_FOREST)
    of = cases.oracle_forest(fa)
    dev = torch.device("cuda", 0)
    views, dets, bufs = [], [], []
    for k, (nx, ny, thr) in enumerate([(70, 60, 0.85), (90, 50, 0.5), (33, 31, 0.0), (64, 64, 0.85)]):
        xyz, nrm = synth.make_cloud(nx, ny, seed=20 + k, nan_points=3 * k, nan_normals=2 * k)
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1020 + k)
        mr = oracle.cloud_resolution(xyz)
        r, rn = float(np.float32((5 + k) * mr)), float(np.float32(4 * mr))
        det = detector(kpl, 5, 6, r, rn, float(np.float32(thr)), CFG_FOREST)
        n = len(xyz)
        dx, dn = torch.from_numpy(xyz.copy()).to(dev), torch.from_numpy(nrm.copy()).to(dev)
        ds = torch.empty(n, dtype=torch.float32, device=dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        views.append((xyz, nrm, r, rn, float(np.float32(thr))))
        dets.append(det)
        bufs.append((dx, dn, ds, dk))
    st = torch.cuda.Stream()
    for rep in range(3):                      # repeated batches reuse the self-cleaning scratch
        kpl.compute_batch_device(dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() for b in bufs],
                                 [len(b[2]) for b in bufs], [b[3][0:1].data_ptr() for b in bufs], st.cuda_stream)
        st.synchronize()
        for (xyz, nrm, r, rn, thr), det, (dx, dn, ds, dk) in zip(views, dets, bufs):
            assert det.syncStatus(st.cuda_stream) == kpl.OK
            o_sc, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, thr, of)
            assert cases.same_bits(ds.cpu().numpy(), o_sc)
            assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)
    # a handle twice in one batch is refused
    with pytest.raises(kpl.KplError):
        kpl.compute_batch_device([dets[0], dets[0]], None, [bufs[0][3][1:].data_ptr()] * 2, [4, 4],
                                 [bufs[0][3][0:1].data_ptr()] * 2, None)


def test_batched_mixed_modes_and_shapes(kpl, oracle, cases):
    """A full batch of 8 views that differ in everything a view descriptor carries: size (one view
    is empty, one has 3 points), A x B and forest, NMS on/off, draws_remove on/off."""
    import torch
    from tools import forest_yaml, synth
    fa30 = forest_yaml.load_forest(CFG_FOREST)
    dev = torch.device("cuda", 0)
    specs = [  # nx, ny, A, B, thr, non_maxima, draws_remove
        (60, 50, 5, 6, 0.85, True, False), (0, 0, 5, 6, 0.85, True, False), (40, 45, 4, 3, 0.2, True, True),
        (3, 1, 5, 6, 0.0, True, False), (50, 40, 5, 6, 0.5, False, False), (64, 33, 2, 2, 0.0, True, True),
        (45, 45, 4, 3, 0.6, True, False), (70, 20, 5, 6, 0.85, True, True)]
    forests = {(5, 6): fa30}
    views, dets, bufs = [], [], []
    for k, (nx, ny, A, B, thr, nms, draws) in enumerate(specs):
        if nx * ny > 0:
            xyz, nrm = synth.make_cloud(nx, ny, seed=40 + k, nan_points=k % 3, nan_normals=k % 2)
            xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1040 + k)
        else:
            xyz, nrm = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
        n = len(xyz)
        mr = oracle.cloud_resolution(xyz) if n > 8 else 1.0
        r, rn = float(np.float32(5 * mr)), float(np.float32(3.5 * mr))
        if (A, B) not in forests:
            forests[(A, B)] = cases.trained_forest(A=A, B=B, ntrees=7, max_depth=8)
        fa = forests[(A, B)]
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(nms); det.setNonMaxRadius(rn)
        det.setNonMaximaDrawsRemove(draws); det.setNonMaximaDrawsThreshold(float(np.float32(2 * mr)))
        det.setPredictionThreshold(float(np.float32(thr))); det.setRadiusSearch(r)
        cases.load_arrays(det, fa)
        dx = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev) if n else torch.zeros(1, 3, device=dev)
        dn = torch.from_numpy(np.ascontiguousarray(nrm)).to(dev) if n else torch.zeros(1, 3, device=dev)
        ds = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        views.append((xyz, nrm, A, B, r, rn, float(np.float32(thr)), nms, draws, float(np.float32(2 * mr)), fa))
        dets.append(det)
        bufs.append((dx, dn, ds, dk))
    for rep in range(2):
        kpl.compute_batch_device(dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() if len(b[3]) > 1 else None for b in bufs],
                                 [len(b[3]) - 1 for b in bufs], [b[3][0:1].data_ptr() for b in bufs], None)
        torch.cuda.synchronize()
        for (xyz, nrm, A, B, r, rn, thr, nms, draws, dthr, fa), det, (dx, dn, ds, dk) in zip(views, dets, bufs):
            assert det.syncStatus(None) == kpl.OK
            n = len(xyz)
            o_sc, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa), non_maxima=nms,
                                       draws_remove=draws, draws_threshold=dthr)
            assert cases.same_bits(ds.cpu().numpy()[:n], o_sc)
            assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)


def test_batch_call_can_be_captured_in_a_hip_graph(kpl, oracle, cases):
    """kpl_compute_batch_device only enqueues (no host sync, no allocation once the scratch is sized), so a
    caller can capture it with hipStreamBeginCapture -- here through torch.cuda.graph -- and replay it."""
    import torch
    from tools import forest_yaml
    fa = forest_yaml.load_forest(CFG_FOREST)
    of = cases.oracle_forest(fa)
    dev = torch.device("cuda", 0)
    dets, bufs, views = [], [], []
    for k in range(3):
        xyz, nrm = cases.cloud(60 + 10 * k, 50, seed=70 + k)
        mr = oracle.cloud_resolution(xyz)
        r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
        det = detector(kpl, 5, 6, r, rn, thr, CFG_FOREST)
        n = len(xyz)
        dx, dn = torch.from_numpy(xyz.copy()).to(dev), torch.from_numpy(nrm.copy()).to(dev)
        ds = torch.empty(n, dtype=torch.float32, device=dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det); bufs.append((dx, dn, ds, dk)); views.append((xyz, nrm, r, rn, thr))
    args = (dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() for b in bufs], [len(b[2]) for b in bufs],
            [b[3][0:1].data_ptr() for b in bufs])
    st = torch.cuda.Stream()
    for _ in range(2):                              # sizes the scratch (and grows the cell tables once)
        kpl.compute_batch_device(*args, st.cuda_stream)
        st.synchronize()
        for d in dets:
            d.syncStatus(st.cuda_stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        kpl.compute_batch_device(*args, torch.cuda.current_stream().cuda_stream)
    for rep in range(4):
        if rep == 2:        # the captured call on OTHER data in the same buffers (same sizes): nothing of a replay may
            for k in range(len(views)):                       # depend on what the previous one left behind
                xyz, nrm = cases.cloud(60 + 10 * k, 50, seed=170 + k)
                views[k] = (xyz, nrm) + views[k][2:]
                bufs[k][0].copy_(torch.from_numpy(xyz.copy()))
                bufs[k][1].copy_(torch.from_numpy(nrm.copy()))
        for b in bufs:
            b[2].fill_(-1.0); b[3].zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for (xyz, nrm, r, rn, thr), (dx, dn, ds, dk) in zip(views, bufs):
            o_sc, o_kp = oracle.detect(xyz, nrm, 5, 6, r, rn, thr, of)
            assert cases.same_bits(ds.cpu().numpy(), o_sc)
            assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)


def _child(cmd, timeout=600):
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[-1])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_device_in_fresh_processes():
    """The N > 1 path of bench.py exactly as the driver launches it (torch.distributed.run, one process per
    rank), 2 ranks sharing the one GPU of this box, gloo instead of RCCL: views sharded, every step ends
    with the gather of the packed keypoint lists, each rank finds its own lists in the gathered tensor
    (asserted inside bench.py), parity gate on, one JSON line from rank 0."""
    j = _child(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "3",
                "--warmup", "1", "--repeats", "2", "--batch", "2", "--nx", "80", "--ny", "60", "--lean",
                "--no-cpu-baseline", "--share-devices"])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["parity"]["scores_bit_exact"] and j["parity"]["keypoints_identical"]
    assert j["value"] > 0 and j["config"]["views_per_step_per_gpu"] == 2


def test_bench_gpus_2_launched_bare_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (how the driver starts N = 1, and what a user types): the
    script starts its two ranks itself -- torch.distributed.run as a child process, before anything touches the GPU -- and
    the line that comes back is the two-rank job's: n_gpus 2, a collective of world size 2, one timing row per rank."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                          "--repeats", "2", "--batch", "2", "--nx", "80", "--ny", "60", "--lean", "--no-cpu-baseline",
                          "--share-devices"], cwd=ROOT, env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["collective"]["world_size"] == 2 and len(j["per_rank_ms_per_step"]) == 2
    assert j["devices_shared"] is True          # (this box has one GPU; on an 8-GPU node the flag is not passed)
    assert j["parity"]["scores_bit_exact"] and j["parity"]["keypoints_identical"]
    assert j["roofline"]["counters"]["measured_in_this_run"] is False


def test_config3_recipe_two_ranks_gloo():
    """tools/run_cfg3.py (the one-command recipe of BASELINE.json configs[2]) at reduced size: 8 views over 2
    ranks on this one GPU, gloo; rank 0 checks its views against the oracle and its own lists in the gather."""
    j = _child(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "tools/run_cfg3.py", "--backend", "gloo", "--views", "8", "--nx", "80",
                "--ny", "60", "--rounds", "2"])
    assert j["n_gpus"] == 2 and j["views_per_rank"] == 4 and j["parity_first_4_views_rank0"] is True
