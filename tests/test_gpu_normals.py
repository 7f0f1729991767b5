"""GPU normal estimation (kpl_estimate_normals[_device], the step before the path: pcl::NormalEstimation,
/root/reference/src/main_test_detector.cpp:162-169 and include/impl/KeypointLearning.hpp:125-148) against the
oracle: same neighbors, same double arithmetic, so normals and curvature are compared bit for bit
(tolerance the north star would allow for this step: 1e-4 rad)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("k", [10, 3, 16, 25])
def test_k_search_bit_exact(kpl, oracle, cases, k):
    xyz, _ = cases.cloud(90, 70, seed=11, nan_points=5)
    det = kpl.KeypointLearningDetector()
    vp = (3.0, -2.0, 500.0)
    nrm, curv = det.estimateNormals(xyz, k=k, viewpoint=vp)
    o_nrm, o_curv = oracle.estimate_normals(xyz, k=k, viewpoint=vp)
    assert np.isfinite(o_nrm).all(1).sum() == len(xyz) - 5
    assert same(nrm, o_nrm) and same(curv, o_curv)


@pytest.mark.parametrize("rmul", [1.5, 3.0, 6.0])
def test_radius_search_bit_exact(kpl, oracle, cases, rmul):
    xyz, _ = cases.cloud(80, 60, seed=12, nan_points=3, layers=2)
    mr = oracle.cloud_resolution(xyz)
    r = float(np.float32(rmul * mr))
    det = kpl.KeypointLearningDetector()
    nrm, curv = det.estimateNormals(xyz, k=0, radius=r, viewpoint=(0, 0, 0))
    o_nrm, o_curv = oracle.estimate_normals(xyz, k=0, radius=r, viewpoint=(0, 0, 0))
    assert same(nrm, o_nrm) and same(curv, o_curv)


def test_lattice_ties_and_small_inputs(kpl, oracle):
    g = np.stack(np.meshgrid(np.arange(12), np.arange(9), [0.0]), -1).reshape(-1, 3).astype(np.float32)
    g[:, 2] = 0.25 * g[:, 1]
    det = kpl.KeypointLearningDetector()
    for k in (4, 9, 10):
        nrm, curv = det.estimateNormals(g, k=k, viewpoint=(0, 0, 50))
        o_nrm, o_curv = oracle.estimate_normals(g, k=k, viewpoint=(0, 0, 50))
        assert same(nrm, o_nrm) and same(curv, o_curv)
    for n in (0, 1, 2, 3, 7):
        nrm, curv = det.estimateNormals(g[:n], k=10)
        o_nrm, o_curv = oracle.estimate_normals(g[:n], k=10)
        assert nrm.shape == (n, 3) and same(nrm, o_nrm) and same(curv, o_curv)
    with pytest.raises(kpl.KplError):
        det.estimateNormals(g, k=33)
    with pytest.raises(kpl.KplError):
        det.estimateNormals(g, k=0, radius=0.0)


def test_full_size_view_then_detect_on_estimated_normals(kpl, oracle, cases):
    """200 k points: normals on the device, straight into pcl::Normal-shaped storage (32-byte records), then the
    hot path on them; everything equal to the oracle run on the oracle's normals."""
    import torch
    from tools import forest_yaml, synth
    import os
    root = os.path.dirname(os.path.dirname(__file__))
    forest = os.path.join(root, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
    xyz, _ = synth.make_cloud(500, 400, seed=1)
    n = len(xyz)
    dev = torch.device("cuda", 0)
    det = kpl.KeypointLearningDetector()
    mr = det.cloudResolution(xyz)
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
    det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
    assert det.loadForest(forest)
    dx = torch.from_numpy(xyz).to(dev)
    dnrm = torch.zeros(n, 8, dtype=torch.float32, device=dev)          # pcl::Normal: nx ny nz pad curvature pad pad pad
    det.bindCloudDevice(dx.data_ptr(), 12, dnrm.data_ptr(), 32, n)
    det.estimateNormalsDevice(10, 0.0, (0, 0, 1e4), dnrm.data_ptr(), 32, dnrm.data_ptr() + 16, 32, None)
    assert det.syncStatus(None) in (kpl.OK, kpl.ERR_RETRY)
    det.estimateNormalsDevice(10, 0.0, (0, 0, 1e4), dnrm.data_ptr(), 32, dnrm.data_ptr() + 16, 32, None)
    assert det.syncStatus(None) == kpl.OK
    o_nrm, o_curv = oracle.estimate_normals(xyz, k=10, viewpoint=(0, 0, 1e4))
    got = dnrm.cpu().numpy()
    assert same(got[:, :3], o_nrm) and same(got[:, 4], o_curv)
    ds = torch.empty(n, dtype=torch.float32, device=dev)
    dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    for _ in range(2):
        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), None)
        rc = det.syncStatus(None)
    assert rc == kpl.OK
    fa = forest_yaml.load_forest(forest)
    o_sc, o_kp = oracle.detect(xyz, o_nrm, 5, 6, r, rn, thr, cases.oracle_forest(fa), threads=cases.usable_cores())
    assert same(ds.cpu().numpy(), o_sc)
    assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)


def test_committed_fixture(kpl):
    """no oracle in the loop: the committed normals of the small case"""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    z, c = np.load(os.path.join(gold, "normals_case.npz")), np.load(os.path.join(gold, "small_case.npz"))
    det = kpl.KeypointLearningDetector()
    nk, ck = det.estimateNormals(c["xyz"], k=int(z["k"]), viewpoint=z["viewpoint"])
    nr, cr = det.estimateNormals(c["xyz"], k=0, radius=float(z["radius"]), viewpoint=z["viewpoint"])
    assert same(nk, z["nrm_k"]) and same(ck, z["curv_k"]) and same(nr, z["nrm_r"]) and same(cr, z["curv_r"])


def same_nan(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)])


@pytest.mark.parametrize("W,H,step,holes,smoothing,vp", [
    (64, 48, None, 0, 5.0, (0.0, 0.0, 0.0)),
    (160, 120, 77, 25, 5.0, (0.0, 0.0, 0.0)),
    (333, 77, 40, 60, 5.0, (1.0, -2.0, 9.0)),
    (40, 1100, 13, 80, 5.0, (0.0, 0.0, 0.0)),          # more rows than one band of the skewed recurrences
    (90, 70, 31, 10, 10.0, (0.0, 0.0, 0.0)),           # PCL's default smoothing size
    (50, 40, 20, 5, 3.5, (0.0, 0.0, 0.0)),
    (640, 480, 300, 400, 5.0, (0.0, 0.0, 0.0)),        # a Kinect frame
])
def test_organized_normals_bit_exact(kpl, oracle, W, H, step, holes, smoothing, vp):
    """kpl_estimate_normals_organized (pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT: the detector's
    fallback on an organized cloud, hpp:138-145) against the oracle's loop-by-loop restatement."""
    from tests.test_oracle_organized_normals import depth_image
    xyz = depth_image(W, H, seed=W + H, step=step, holes=holes)
    det = kpl.KeypointLearningDetector()
    nrm, curv = det.estimateNormalsOrganized(xyz, W, H, smoothing, vp)
    o_nrm, o_curv = oracle.integral_image_normals(xyz, W, H, smoothing, vp)
    assert same_nan(nrm, o_nrm)
    assert np.isnan(curv).all() and np.isnan(o_curv).all()
    assert np.isfinite(o_nrm[:, 0]).sum() > (W - 2 * int(smoothing)) * (H - 2 * int(smoothing)) // 2


def test_organized_normals_strides_and_degenerate_shapes(kpl, oracle):
    from tests.test_oracle_organized_normals import depth_image
    W, H = 57, 41
    xyz = depth_image(W, H, seed=3, step=30, holes=8)
    rec = np.zeros((W * H, 4), np.float32)               # pcl::PointXYZ: 16-byte records
    rec[:, :3] = xyz
    det = kpl.KeypointLearningDetector()
    nrm, _ = det.estimateNormalsOrganized(rec, W, H)
    assert same_nan(nrm, oracle.integral_image_normals(xyz, W, H)[0])
    for w, h in ((9, 40), (40, 10), (1, 1)):             # no pixel is further than the border from an edge
        small = depth_image(w, h, seed=1)
        n2, c2 = det.estimateNormalsOrganized(small, w, h)
        assert np.isnan(n2).all() and np.isnan(c2).all()
    n0, _ = det.estimateNormalsOrganized(np.zeros((0, 3), np.float32), 0, 0)
    assert n0.shape == (0, 3)


def test_organized_normals_on_device_buffers_in_pcl_layouts(kpl, oracle):
    """kpl_estimate_normals_organized_device: pcl::PointXYZ records in (16 bytes), pcl::Normal records out (32 bytes,
    curvature at +16) -- the buffers a caller with a device-resident organized cloud already has; asynchronous."""
    import torch
    from tests.test_oracle_organized_normals import depth_image
    W, H = 120, 90
    xyz = depth_image(W, H, seed=9, step=70, holes=15)
    rec = np.zeros((W * H, 4), np.float32)
    rec[:, :3] = xyz
    dev = torch.device("cuda", 0)
    dx = torch.from_numpy(rec).to(dev)
    dn = torch.full((W * H, 8), 7.0, dtype=torch.float32, device=dev)
    det = kpl.KeypointLearningDetector()
    st = torch.cuda.current_stream().cuda_stream
    det.estimateNormalsOrganizedDevice(dx.data_ptr(), 16, W, H, 5.0, (0.0, 0.0, 0.0), dn.data_ptr(), 32, dn.data_ptr() + 16, 32, st)
    torch.cuda.synchronize()
    out = dn.cpu().numpy()
    o_nrm, _ = oracle.integral_image_normals(xyz, W, H, 5.0)
    assert same_nan(out[:, :3], o_nrm)
    assert np.isnan(out[:, 4]).all()                                   # curvature
    assert (out[:, 3] == 7.0).all() and (out[:, 5:] == 7.0).all()       # nothing else of the records is touched


def test_organized_normals_equal_the_committed_golden(kpl):
    """the device kernels against tests/golden/organized_case.npz (tools/make_organized_golden.py; README item 6)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "organized_case.npz"))
    from tests.test_oracle_organized_normals import same_bits
    det = kpl.KeypointLearningDetector()
    W, H = int(z["width"]), int(z["height"])
    for name in ("origin", "off"):
        nrm, curv = det.estimateNormalsOrganized(z["xyz"], W, H, float(z["smoothing"]), tuple(float(x) for x in z["viewpoint_" + name]))
        assert same_bits(nrm, z["normals_" + name]) and np.isnan(curv).all()
