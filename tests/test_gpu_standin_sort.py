"""The sorted mode's lists are ordered through 32-bit stand-ins of their keys (csrc/kernels.hip: sort_key_lists, wave_sort_store):
q(d2) << slot bits | slot.  The stand-ins decide the order only where the q of neighbours differ; these cases put equal and almost
equal distances NEXT to ordinary ones in the same neighborhoods -- some waves sort by the stand-ins, some by the 64-bit keys -- at
every list size (register lists, word lists, a wave per point), and a radius so small that the scale of q overflows.  Bit for bit
against the oracle's sorted order (FLANN: ascending (distance, index))."""
import numpy as np
import pytest

from tests.test_gpu_sorted import _score_both_ways, make_det

pytestmark = pytest.mark.gpu


def partly_quantized(seed, nx=150, ny=120):
    """a jittered surface whose points are, patch by patch: left alone / snapped to a 1/4 lattice (exact ties) / snapped and then
    moved by one part in 10^7 (distances that differ in their last bits: equal q, different d2)"""
    from tools import synth
    xyz, nrm = synth.make_cloud(nx, ny, seed=seed)
    rng = np.random.default_rng(seed)
    patch = (np.floor(xyz[:, 0] / 9.0) + 3 * np.floor(xyz[:, 1] / 7.0)).astype(np.int64) % 3
    snapped = (np.round(xyz * 4.0) / 4.0).astype(np.float32)
    nudged = (snapped * (1.0 + rng.integers(-2, 3, size=snapped.shape) * 1.2e-7)).astype(np.float32)
    out = np.where((patch == 1)[:, None], snapped, xyz)
    out = np.where((patch == 2)[:, None], nudged, out).astype(np.float32)
    return synth.shuffle_cloud(out, nrm, 77 + seed)


@pytest.mark.parametrize("rmul,kf_min", [(2.5, 8), (5.0, 40), (8.5, 110), (12.0, 230), (17.0, 480)])
def test_equal_and_almost_equal_distances_among_ordinary_ones(kpl, oracle, cases, rmul, kf_min):
    xyz, nrm = partly_quantized(3)
    r = float(np.float32(rmul * oracle.cloud_resolution(xyz)))
    kf = _score_both_ways(kpl, oracle, cases, xyz, nrm, 5, 6, r, 21)
    assert kf > kf_min, kf
    # a second call of the same handle state (hints measured by the first) is covered by _score_both_ways' feature call


def test_a_radius_whose_scale_overflows(kpl, oracle, cases):
    """r2 ~ 4e-33: 2^24 / r2 is not a float -- the lists are sorted by their 64-bit keys"""
    xyz, nrm = cases.cloud()
    s = np.float32(1e-17)
    xyz = (xyz * s).astype(np.float32)
    r = float(np.float32(6 * cases.resolution()) * s)
    assert 0 < np.float32(r) * np.float32(r) < 5e-32
    det = make_det(kpl, 5, 6, r, 0.0, 0.5)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    q = np.arange(0, len(xyz), 2, dtype=np.int32)
    q = q[np.isfinite(nrm[q]).all(axis=1)]
    want = oracle.Grid(xyz, r).features(nrm, 5, 6, r, q, order=oracle.ORDER_SORTED)
    assert np.abs(want).sum() > 0
    assert cases.same_bits(det.computePointsForTrainingFeatures(q), want)
    kf = _score_both_ways(kpl, oracle, cases, xyz, nrm, 5, 6, r, 22)
    assert kf > 30, kf


def test_a_stream_of_lattice_views_goes_straight_to_the_exact_kernels(kpl, oracle, cases):
    """every neighborhood of a lattice holds equal distances: the kernels that sort by stand-ins hand every point on.  The
    handle notices (most points listed although their lists held them) and lists every point at once for the next launches."""
    from tests.test_oracle_sorted import lattice
    from tools import synth
    xyz, nrm = lattice(40, 36, dup=40)
    r = 3.3
    fa = synth.random_forest(30, ntrees=6, max_depth=8, seed=5, target_nodes_per_tree=120)
    det = make_det(kpl, 5, 6, r, 0.0, 0.0, fa)
    det.setNonMaxima(False)
    want, _ = oracle.detect(xyz, nrm, 5, 6, r, 0.0, 0.0, cases.oracle_forest(fa), non_maxima=False, order=oracle.ORDER_SORTED,
                            threads=cases.usable_cores())
    record = []
    for k in range(5):
        det.setInputCloud(xyz)
        det.setNormals(nrm)
        _, scores = det.compute()
        assert cases.same_bits(scores, want), k
        record.append(det.getLastLaunch()["sorted_all_large"])
    assert 0 in record[:2] and record[2:] == [1, 1, 1], record


@pytest.mark.parametrize("A,B", [(8, 10), (1, 1), (15, 17), (3, 2)])
@pytest.mark.parametrize("rmul,mode", [(5.0, (-1, 0)), (9.0, (1, 256)), (12.5, (1, 512))])
def test_histogram_shapes_through_every_sorted_kernel(kpl, oracle, cases, A, B, rmul, mode):
    """whole views (compute(), four calls on one handle: the last ones run the kernel the handle settled on -- the position-list
    kernel, the word lists with 256 and with 512 positions) at histogram shapes from one cell to the 255-cell limit"""
    from tools import synth
    xyz, nrm = synth.make_cloud(110, 90, seed=8, nan_points=7, nan_normals=9)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1234)
    r = float(np.float32(rmul * oracle.cloud_resolution(xyz)))
    fa = synth.random_forest(A * B, ntrees=6, max_depth=8, seed=30 + A, target_nodes_per_tree=120)
    det = make_det(kpl, A, B, r, 0.0, 0.0, fa)
    det.setNonMaxima(False)
    want, _ = oracle.detect(xyz, nrm, A, B, r, 0.0, 0.0, cases.oracle_forest(fa), non_maxima=False, order=oracle.ORDER_SORTED,
                            threads=cases.usable_cores())
    for k in range(4):
        det.setInputCloud(xyz)
        det.setNormals(nrm)
        _, scores = det.compute()
        assert cases.same_bits(scores, want), k
    ll = det.getLastLaunch()
    assert ll["walk"] == mode[0] and ll["sorted_all_large"] == 0, ll
    if mode[1]:
        assert ll["sorted_list_keys"] == mode[1], ll
