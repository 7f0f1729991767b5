"""One cloud split into slabs (what several GPUs would each get, SURVEY.md 8(e) "single huge cloud"): with the
whole cloud's grid origin every slab reproduces the whole-cloud result on its interior bit for bit."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parts,draws", [(3, False), (5, False), (2, True)])
def test_slabs_equal_the_whole_cloud(kpl, oracle, cases, parts, draws):
    kd = importlib.import_module("keypoint-learning_amd.dist")
    A, B = 5, 6
    xyz, nrm = cases.cloud(150, 90, seed=31, nan_points=9, nan_normals=5)
    fa = cases.trained_forest(A, B)
    mr = oracle.cloud_resolution(xyz)
    r, rn, thr = float(np.float32(5 * mr)), float(np.float32(3.5 * mr)), float(np.float32(0.6))
    dthr = float(np.float32(2 * mr))
    o_sc, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa), draws_remove=draws,
                               draws_threshold=dthr, threads=cases.usable_cores())
    origin, plans = kd.slab_plan(xyz, parts, (r + rn) * 1.001)
    scores, kps = [], []
    for plan in plans:                              # each iteration = what one rank does on its GPU
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
        det.setNonMaximaDrawsRemove(draws); det.setNonMaximaDrawsThreshold(dthr)
        det.setPredictionThreshold(thr); det.setRadiusSearch(r)
        cases.load_arrays(det, fa)
        det.setGridOrigin(origin)
        det.setInputCloud(np.ascontiguousarray(xyz[plan["idx"]]))
        det.setNormals(np.ascontiguousarray(nrm[plan["idx"]]))
        _, sc = det.compute()
        scores.append(sc)
        kps.append(det.getKeypointsIndices())
        assert len(plan["idx"]) < len(xyz)
    m_sc, m_kp = kd.merge_slabs(len(xyz), plans, scores, kps)
    assert cases.same_bits(m_sc, o_sc)
    if not draws:
        assert np.array_equal(m_kp, o_kp)
    else:
        # the greedy draws pass walks plateaus in index order across the whole cloud; slabs agree wherever a
        # plateau does not straddle a cut -- on this cloud with real-valued scores there are no ties at all
        assert np.array_equal(m_kp, o_kp)


def test_origin_above_the_view_minimum_is_refused(kpl, cases):
    xyz, nrm = cases.cloud(40, 30, seed=3)
    fa = cases.trained_forest(5, 6)
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(5); det.setNBins(6); det.setNonMaxRadius(2.0); det.setRadiusSearch(3.0)
    cases.load_arrays(det, fa)
    det.setGridOrigin(xyz.min(axis=0) + np.float32([1, 0, 0]))
    det.setInputCloud(xyz); det.setNormals(nrm)
    with pytest.raises(kpl.KplError) as e:
        det.compute()
    assert "origin" in str(e.value)
    det.setGridOrigin(None)                          # automatic again
    det.compute()
