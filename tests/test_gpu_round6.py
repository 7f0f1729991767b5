"""Round 6: the hints a handle carries from call to call must follow the views (advisor findings of round 5), kpl_reserve,
the side-effect-free launch record, and a handle that changes streams."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _det(kpl, cases, A, B, r, rn, thr, sorted_search):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
    det.setSortedSearch(sorted_search)
    cases.load_arrays(det, cases.trained_forest(A, B))
    return det


def test_sorted_mode_hints_follow_a_stream_that_goes_from_dense_to_sparse(kpl, oracle, cases):
    """One handle, one radius, views of one size: dense frames switch the handle to "every point straight to the collect / add
    kernels" (all_large); that path measures no neighborhood lengths, and until round 5 the hint then stayed for good.  Now the
    keys such a launch stores per listed point are looked at: sparse frames bring the register lists back."""
    A, B = 5, 6
    dense, nrm = cases.cloud(120, 100, seed=3)
    mr = oracle.cloud_resolution(dense)
    # (dense: ~500 neighbors per point -- beyond what the word lists take, see the next test)
    r, rn, thr = float(np.float32(17.0 * mr)), float(np.float32(4 * mr)), float(np.float32(0.5))
    sparse = np.ascontiguousarray(dense * np.float32([3.0, 3.0, 1.0]))       # the same points, a ninth of the density
    det = _det(kpl, cases, A, B, r, rn, thr, True)
    of = cases.oracle_forest(cases.trained_forest(A, B))
    want = {}
    for name, xyz in (("dense", dense), ("sparse", sparse)):
        want[name] = oracle.detect(xyz, nrm, A, B, r, rn, thr, of, order=oracle.ORDER_SORTED, threads=cases.usable_cores())
    record = []
    for name, xyz in [("dense", dense)] * 3 + [("sparse", sparse)] * 4 + [("dense", dense)] * 3:
        det.setInputCloud(xyz)
        det.setNormals(nrm)
        _, scores = det.compute()
        assert cases.same_bits(scores, want[name][0]), (name, len(record))
        assert np.array_equal(det.getKeypointsIndices(), want[name][1]), (name, len(record))
        record.append((name, det.getLastLaunch()["sorted_all_large"], det.getLastLaunch()["sorted_list_keys"]))
    assert record[2][1] == 1, record                     # dense frames: every point listed at once
    assert record[3][1] == 1, record                     # the first sparse frame still runs with the old hint ...
    assert record[5][1] == 0 and record[6][1] == 0, record   # ... the later ones do not
    assert record[6][2] < 128, record                    # and the list capacity follows what the register sort measured again
    assert record[9][1] == 1, record                     # dense again: back to the wave / workgroup kernels


def test_sorted_mode_takes_the_kernel_that_fits_the_neighborhoods(kpl, oracle, cases):
    """one handle, three radii: ~60 neighbors per point -> the register / position lists of feature_sorted_kernel; ~150 -> the word
    lists with 256 positions per point; ~300 -> with 512.  Every call bit-exact, the third call at a radius in its mode."""
    A, B = 5, 6
    xyz, nrm = cases.cloud(120, 100, seed=4)
    mr = oracle.cloud_resolution(xyz)
    det = _det(kpl, cases, A, B, 1.0, 0.0, 0.5, True)
    det.setNonMaxima(False)
    of = cases.oracle_forest(cases.trained_forest(A, B))
    seen = {}
    for rmul in (5.5, 9.0, 12.5, 5.5):
        r = float(np.float32(rmul * mr))
        det.setRadiusSearch(r)
        want = oracle.detect(xyz, nrm, A, B, r, 0.0, float(np.float32(0.5)), of, non_maxima=False, order=oracle.ORDER_SORTED,
                             threads=cases.usable_cores())[0]
        for k in range(3):
            det.setInputCloud(xyz)
            det.setNormals(nrm)
            _, scores = det.compute()
            assert cases.same_bits(scores, want), (rmul, k)
        ll = det.getLastLaunch()
        seen[rmul] = (ll["walk"], ll["sorted_list_keys"], ll["sorted_all_large"])
    assert seen[5.5][0] == -1 and seen[5.5][1] <= 128 and seen[5.5][2] == 0, seen
    assert seen[9.0] == (1, 256, 0), seen
    assert seen[12.5] == (1, 512, 0), seen


def test_reserve_sizes_a_handle_before_its_first_call(kpl, oracle, cases):
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.85))
    det = _det(kpl, cases, A, B, r, rn, thr, False)
    det.reserve(len(xyz), 12, 12)
    det.reserve(len(xyz) // 2, 16, 32)                   # smaller: nothing shrinks, nothing breaks
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(cases.trained_forest(A, B)))
    assert cases.same_bits(scores, o_scores) and np.array_equal(det.getKeypointsIndices(), o_kp)
    det.reserve(4 * len(xyz), 12, 12)                    # growing under a bound host view leaves the view alone
    _, scores = det.compute()
    assert cases.same_bits(scores, o_scores)
    lib = kpl.load_library()
    assert lib.kpl_reserve(det._h, -1, 12, 12) == kpl.ERR_INVALID_ARG
    assert lib.kpl_reserve(det._h, 10, 7, 12) == kpl.ERR_INVALID_ARG
    assert lib.kpl_reserve(None, 10, 12, 12) == kpl.ERR_INVALID_ARG


def test_last_launch_is_a_plain_read_that_leaves_the_timing_alone(kpl, cases):
    import torch
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    det = _det(kpl, cases, A, B, float(np.float32(6 * mr)), float(np.float32(4 * mr)), 0.85, False)
    dev = torch.device("cuda", 0)
    dx, dn = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev), torch.from_numpy(np.ascontiguousarray(nrm)).to(dev)
    dk = torch.zeros(len(xyz) + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
    det.enableTiming(True)
    for _ in range(3):
        det.computeDevice(None, dk[1:].data_ptr(), len(xyz), dk[0:1].data_ptr())
    assert det.syncStatus(None) == kpl.OK
    li = det.getLastLaunch()
    assert li["walk"] == kpl.WALK_LANES and li["lanes_per_point"] == 2 and li["accept_words"] in (12, 16, 20, 24)
    assert det.getLastLaunch() == li
    t = det.getTiming()                                  # still everything that was recorded
    assert t["calls"] == 3 and t["feature_ms"] > 0.0 and t["walk"] == li["walk"]


def test_a_handle_that_moves_between_streams_is_ordered_without_a_host_wait(kpl, oracle, cases):
    """device entry point on a caller's stream, then -- without waiting -- a host entry point on the handle's own stream with
    another view: the second call grows and rewrites the handle's scratch while the first may still be running on the other
    stream (advisor finding, round 5).  Both results must be the oracle's."""
    import torch
    A, B = 5, 6
    big, bn = cases.cloud(160, 120, seed=5)
    small, sn = cases.cloud()
    mr = cases.resolution()
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.5))
    of = cases.oracle_forest(cases.trained_forest(A, B))
    w_small = oracle.detect(small, sn, A, B, r, rn, thr, of)
    w_big = oracle.detect(big, bn, A, B, r, rn, thr, of)
    dev = torch.device("cuda", 0)
    ds, dsn = torch.from_numpy(np.ascontiguousarray(small)).to(dev), torch.from_numpy(np.ascontiguousarray(sn)).to(dev)
    sc = torch.zeros(len(small), dtype=torch.float32, device=dev)
    dk = torch.zeros(len(small) + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep in range(6):
        det = _det(kpl, cases, A, B, r, rn, thr, False)
        det.bindCloudDevice(ds.data_ptr(), 12, dsn.data_ptr(), 12, len(small))
        det.computeDevice(sc.data_ptr(), dk[1:].data_ptr(), len(small), dk[0:1].data_ptr(), side.cuda_stream)
        det.setInputCloud(big)                           # host view, larger: every table of the handle is replaced
        det.setNormals(bn)
        _, scores = det.compute()
        assert cases.same_bits(scores, w_big[0]) and np.array_equal(det.getKeypointsIndices(), w_big[1]), rep
        side.synchronize()
        k = int(dk[0].item())
        assert k == len(w_small[1]) and np.array_equal(dk[1:1 + k].cpu().numpy(), w_small[1]), rep
        assert cases.same_bits(sc.cpu().numpy(), w_small[0]), rep


def test_every_fresh_handle_of_a_process_counts_its_keypoints(kpl, oracle, cases):
    """The one unexplained event of round 6 (profiles/r06_notes.md): with a stream-ordered scratch block cleared and freed in a
    handle's set-up, the FIRST compute() of every second handle of a process returned no keypoints (scores right, second
    call right).  Handles are created, used once through the keypoints-only entry point and destroyed, in the order that
    failed (sorted, sorted, canonical without NMS, canonical with NMS, ...)."""
    A, B = 5, 6
    xyz, nrm = cases.cloud(40, 40, seed=7, nan_points=5, nan_normals=7)
    mr = oracle.cloud_resolution(xyz)
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    of = cases.oracle_forest(cases.trained_forest(A, B))
    want = {}
    for srt in (False, True):
        for thr in (0.0, 0.5):
            want[(srt, thr)] = oracle.detect(xyz, nrm, A, B, r, rn, float(np.float32(thr)), of,
                                             order=oracle.ORDER_SORTED if srt else oracle.ORDER_CANONICAL)
    scoreable = int(np.isfinite(want[(False, 0.0)][0]).sum())
    for rep in range(3):
        for srt, thr, nms in [(True, 0.5, False), (True, 0.5, True), (False, 0.0, False), (False, 0.0, True),
                              (False, 0.5, True), (False, 0.5, False)]:
            det = _det(kpl, cases, A, B, r, rn, float(np.float32(thr)), srt)
            det.setNonMaxima(nms)
            det.setInputCloud(xyz)
            det.setNormals(nrm)
            kp, _ = det.compute(with_scores=False)
            idx = det.getKeypointsIndices()
            if nms:
                assert np.array_equal(idx, want[(srt, thr)][1]), (rep, srt, thr, nms, len(idx))
            else:
                assert len(idx) == scoreable, (rep, srt, thr, nms, len(idx))
            del det


@pytest.mark.parametrize("rmul,nan", [(8.5, False), (10.0, True)])
def test_sorted_order_through_the_word_lists(kpl, oracle, cases, rmul, nan):
    """125 .. ~250 neighbors per point in sorted order: once a handle has seen that its points overflow the register lists and
    hold ~100-200 keys each, it searches with the two-pass walk's search kernel and sorts 256 keys per point in the registers
    of eight lanes (sorted_words_kernel).  Every call -- before, at and after the switch -- gives the oracle's bits."""
    A, B = 5, 6
    xyz, nrm = cases.cloud(150, 120, seed=11, nan_points=6 if nan else 0, nan_normals=9 if nan else 0)
    mr = oracle.cloud_resolution(xyz)
    r, rn, thr = float(np.float32(rmul * mr)), float(np.float32(4 * mr)), float(np.float32(0.5))
    of = cases.oracle_forest(cases.trained_forest(A, B))
    want = oracle.detect(xyz, nrm, A, B, r, rn, thr, of, order=oracle.ORDER_SORTED, threads=cases.usable_cores())
    det = _det(kpl, cases, A, B, r, rn, thr, True)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    launches = []
    for rep in range(6):
        _, scores = det.compute()
        assert cases.same_bits(scores, want[0]), rep
        assert np.array_equal(det.getKeypointsIndices(), want[1]), rep
        launches.append(det.getLastLaunch())
    assert launches[0]["walk"] == -1                                  # the first call knows nothing
    assert launches[-1]["walk"] == kpl.WALK_TWO_PASS and launches[-1]["lanes_per_point"] == 8 and launches[-1]["sorted_list_keys"] == 256, launches
    # the query entry point (training features) is untouched by the handle's mode
    q = np.flatnonzero(np.isfinite(nrm).all(axis=1)).astype(np.int32)[::37]
    g = oracle.Grid(xyz, r)
    assert cases.same_bits(det.computePointsForTrainingFeatures(q), g.features(nrm, A, B, r, q, order=oracle.ORDER_SORTED))


def test_sorted_word_lists_with_exact_ties_and_points_beyond_the_list(kpl, oracle, cases):
    """a lattice with duplicates (hundreds of equal distances, broken by index) whose denser half overflows the 256 keys of a
    point's list: those points go on to the wave / workgroup kernels, the others are sorted in the eight lanes"""
    from tests.test_oracle_sorted import lattice
    A, B = 5, 6
    xyz, nrm = lattice(48, 40, dup=60)
    dense = xyz[: len(xyz) // 3] * np.float32([0.55, 0.55, 1.0]) + np.float32([100.0, 0.0, 0.0])     # a denser patch beside it
    xyz = np.ascontiguousarray(np.concatenate([xyz, dense]), np.float32)
    nrm = np.ascontiguousarray(np.concatenate([nrm, nrm[: len(dense)]]), np.float32)
    r, rn, thr = 7.3, 3.1, float(np.float32(0.5))
    of = cases.oracle_forest(cases.trained_forest(A, B))
    want = oracle.detect(xyz, nrm, A, B, r, rn, thr, of, order=oracle.ORDER_SORTED, threads=cases.usable_cores())
    det = _det(kpl, cases, A, B, r, rn, thr, True)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    seen = set()
    for rep in range(6):
        _, scores = det.compute()
        assert cases.same_bits(scores, want[0]), rep
        assert np.array_equal(det.getKeypointsIndices(), want[1]), rep
        seen.add(det.getLastLaunch()["walk"])
    assert kpl.WALK_TWO_PASS in seen, seen


def test_a_handle_lives_through_a_random_stream_of_views():
    """tools/fuzz_streams.py for a few seconds: one handle, views whose density / size / radius / order change, every call
    against the oracle (the hints a handle carries are choices of speed, never of result)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_streams.py"), "25", "4242"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["all_bit_exact"] and line["calls"] >= 20, line
