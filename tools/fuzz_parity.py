"""Randomised parity soak: random small clouds, radii, histogram shapes, forests, thresholds and NMS modes through
libkpl and through the oracle, in the canonical and (40 % of the cases) the sorted neighbor order, with the walk of the
canonical order forced at random (one kernel / two passes, two / four lanes per point, or the handle's own choice: round 5);
every score must match bit for bit and every keypoint list exactly.  Every 25th case also
runs a random organized depth image (steps, holes, non-finite x) through the integral-image normal estimation.
    python tools/fuzz_parity.py [seconds] [seed] [--log FILE]
A mismatch is reported with everything needed to classify it (scores vs list vs count, how many, where), the device's
outputs are saved next to the inputs, the same call is repeated on the same handle AND on a fresh handle, and the state of
the random generator BEFORE the failing case is printed (and, with --log, appended per case to FILE) so that the case -- and
the ones before it -- can be regenerated without replaying the whole run:
    rng = np.random.default_rng(); rng.bit_generator.state = <the printed dict>
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo  # noqa: E402
from tests import helpers  # noqa: E402
from tools import synth  # noqa: E402


def random_view(rng):
    n = int(rng.integers(0, 1500))
    if rng.random() < 0.5 and n > 20:
        nx = max(2, int(np.sqrt(n)))
        xyz, nrm = synth.make_cloud(nx, max(2, n // nx), seed=int(rng.integers(1, 1 << 30)), nan_points=int(rng.integers(0, 3)))
    else:
        xyz = (rng.uniform(-1, 1, size=(n, 3)) * rng.uniform(1, 20)).astype(np.float32)
        nrm = rng.normal(size=(n, 3)).astype(np.float32)
    return np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3), np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)


def batch_case(kpl, rng):
    """2..8 random views with their own shapes, forests and NMS modes through kpl_compute_batch_device"""
    import torch
    dev = torch.device("cuda", 0)
    k = int(rng.integers(2, 9))
    dets, bufs, expect = [], [], []
    for _ in range(k):
        xyz, nrm = random_view(rng)
        n = len(xyz)
        A, B = [(5, 6), (5, 10), (2, 7), (1, 1)][int(rng.integers(0, 4))]
        mr = kplo.cloud_resolution(xyz) if n > 1 else 1.0
        mr = mr if mr > 0 else 1.0
        r, rn = float(np.float32(mr * rng.uniform(2, 7))), float(np.float32(mr * rng.uniform(0.5, 5)))
        thr = float(np.float32(rng.choice([0.0, 0.5, 0.85])))
        nms, draws = bool(rng.random() < 0.8), bool(rng.random() < 0.4)
        srt = bool(rng.random() < 0.4)
        dthr = float(np.float32(mr * rng.uniform(0, 3)))
        fa = synth.random_forest(A * B, ntrees=int(rng.integers(1, 12)), max_depth=int(rng.integers(1, 10)),
                                 seed=int(rng.integers(1, 1 << 30)), target_nodes_per_tree=int(rng.integers(3, 200)))
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(nms); det.setNonMaxRadius(rn)
        det.setNonMaximaDrawsRemove(draws); det.setNonMaximaDrawsThreshold(dthr)
        det.setPredictionThreshold(thr); det.setRadiusSearch(r); det.setSortedSearch(srt)
        det.setFeatureWalk(*WALKS[int(rng.integers(0, len(WALKS)))])
        helpers.load_arrays(det, fa)
        dx = torch.from_numpy(xyz).to(dev) if n else torch.zeros(1, 3, device=dev)
        dn = torch.from_numpy(nrm).to(dev) if n else torch.zeros(1, 3, device=dev)
        ds = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det); bufs.append((dx, dn, ds, dk, n))
        expect.append(kplo.detect(xyz, nrm, A, B, r, rn, thr, helpers.oracle_forest(fa), non_maxima=nms,
                                  draws_remove=draws, draws_threshold=dthr, order=kplo.ORDER_SORTED if srt else kplo.ORDER_CANONICAL))
    args = (dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() if b[4] else None for b in bufs],
            [b[4] for b in bufs], [b[3][0:1].data_ptr() for b in bufs])
    for attempt in range(4):
        kpl.compute_batch_device(*args, None)
        torch.cuda.synchronize()
        st = [d.syncStatus(None) for d in dets]
        if all(x == kpl.OK for x in st):
            break
    for (dx, dn, ds, dk, n), (o_sc, o_kp) in zip(bufs, expect):
        cnt = int(dk[0].item())
        if not (helpers.same_bits(ds.cpu().numpy()[:n], o_sc) and np.array_equal(dk[1:1 + cnt].cpu().numpy(), o_kp)):
            return False
    return True


def organized_case(det, rng):
    """a random depth image (smooth surface + random steps, NaN holes, NaN columns) through
    kpl_estimate_normals_organized and through the oracle's pcl::IntegralImageNormalEstimation restatement"""
    W, H = int(rng.integers(1, 180)), int(rng.integers(1, 140))
    v, u = np.mgrid[0:H, 0:W].astype(np.float32)
    z = (rng.uniform(0.5, 5.0) + rng.uniform(-0.01, 0.01) * u + rng.uniform(-0.01, 0.01) * v +
         rng.uniform(0, 0.2) * np.sin(u / rng.uniform(2, 12)) * np.cos(v / rng.uniform(2, 12))).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):           # depth steps
        c = int(rng.integers(0, max(W, 1)))
        z[:, c:] += np.float32(rng.uniform(-1.0, 1.0))
    xyz = np.stack([(u - W / 2) * z / 300.0, (v - H / 2) * z / 300.0, z], -1).astype(np.float32)
    for _ in range(int(rng.integers(0, 30))):          # holes
        r, c = int(rng.integers(0, H)), int(rng.integers(0, W))
        xyz[r:r + int(rng.integers(1, 5)), c:c + int(rng.integers(1, 5))] = np.nan
    if rng.random() < 0.2:
        xyz[:, :, 0] = np.where(rng.random((H, W)) < 0.02, np.inf, xyz[:, :, 0])      # x not finite, z finite
    xyz = xyz.reshape(-1, 3)
    smoothing = float(rng.choice([2.5, 3.5, 5.0, 7.0, 10.0]))
    vp = tuple(float(x) for x in rng.uniform(-3, 3, size=3)) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
    nrm, curv = det.estimateNormalsOrganized(xyz, W, H, smoothing, vp)
    o_nrm, _ = kplo.integral_image_normals(xyz, W, H, smoothing, vp)
    nan_a, nan_b = np.isnan(nrm), np.isnan(o_nrm)
    ok = np.array_equal(nan_a, nan_b) and np.array_equal(nrm.view(np.uint32)[~nan_a], o_nrm.view(np.uint32)[~nan_b]) and \
        bool(np.isnan(curv).all())
    if not ok:
        np.savez(failure_path("fuzz_failure_organized.npz"), xyz=xyz, W=W, H=H, smoothing=smoothing, vp=np.float32(vp))
    return ok


def failure_path(name="fuzz_failure.npz"):
    """under gpurun_out/ when it exists (what gpurun merges back), else the working directory"""
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d, name) if os.path.isdir(d) else name


WALKS = [(-1, 2), (0, 2), (0, 4), (1, 2), (1, 4)]      # (kpl_set_feature_walk: AUTO / LANES / TWO_PASS, lanes per point)


def configure(det, A, B, nms, rn, draws, dthr, thr, r, srt, fa, walk=(-1, 2)):
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(nms); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(draws); det.setNonMaximaDrawsThreshold(dthr)
    det.setPredictionThreshold(thr); det.setRadiusSearch(r); det.setSortedSearch(srt)
    det.setFeatureWalk(*walk)
    helpers.load_arrays(det, fa)


def main():
    argv = [a for a in sys.argv[1:]]
    log = None
    if "--log" in argv:
        k = argv.index("--log")
        log = open(argv[k + 1], "a")
        del argv[k:k + 2]
    # --replay LOGFILE CASE: start from the generator state a --log file recorded before case CASE (a multiple of 50) -- on a
    # fresh handle, so hint-driven choices may differ from the run that wrote the log; --trace: every case's parameters before its call
    replay, trace = None, "--trace" in argv
    if trace:
        argv.remove("--trace")
    if "--replay" in argv:
        k = argv.index("--replay")
        replay = (argv[k + 1], int(argv[k + 2]))
        del argv[k:k + 3]
    budget = float(argv[0]) if len(argv) > 0 else 60.0
    seed = int(argv[1]) if len(argv) > 1 else 1
    kpl = importlib.import_module("keypoint-learning_amd")
    rng = np.random.default_rng(seed)
    det = kpl.KeypointLearningDetector()
    t0, cases, points = time.time(), 0, 0
    if replay is not None:
        rows = [json.loads(l) for l in open(replay[0])]
        row = [r for r in rows if r["case"] == replay[1]][-1]
        rng.bit_generator.state = row["rng"]
        seed, cases = row["seed"], row["case"]
    while time.time() - t0 < budget:
        state_before = rng.bit_generator.state          # regenerates this case (and what follows) without the run before it
        if log is not None and cases % 50 == 0:
            log.write(json.dumps({"seed": seed, "case": cases, "rng": state_before}) + "\n")
            log.flush()
        kind = rng.integers(0, 5)
        if kind == 0:
            nx, ny = int(rng.integers(1, 70)), int(rng.integers(1, 60))
            xyz, nrm = synth.make_cloud(nx, ny, seed=int(rng.integers(1, 1 << 30)), nan_points=int(rng.integers(0, 4)),
                                        nan_normals=int(rng.integers(0, 4)), overlap_layers=int(rng.integers(1, 4)))
        elif kind == 1:                                # random volume
            n = int(rng.integers(0, 2500))
            xyz = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) * np.float32(rng.uniform(0.5, 30))
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
        elif kind == 2:                                # lattice with exact ties and duplicates
            g = np.stack(np.meshgrid(np.arange(int(rng.integers(2, 25))), np.arange(int(rng.integers(2, 25))), [0.0]), -1)
            xyz = g.reshape(-1, 3).astype(np.float32)
            xyz = np.concatenate([xyz, xyz[: len(xyz) // 3]])
            nrm = np.tile(np.float32([[0, 0, 1]]), (len(xyz), 1))
            nrm[::7] = np.float32([0, 0.6, 0.8])
        elif kind == 3:                                # far from the origin, anisotropic
            n = int(rng.integers(10, 2000))
            xyz = (rng.uniform(0, 1, size=(n, 3)) * [200, 3, 0.5] + [5e4, -3e4, 100]).astype(np.float32)
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
        else:                                          # clumps: very uneven cell populations
            n = int(rng.integers(50, 3000))
            centres = rng.uniform(-20, 20, size=(6, 3))
            xyz = (centres[rng.integers(0, 6, size=n)] + rng.normal(0, rng.uniform(0.05, 2.0), size=(n, 3))).astype(np.float32)
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
        if len(xyz) and rng.random() < 0.5:
            xyz, nrm = synth.shuffle_cloud(xyz, nrm, int(rng.integers(1, 1 << 30)))
        n = len(xyz)
        A, B = [(5, 6), (5, 10), (8, 10), (1, 1), (2, 7), (15, 17), (3, 3)][int(rng.integers(0, 7))]
        mr = kplo.cloud_resolution(xyz) if n > 1 else 1.0
        mr = mr if mr > 0 else 1.0
        r = float(np.float32(mr * rng.uniform(1.5, 9.0)))
        if rng.random() < 0.15:                        # large neighborhoods: hundreds of points per cell (what the two-pass walk is for)
            r = float(np.float32(mr * rng.uniform(9.0, 30.0)))
        walk = WALKS[int(rng.integers(0, len(WALKS)))]
        # one case in 400: a radius so small against the extent of the cloud that the grid has 1e8 .. 2.7e8 cells (the limit is
        # 2^28): a cell table of ~1 GB that the handle has to grow -- the class of view behind the fuzz events of rounds 3 and 4
        # (an asynchronous clear of that table raced with the index build, profiles/r04_notes.md section 1)
        if n > 50 and rng.random() < 0.0025:
            fin = np.isfinite(xyz).all(axis=1)
            if fin.sum() > 10:
                ext = (xyz[fin].max(axis=0) - xyz[fin].min(axis=0)).astype(np.float64)
                vol = float(np.prod(np.maximum(ext, 1e-9)))
                if vol > 0:
                    r = float(np.float32((vol / rng.uniform(1.0e8, 2.7e8)) ** (1.0 / 3.0)))
        rn = float(np.float32(mr * rng.uniform(0.0, 6.0)))
        thr = float(np.float32(rng.choice([0.0, 0.3, 0.5, 0.85, 1.0])))
        nms, draws = bool(rng.random() < 0.85), bool(rng.random() < 0.4)
        srt = bool(rng.random() < 0.4)                 # neighbors in sorted (distance, index) order
        dthr = float(np.float32(mr * rng.uniform(0, 4)))
        many = rng.random() < 0.15                     # many trees: out-of-step walks; chained layout when A * B >= 32 (forest.h)
        fa = synth.random_forest(A * B, ntrees=int(rng.integers(40, 90)) if many else int(rng.integers(1, 14)),
                                 max_depth=int(rng.integers(1, 20 if many else 12)),
                                 seed=int(rng.integers(1, 1 << 30)), target_nodes_per_tree=int(rng.integers(3, 1500 if many else 400)))
        if rng.random() < 0.5:                         # coarse leaf values: many exact score ties
            fa.value[:] = np.round(fa.value * 2) / 2
        configure(det, A, B, nms, rn, draws, dthr, thr, r, srt, fa, walk)
        det.setInputCloud(np.ascontiguousarray(xyz).reshape(-1, 3)); det.setNormals(np.ascontiguousarray(nrm).reshape(-1, 3))
        if trace:
            print("case %d kind %d n %d A %d B %d r %g rn %g thr %g nms %d draws %d sorted %d walk %s trees %d" %
                  (cases, kind, n, A, B, r, rn, thr, nms, draws, srt, walk, fa.ntrees), flush=True)
        try:
            _, sc = det.compute()
        except kpl.KplError as e:
            if e.status == kpl.ERR_GRID_TOO_LARGE:
                continue
            raise
        o_sc, o_kp = kplo.detect(xyz, nrm, A, B, r, rn, thr, helpers.oracle_forest(fa), non_maxima=nms,
                                 draws_remove=draws, draws_threshold=dthr, order=kplo.ORDER_SORTED if srt else kplo.ORDER_CANONICAL)
        ok = helpers.same_bits(sc, o_sc) and np.array_equal(det.getKeypointsIndices(), o_kp)
        if trace:
            print("  first call done", flush=True)
        if ok and rng.random() < 0.35:
            # the same view again on the same handle: now with what the first call measured -- the walk, the accept words per
            # point, the list capacity and the all-large switch of the sorted mode all follow the handle's own history
            _, sc_again = det.compute()
            ok = helpers.same_bits(sc_again, o_sc) and np.array_equal(det.getKeypointsIndices(), o_kp)
            if not ok:
                sc = sc_again
                print("  (the FIRST call of the handle was right; this is its second, hint-driven one: %s)" % (det.getTiming(),))
        if not ok:
            kp = det.getKeypointsIndices()
            out = failure_path()
            np.savez(out, xyz=xyz, nrm=nrm, A=A, B=B, r=r, rn=rn, thr=thr, nms=nms, draws=draws, dthr=dthr, srt=srt,
                     root=fa.root, var=fa.var, thrs=fa.thr, left=fa.left, right=fa.right, value=fa.value,
                     dev_scores=sc, dev_kp=kp, ora_scores=o_sc, ora_kp=o_kp)
            bad = np.nonzero(~((helpers.bits(sc) == helpers.bits(o_sc)) | (np.isnan(sc) & np.isnan(o_sc))))[0] if len(sc) == len(o_sc) else []
            print("MISMATCH seed %d case %d kind %d n %d A %d B %d r %g rn %g thr %g nms %d draws %d sorted %d walk %s trees %d nodes %d -> %s"
                  % (seed, cases, kind, n, A, B, r, rn, thr, nms, draws, srt, walk, fa.ntrees, len(fa.var), out))
            print("  WHAT DIFFERED: scores %s (%d of %d differ, first at %s); keypoint COUNT device %d oracle %d; LIST %s; only device %s only oracle %s"
                  % ("same" if len(bad) == 0 else "DIFFER", len(bad), n, bad[:6], len(kp), len(o_kp),
                     "same" if np.array_equal(kp, o_kp) else "DIFFERS", np.setdiff1d(kp, o_kp)[:8], np.setdiff1d(o_kp, kp)[:8]))
            for i in bad[:6]:
                print("    score[%d]: device %.9g (0x%08x) oracle %.9g (0x%08x)" % (i, sc[i], helpers.bits(sc)[i], o_sc[i], helpers.bits(o_sc)[i]))
            print("  rng state before the case:", json.dumps(state_before))
            _, sc2 = det.compute()
            print("  the same call again, same handle: scores %s keypoints %s" % (helpers.same_bits(sc2, o_sc), np.array_equal(det.getKeypointsIndices(), o_kp)))
            fresh = kpl.KeypointLearningDetector()
            configure(fresh, A, B, nms, rn, draws, dthr, thr, r, srt, fa, walk)
            fresh.setInputCloud(np.ascontiguousarray(xyz).reshape(-1, 3)); fresh.setNormals(np.ascontiguousarray(nrm).reshape(-1, 3))
            _, sc3 = fresh.compute()
            print("  the same call on a fresh handle: scores %s keypoints %s" % (helpers.same_bits(sc3, o_sc), np.array_equal(fresh.getKeypointsIndices(), o_kp)))
            return 1
        if rng.random() < 0.25 and n > 0:               # the preparation steps as well
            k = int(rng.integers(3, 33))
            if trace:
                print("  preparation steps, k %d" % k, flush=True)
            nk, ck = det.estimateNormals(xyz, k=k, viewpoint=(1.0, 2.0, 300.0))
            o_nk, o_ck = kplo.estimate_normals(xyz, k=k, viewpoint=(1.0, 2.0, 300.0))
            nr, cr = det.estimateNormals(xyz, k=0, radius=r)
            o_nr, o_cr = kplo.estimate_normals(xyz, k=0, radius=r)
            res_ok = det.cloudResolution(xyz) == kplo.cloud_resolution(xyz)
            if not (helpers.same_bits(nk, o_nk) and helpers.same_bits(ck, o_ck) and helpers.same_bits(nr, o_nr)
                    and helpers.same_bits(cr, o_cr) and res_ok):
                np.savez(failure_path("fuzz_failure_normals.npz"), xyz=xyz, k=k, r=r)
                print("MISMATCH in normals / resolution: case %d kind %d n %d k %d r %g -> fuzz_failure.npz" % (cases, kind, n, k, r))
                return 1
        if trace and cases % 25 == 24:
            print("  organized case", flush=True)
        if cases % 25 == 24 and not organized_case(det, rng):
            print("MISMATCH in the normals of an organized cloud after case %d -> fuzz_failure.npz" % cases)
            return 1
        if trace and cases % 40 == 39:
            print("  batch case", flush=True)
        if cases % 40 == 39 and not batch_case(kpl, rng):
            print("MISMATCH in a batched call after case %d" % cases)
            return 1
        cases += 1
        points += n
    print("fuzz parity: %d cases, %d points, all bit-exact (seed %d, %.0f s)" % (cases, points, seed, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
