#!/usr/bin/env python3
"""train_detector.py -- counterpart of the reference's TrainDetector
(/root/reference/src/main_train_detector.cpp:153-519) on top of libkpl.

Per view of the dataset: load the cloud, drop non-finite points (:341-342), optional uniform
sampling (:345-349), normals with k-search or radius search (:356-364, kpl_estimate_normals on the
device), optional flip (:366-370), snap every positive / negative of the training set to its nearest
input point (:424-437), compute their features with the SAME device kernel the detector scores with
(computePointsForTrainingFeatures, :441 -> kpl_compute_features).  Then train a random forest on the
first 80 % of the rows (:496-499; label 0 = keypoint, 1 = not a keypoint, :405-407) and save it in the
OpenCV RTrees YAML layout libkpl / cv::ml::RTrees::load read (:512).

Layout of the inputs (the reference's, with '/' separators):
    <pathDataset>/<model>/<view><ext>                 the views          (.pcd or .ply, ascii / binary, x y z)
    <pathTrainingData>/<model>/positives/<view>.pcd   keypoints          (x y z [intensity])
    <pathTrainingData>/<model>/negatives/<view>.pcd   non-keypoints

The forest trainer is not the reference's (OpenCV is absent): scikit-learn's RandomForestClassifier
with the same knobs (ntrees, depth, min sample count, sqrt(F) active variables) or, with
--trainer extra, the numpy extremely-randomised-trees trainer of tools/synth.py.  Leaves carry the
majority class label as their value, which is what cv::ml::RTrees stores for a classifier and what
predict(PREDICT_SUM) adds up (/root/reference/include/impl/KeypointLearning.hpp:281-287).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools import cloud_io, forest_yaml, synth  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--pathDataset", required=True)
    ap.add_argument("--ext", default=".pcd")
    ap.add_argument("--pathTrainingData", required=True)
    ap.add_argument("--pathRF", required=True, help="output folder")
    ap.add_argument("--nameRF", default="forest.yaml.gz")
    ap.add_argument("--ntrees", type=int, default=100)
    ap.add_argument("--depth", type=int, default=25)
    ap.add_argument("--msc", type=int, default=1, help="min sample count of a node that may still be split")
    ap.add_argument("--annuli", type=int, default=5)
    ap.add_argument("--bins", type=int, default=10)
    ap.add_argument("--radiusFeatures", type=float, required=True)
    ap.add_argument("--nnNormals", type=int, default=10)
    ap.add_argument("--radiusNormals", type=float, default=0.0, help="> 0: normals on radius instead of k-search")
    ap.add_argument("--flipNormals", action="store_true")
    ap.add_argument("--subSampling", action="store_true")
    ap.add_argument("--leaf", type=float, default=0.0)
    ap.add_argument("--trainer", choices=("sklearn", "extra"), default="sklearn")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv)


def uniform_sampling(xyz, leaf):
    """pcl::UniformSampling: per leaf-sized voxel the point closest to the voxel centre (ascending index)."""
    v = np.floor(xyz.astype(np.float64) / leaf)
    centre = (v + 0.5) * leaf
    d = ((xyz - centre) ** 2).sum(axis=1)
    _, inv = np.unique(v.astype(np.int64), axis=0, return_inverse=True)
    order = np.lexsort((d, inv.ravel()))
    first = np.ones(len(order), dtype=bool)
    first[1:] = inv.ravel()[order][1:] != inv.ravel()[order][:-1]
    return np.sort(order[first])


def snap_to_cloud(xyz, query):
    """index of the nearest input point of every training point (kdtree.nearestKSearch(.., 1), :424-437)"""
    from scipy.spatial import cKDTree
    if len(query) == 0:
        return np.zeros(0, dtype=np.int32)
    return cKDTree(xyz.astype(np.float64)).query(query.astype(np.float64), k=1)[1].astype(np.int32)


def sklearn_to_arrays(clf, F):
    """Flatten the fitted estimators: split `x[var] <= thr -> left`, float32 thresholds that decide
    exactly like sklearn's float64 ones on float32 features, leaf value = majority class label."""
    root, var, thr, left, right, value, depth, cidx, qual = ([] for _ in range(9))
    classes = clf.classes_
    for est in clf.estimators_:
        t = est.tree_
        base = len(var)
        root.append(base)
        dep = np.zeros(t.node_count, dtype=np.int32)
        for nd in range(t.node_count):
            is_split = t.children_left[nd] >= 0
            th32 = np.float32(0)
            if is_split:
                th32 = np.float32(t.threshold[nd])
                if float(th32) > t.threshold[nd]:            # round DOWN: x <= thr64  <=>  x <= thr32 for float32 x
                    th32 = np.nextafter(th32, np.float32(-np.inf), dtype=np.float32)
                dep[t.children_left[nd]] = dep[t.children_right[nd]] = dep[nd] + 1
            label = int(classes[int(np.argmax(t.value[nd][0]))])
            var.append(int(t.feature[nd]) if is_split else -1)
            thr.append(th32)
            left.append(base + int(t.children_left[nd]) if is_split else -1)
            right.append(base + int(t.children_right[nd]) if is_split else -1)
            value.append(float(label))
            depth.append(int(dep[nd]))
            cidx.append(label)
            qual.append(float(t.impurity[nd] * t.n_node_samples[nd]) if is_split else 0.0)
    return forest_yaml.ForestArrays(root, var, thr, left, right, value, F, depth, cidx, qual)


VIEWS_PER_BATCH = 8          # kpl_compute_features_batch_device: the training points of up to 8 views in one launch


def batch_features(kpl, dets, pending, device):
    """computePointsForTrainingFeatures of up to 8 views at once (the caller loop of main_train_detector.cpp:413-446, which
    takes a few hundred points from each of many views): clouds, normals and indices to the device, ONE batched index build
    and ONE feature launch for all of them (kpl_compute_features_batch_device), the rows back.  Returns (rows per view, seconds
    on the device path)."""
    import torch
    dev = torch.device("cuda", device)
    t0 = time.perf_counter()
    keep, idx_p, out_p, ms, outs = [], [], [], [], []
    F = dets[0]._p.n_annulus * dets[0]._p.n_bins
    for det, (xyz, nrm, idx) in zip(dets, pending):
        dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(np.ascontiguousarray(nrm, dtype=np.float32)).to(dev)
        di = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(dev)
        do = torch.empty((len(idx), F), dtype=torch.float32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
        keep.append((dx, dn, di))
        idx_p.append(di.data_ptr()); out_p.append(do.data_ptr()); ms.append(len(idx)); outs.append(do)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    use = dets[:len(pending)]
    for attempt in range(6):                          # (a first view of a size grows the cell tables: KPL_ERR_RETRY)
        kpl.compute_features_batch_device(use, idx_p, ms, out_p, st)
        rcs = [d.syncStatus(st) for d in use]         # every handle, no short circuit
        if kpl.ERR_RETRY not in rcs:
            break
    else:
        raise SystemExit("the views keep asking for larger tables")
    rows = [o.cpu().numpy() for o in outs]
    return rows, time.perf_counter() - t0


def collect(args, log=print):
    kpl = importlib.import_module("keypoint-learning_amd")
    dets = []
    for _ in range(VIEWS_PER_BATCH):
        det = kpl.KeypointLearningDetector(device=args.device)
        det.setNAnnulus(args.annuli)
        det.setNBins(args.bins)
        det.setRadiusSearch(args.radiusFeatures)
        dets.append(det)
    det = dets[0]
    feats, labels, views = [], [], 0
    pending, names, t_feat = [], [], 0.0

    def flush():
        nonlocal t_feat
        if not pending:
            return
        rows, dt = batch_features(kpl, dets, pending, args.device)
        t_feat += dt
        for name, r in zip(names, rows):
            feats.append(r)
            log("[END] Compute features for: " + name)
        pending.clear()
        names.clear()
    for model in sorted(os.listdir(args.pathDataset)):
        mdir = os.path.join(args.pathDataset, model)
        if not os.path.isdir(mdir):
            continue
        for name in sorted(f for f in os.listdir(mdir) if f.endswith(args.ext)):
            stem = os.path.splitext(name)[0] + ".pcd"
            pos_p = os.path.join(args.pathTrainingData, model, "positives", stem)
            neg_p = os.path.join(args.pathTrainingData, model, "negatives", stem)
            if not os.path.exists(pos_p):
                log("Impossible to read positives cloud for: " + name)       # :318-322 (continue)
                continue
            if not os.path.exists(neg_p):
                raise SystemExit("Impossible to read negatives cloud for: " + name)   # :333-337 (exit)
            xyz = cloud_io.read_cloud_xyz(os.path.join(mdir, name))     # .pcd or .ply (:296-310)
            xyz = np.ascontiguousarray(xyz[np.isfinite(xyz).all(axis=1)])
            if args.subSampling:
                xyz = np.ascontiguousarray(xyz[uniform_sampling(xyz, args.leaf)])
            k = 0 if args.radiusNormals > 0 else args.nnNormals
            nrm, _ = det.estimateNormals(xyz, k=k, radius=args.radiusNormals)
            if args.flipNormals:
                nrm = -nrm
            pos, neg = cloud_io.read_pcd_xyz(pos_p), cloud_io.read_pcd_xyz(neg_p)
            idx = snap_to_cloud(xyz, np.concatenate([pos, neg]))
            lab = np.concatenate([np.zeros(len(pos), np.int32), np.ones(len(neg), np.int32)])   # :405-407
            log("[START] Compute features for: " + name)
            pending.append((xyz, nrm, idx))
            names.append(name)
            labels.append(lab)
            views += 1
            if len(pending) == VIEWS_PER_BATCH:
                flush()
    flush()
    if not feats:
        raise SystemExit("no view with a training set found")
    collect.feature_seconds = t_feat
    return np.concatenate(feats).astype(np.float32), np.concatenate(labels), views


def train(feat, lab, args):
    ntrain = int(len(feat) * 0.8)                                            # :496
    F = feat.shape[1]
    if args.trainer == "extra":
        fa = synth.train_extra_trees(feat[:ntrain], lab[:ntrain], ntrees=args.ntrees, max_depth=args.depth,
                                     min_samples=max(2, args.msc), seed=args.seed)
    else:
        from sklearn.ensemble import RandomForestClassifier
        clf = RandomForestClassifier(n_estimators=args.ntrees, max_depth=args.depth,
                                     min_samples_split=max(2, args.msc), max_features="sqrt", bootstrap=True,
                                     random_state=args.seed, n_jobs=1)
        clf.fit(feat[:ntrain], lab[:ntrain])
        fa = sklearn_to_arrays(clf, F)
    return fa, ntrain


def forest_votes(fa, feat):
    """fraction of trees voting 'not a keypoint' per row = what predict(PREDICT_SUM) / ntrees gives"""
    out = np.zeros(len(feat))
    for r in fa.root:
        nd = np.full(len(feat), int(r))
        active = fa.var[nd] >= 0
        while active.any():
            v, th = fa.var[nd[active]], fa.thr[nd[active]]
            go_left = feat[active, v] <= th
            nd[active] = np.where(go_left, fa.left[nd[active]], fa.right[nd[active]])
            active = fa.var[nd] >= 0
        out += fa.value[nd]
    return out / fa.ntrees


def main(argv=None):
    args = parse_args(argv)
    os.makedirs(args.pathRF, exist_ok=True)
    t0 = time.time()
    feat, lab, views = collect(args)
    npos, nneg = int((lab == 0).sum()), int((lab == 1).sum())
    print("Training data: %d positives and %d negatives." % (npos, nneg))
    print("Training labels: %d points." % len(lab))
    t1 = time.time()
    fa, ntrain = train(feat, lab, args)
    t_train = time.time() - t1
    pred = (forest_votes(fa, feat) >= 0.5).astype(np.int32)
    err_train = float((pred[:ntrain] != lab[:ntrain]).mean()) if ntrain else 0.0
    err_test = float((pred[ntrain:] != lab[ntrain:]).mean()) if ntrain < len(lab) else 0.0
    print("Forest trained in: %.3f minutes." % (t_train / 60.0))
    print("Training error: %.4f (held-out 20 %%: %.4f)" % (err_train, err_test))
    out = os.path.join(args.pathRF, args.nameRF)
    forest_yaml.save_forest(fa, out, max_depth=args.depth, min_sample_count=args.msc)
    with open(os.path.join(args.pathRF, "training_parameters.log"), "w") as f:     # :218-234, :509-510
        f.write("Name rf: %s\nNumber of trees: %d\nDepth of each tree: %d\nDataset: %s\nTraining data: %s\nMsc: %d\n"
                "Features annuli: %d\nFeatures bins: %d\nFeatures radius: %s\nForest path: %s\n"
                % (args.nameRF, args.ntrees, args.depth, args.pathDataset, args.pathTrainingData, args.msc,
                   args.annuli, args.bins, repr(args.radiusFeatures), args.pathRF))
        f.write("Trained with: %d positives and %d negatives.\nTrain duration in seconds: %s\n" % (npos, nneg, repr(t_train)))
    print(json.dumps({"forest": out, "views": views, "rows": int(len(lab)), "positives": npos, "negatives": nneg,
                      "ntrees": int(fa.ntrees), "nodes": int(fa.nnodes), "train_error": err_train,
                      "test_error": err_test, "seconds": round(time.time() - t0, 3),
                      # upload + batched index build + batched feature launch + rows back, 8 views per batch
                      "feature_extraction": {"views_per_batch": VIEWS_PER_BATCH, "seconds": round(collect.feature_seconds, 4),
                                             "views_per_s": round(views / max(collect.feature_seconds, 1e-9), 1)}}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
