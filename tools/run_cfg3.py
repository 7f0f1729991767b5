#!/usr/bin/env python3
"""BASELINE.json configs[2]: a batch of 64 Laser-Scanner-sized 2.5D views sharded over the GPUs of one
node, keypoint lists gathered with one RCCL all-gather per round.

    # 8 GPUs (the configuration as written), one command:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/run_cfg3.py
    # 1 GPU (all 64 views on it, 8 batches of 8):  python tools/run_cfg3.py
    # plumbing check without RCCL, 2 ranks on one device:  ... --nproc-per-node 2 tools/run_cfg3.py --backend gloo --views 8 --nx 80 --ny 60

Views = seeded synthetic clouds (tools/synth.py, seeds 100..163, 252 x 250 points ~ 63 k like the cheff
scans), forest data/forests/synth200k_a5b6_t10.yaml.gz, annuli=5 bins=6 r_feat=6*mr r_nms=4*mr thr=0.85.
Rank r takes the views dist.shard(64, world, r) (round robin), scores them in batches of up to 8 views per
launch (kpl_compute_batch_device), two batches in flight, and after each round every rank contributes its
packed keypoint lists to ONE all-gather.  Metric = all points / makespan (max over ranks), rank 0 prints one
JSON line; rank 0 also checks its own views against the oracle before timing (parity gate).
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
FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--nx", type=int, default=252)
    ap.add_argument("--ny", type=int, default=250)
    ap.add_argument("--rounds", type=int, default=20, help="timed repetitions of the whole 64-view job")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--write-pcd", metavar="DIR", default=None,
                    help="only write the views as binary .pcd files (x y z normal_x normal_y normal_z) into DIR and exit: the "
                         "input of the C++ runner, keypoint-learning_amd/DetectViews --devices all DIR/view*.pcd")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-all", action="store_true", help="rank 0 checks ALL its views against the oracle, not the first 4")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1: create the process group anyway, so that the all-gather (RCCL) runs on one GPU")
    args = ap.parse_args()
    if args.write_pcd:
        from tools import synth
        os.makedirs(args.write_pcd, exist_ok=True)
        for v in range(args.views):
            xyz, nrm = synth.make_cloud(args.nx, args.ny, seed=100 + v)
            xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1100 + v)
            arr = np.concatenate([xyz, nrm], axis=1).astype(np.float32)
            with open(os.path.join(args.write_pcd, "view%02d.pcd" % v), "wb") as f:
                f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\n"
                         "SIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                         "POINTS %d\nDATA binary\n" % (len(arr), len(arr))).encode())
                f.write(arr.tobytes())
        print(json.dumps({"wrote": args.views, "dir": args.write_pcd, "points_per_view": args.nx * args.ny}))
        return
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device (no CPU fallback)")
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    kpl = importlib.import_module("keypoint-learning_amd")
    kd = importlib.import_module("keypoint-learning_amd.dist")
    from tools import synth
    A, B, thr = 5, 6, float(np.float32(0.85))
    mine = kd.shard(args.views, world, rank)
    n = args.nx * args.ny
    cap = min(32768, n)                      # keypoints per view that travel (the 63 k views have 7-23 k; a strict unpack refuses a cut list)
    dets, bufs, host = [], [], []
    packed = torch.zeros(len(mine), n + 1, dtype=torch.int32, device=dev)
    for k, v in enumerate(mine):
        xyz, nrm = synth.make_cloud(args.nx, args.ny, seed=100 + v)
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1100 + v)
        det = kpl.KeypointLearningDetector(device=local)
        mr = det.cloudResolution(xyz)
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaximaDrawsRemove(False)
        det.setNonMaxRadius(float(np.float32(4 * mr))); det.setPredictionThreshold(thr)
        det.setRadiusSearch(float(np.float32(6 * mr)))
        assert det.loadForest(FOREST), det.lastError()
        dx, dn = torch.from_numpy(np.array(xyz)).to(dev), torch.from_numpy(np.array(nrm)).to(dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det); bufs.append((dx, dn, torch.empty(n, dtype=torch.float32, device=dev)))
        host.append((xyz, nrm, mr))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    batches = [list(range(i, min(i + 8, len(mine)))) for i in range(0, len(mine), 8)]
    gathered = [None]

    def job():
        for j, idx in enumerate(batches):
            st = streams[j % 2]
            kpl.compute_batch_device([dets[i] for i in idx], [bufs[i][2].data_ptr() for i in idx],
                                     [packed[i, 1:].data_ptr() for i in idx], [n] * len(idx),
                                     [packed[i, 0:1].data_ptr() for i in idx], st.cuda_stream)
        torch.cuda.synchronize()
        if use_dist:                         # the one exchange step: every rank's packed lists, one all-gather
            send = packed[:, :cap + 1].contiguous().view(-1)
            gathered[0] = kd.gather_keypoints(send if args.backend == "nccl" else send.cpu())

    torch.cuda.synchronize()            # (the buffers above were filled on the default stream; the batches run on others)
    job()
    while kpl.ERR_RETRY in [d.syncStatus(None) for d in dets]:
        job()
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import kplo
        from tests import helpers
        from tools import forest_yaml
        of = helpers.oracle_forest(forest_yaml.load_forest(FOREST))
        ok = True
        for i, (xyz, nrm, mr) in enumerate(host if args.parity_all else host[:4]):
            o_sc, o_kp = kplo.detect(xyz, nrm, A, B, float(np.float32(6 * mr)), float(np.float32(4 * mr)), thr, of,
                                     threads=helpers.usable_cores())
            ok &= bool(np.array_equal(bufs[i][2].cpu().numpy().view(np.uint32), o_sc.view(np.uint32)))
            ok &= bool(np.array_equal(packed[i, 1:1 + int(packed[i, 0])].cpu().numpy(), o_kp))
        parity = ok
        assert ok, "PARITY FAILURE vs oracle"
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.rounds):
        job()
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        lists = kd.unpack_keypoints(gathered[0].view(world * len(mine), cap + 1))
        own = lists[rank * len(mine)]
        assert np.array_equal(own.numpy(), packed[0, 1:1 + len(own)].cpu().numpy())
    if rank == 0:
        print(json.dumps({"config": "configs[2]: %d views x %d points sharded over %d GPU(s)" % (args.views, n, world),
                          "n_gpus": world, "views": args.views, "points_per_view": n, "rounds": args.rounds,
                          "makespan_ms_per_job": round(el * 1e3 / args.rounds, 4),
                          "Mpoints_per_s": round(args.views * n * args.rounds / el / 1e6, 2),
                          "views_per_rank": len(mine), "exchange": "one all-gather of %d x %d int32 per job" %
                          (world * len(mine), cap + 1) if use_dist else "none", "backend": args.backend if use_dist else None,
                          "parity_first_4_views_rank0": parity,
                          "views_checked_against_oracle": (len(host) if args.parity_all else min(4, len(host))) if parity is not None else 0}))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
