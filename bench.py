#!/usr/bin/env python3
"""bench.py -- Mpoints/s scored (feature + forest + NMS) on 200k-point 2.5D views, MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it with
torch.distributed.run (one rank per GPU, RCCL).  One "step" = one pass of the hot path
(pcl::Keypoint::compute = index build + feature + forest + NMS + keypoint compaction) over one
BATCH of independent synthetic views that are already resident in HBM.  A single 200 k-point view
is only ~3 waves per SIMD of an MI355X, so the engine runs every kernel of the pipeline once per
batch of views (kpl_compute_batch_device; default 8 views per step, `--batch 1` = one view per
step) and the bench keeps two batches in flight on two HIP streams (`--groups`).  Views are
independent, so ranks never exchange data on the data path; with N > 1 every step ends with one
RCCL all-gather of the packed keypoint lists -- the only exchange the path has -- and scaling is
weak (every GPU gets its own batch).

Workload = BASELINE.json configs[1]: 200k-pt synthetic 2.5D views (tools/synth.py, seeds 1, 2, ...),
10-tree forest data/forests/synth200k_a5b6_t10.yaml.gz (stand-in for the missing SHOT forest),
annuli=5 bins=6 r_feat=6*mr r_nms=4*mr thr=0.85, draws_remove=false.

The timed region is repeated REPEATS times (each repetition = exactly K steps between barriers);
`value` / `ms_per_step` are those of the median repetition and every repetition is listed.

Rank 0 prints ONE JSON line.  The oracle (oracle/) is used here only (a) as the parity gate
before timing counts and (b) as the timed `cpu_baseline` -- never on the measured path.
"""
import argparse
import hashlib
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
REPEATS = 5         # timed repetitions of the K-step loop; the median one is reported
KERNEL_SOURCES = ("keypoint-learning_amd/csrc/kernels.hip", "keypoint-learning_amd/csrc/kernels.h",
                  "keypoint-learning_amd/csrc/exact_math.h", "keypoint-learning_amd/csrc/soft_pair.h")
A, B = 5, 6
HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
LDS_PEAK = 75.0e12  # B/s, MI355X_MICROARCH.md "LDS": ~75 TB/s aggregate for ds_read_b32 with every CU streaming (b64: ~150)
try:                # the metric's name is BASELINE.json's, verbatim
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except (OSError, ValueError, KeyError):
    METRIC = "Mpoints/sec scored (feature+forest+NMS), 200k-pt cloud, 1/2/4/8 MI355X"


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not the machine's
    core count (a container on a 256-core host is usually limited to a few of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 32))


def kernel_source_sha256():
    """Identity of the kernels being benchmarked: profiles/counters.json records the same hash of the
    sources its counters were collected on; a mismatch means the profile is stale and is not quoted."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def load_profile(views_per_launch):
    """(counters of the profiled kernels, reason why not) from profiles/counters.json -- only if it was
    taken on exactly these kernel sources and on the same launch shape."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None, "profiles/counters.json missing"
    if prof.get("source_sha256") != kernel_source_sha256():
        return None, "profiles/counters.json was taken on other kernel sources (stale)"
    if prof.get("views_per_launch") != views_per_launch:
        return None, "profiles/counters.json was taken with %s views per launch" % prof.get("views_per_launch")
    return prof, None


def iso_ms_val(t):
    return t["feature_ms"] / max(t["calls"], 1)


def test_detector_child():
    """One TestDetector process at the reference main's defaults -- one detector, one compute() -- on cheff001 as the reference
    ships it (data/point_cloud_test/cheff001.pcd is DATA ascii).  Run BEFORE this process touches the GPU: a child that starts
    while the parent holds sixteen handles and their streams takes 16 ms for its first compute() instead of 1.9 (its set-up thread
    is still creating streams when compute() is called)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "keypoint-learning_amd", "TestDetector")
    gold = os.path.join(ROOT, "tests", "golden", "cheff001.npz")
    forest = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")
    if not (os.path.exists(exe) and os.path.exists(gold)):
        return None
    xyz = np.load(gold)["xyz"]
    n = len(xyz)
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        cloud = os.path.join(tmp, "cheff001.pcd")
        with open(cloud, "w") as f:
            f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                    "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (n, n))
            f.write("\n".join("%.9g %.9g %.9g" % (p[0], p[1], p[2]) for p in xyz) + "\n")
        for extra in ([], ["--sortedSearch"]):
            res = subprocess.run([exe, "--pathCloud", cloud, "--pathRF", forest, "--json"] + extra, capture_output=True, text=True, timeout=300)
            if res.returncode == 0 and res.stdout.strip():
                j = json.loads(res.stdout.strip().splitlines()[-1])
                rows.append({"options": " ".join(extra) or "(none)", "first_compute_ms": round(j["compute_first_s"] * 1e3, 3),
                             "warm_compute_ms": round(j["compute_s"] * 1e3, 3), "prepare_ms": round(j["prepare_s"] * 1e3, 3),
                             "keypoints": j["keypoints"], "walk": j["walk"]})
    return rows


def reference_defaults(kpl, torch, dev, local_rank, child_rows=None):
    """TestDetector with no options: tests/golden/cheff001.npz (the reference's data file + k = 10 normals), the 50-variable
    fixture forest.  Device-resident compute() in both neighbor orders (keypoint lists against the committed fixture), the
    class's host-array call warm and on a FRESH handle, and -- the true one-shot figure -- the TestDetector binary as a child
    process."""
    import ctypes as C
    z = np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))
    forest = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")
    xyz, nrm = np.ascontiguousarray(z["xyz"], np.float32), np.ascontiguousarray(z["nrm"], np.float32)
    n = len(xyz)
    r, rn, thr = float(z["r_feat"]), float(z["r_nms"]), float(z["thr"])

    def make(sorted_search):
        d = kpl.KeypointLearningDetector(device=local_rank)
        d.setNAnnulus(5); d.setNBins(10); d.setNonMaxima(True); d.setNonMaximaDrawsRemove(False)
        d.setNonMaxRadius(rn); d.setPredictionThreshold(thr); d.setRadiusSearch(r); d.setSortedSearch(sorted_search)
        if not d.loadForest(forest):
            raise RuntimeError(d.lastError())
        return d
    out = {"what": "the reference main's defaults: cheff001 (%d points), radiusFeatures 20, radiusNMS 4, threshold 0.85, 5 x 10" % n}
    if child_rows is not None:
        out["test_detector_process"] = child_rows
    dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    for name, srt, want in (("canonical", False, z["kp_canonical"]), ("sorted", True, z["kp_sorted"])):
        d = make(srt)
        d.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        for _ in range(6):                 # (tables / key array / word list grow, the handle measures the neighborhood)
            d.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st.cuda_stream)
            st.synchronize()
            d.syncStatus(st.cuda_stream)
        d.enableTiming(True)
        t0 = time.perf_counter()
        for _ in range(20):
            d.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st.cuda_stream)
        st.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 20
        assert d.syncStatus(st.cuda_stream) == kpl.OK
        t = d.getTiming()
        k = int(dk[0].item())
        out[name] = {"compute_ms": round(ms, 4), "Mpoints_per_s": round(n / ms / 1e3, 2),
                     "phases_ms": {q: round(t[q] / max(t["calls"], 1), 4) for q in ("index_ms", "feature_ms", "forest_ms", "nms_ms")},
                     "launch": d.getLastLaunch(), "keypoints": k,
                     "keypoints_equal_fixture": bool(k == len(want) and np.array_equal(dk[1:1 + k].cpu().numpy(), want))}
    # the class's own call: host arrays in PCL's layouts, kpl_detect_keypoints (PCIe inclusive)
    pcl_xyz, pcl_nrm = np.zeros((n, 4), np.float32), np.zeros((n, 8), np.float32)
    pcl_xyz[:, :3], pcl_nrm[:, :3] = xyz, nrm
    h_kp, h_kps, h_cnt = np.empty(n, np.int32), np.empty(n, np.float32), C.c_int()

    def host_call(d):
        d._push()
        c0 = time.perf_counter()
        rc = d._lib.kpl_detect_keypoints(d._h, pcl_xyz.ctypes.data, 16, pcl_nrm.ctypes.data, 32, n, h_kp.ctypes.data,
                                         h_kps.ctypes.data, n, C.byref(h_cnt))
        ms = (time.perf_counter() - c0) * 1e3
        if rc != 0:
            raise RuntimeError(d.lastError())
        return ms
    fresh = make(False)
    fresh._lib.kpl_reserve(fresh._h, n, 16, 32)        # (what the class does in setInputCloud)
    time.sleep(0.05)                                   # (a caller reads its cloud here; the handle's set-up thread runs meanwhile)
    first_ms = host_call(fresh)
    warm = sorted(host_call(fresh) for _ in range(12))
    # (the first call of a fresh handle INSIDE this long-lived process is not reported: it meets whatever the earlier phases of the
    # bench left in the device's memory pool -- 1.7 or 16 ms from run to run; the one-shot figure is test_detector_process)
    del first_ms
    out["host_arrays"] = {"warm_call_ms": round(warm[len(warm) // 2], 4),
                          "warm_Mpoints_per_s": round(n / warm[len(warm) // 2] / 1e3, 2),
                          "keypoints_equal_fixture": bool(h_cnt.value == len(z["kp_canonical"]) and
                                                          np.array_equal(h_kp[:h_cnt.value], z["kp_canonical"]))}
    return out


def visible_devices():
    """GPUs this process would see, WITHOUT initialising HIP (torch.cuda.device_count() only counts)."""
    import torch
    return torch.cuda.device_count()


def self_launch(args):
    """`bench.py --gpus N` outside a torch.distributed job: start N ranks (one process per GPU) with the launcher the
    driver uses, as a child process; its stdout (rank 0's JSON line) and exit code are this process's."""
    import socket
    import subprocess
    ndev = visible_devices()
    if ndev < args.gpus and not args.share_devices:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) are visible; refusing to wrap ranks onto shared "
                         "devices (--share-devices allows it for tests)\n" % (args.gpus, ndev))
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs between processes on this driver
    return subprocess.run(cmd, env=env).returncode


def main():
    # the oracle's OpenMP workers must sleep, not spin, once the parity gate is done: spinning
    # workers would compete with the thread that enqueues the timed steps
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="independent views per step and GPU (1..8)")
    ap.add_argument("--groups", type=int, default=2,
                    help="batches in flight: step i runs on HIP stream i %% groups with its own handles and views, so "
                         "that the index build / NMS of one batch overlap the scoring launch of the other")
    ap.add_argument("--nx", type=int, default=500)
    ap.add_argument("--ny", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: skip the single-view and alone-on-GPU extras so that every launch of "
                         "the dominant kernel in the trace is a launch of the timed workload")
    ap.add_argument("--no-parity", action="store_true", help="timing experiments with ablated kernels only")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1: create the process group anyway, so that every step ends with the all-gather "
                         "of the keypoint lists (RCCL with --backend nccl) exactly as it does with N > 1")
    ap.add_argument("--repeats", type=int, default=REPEATS,
                    help="timed repetitions of the K-step loop (the median one is reported); profiling runs use 1")
    ap.add_argument("--share-devices", action="store_true",
                    help="tests on a box with fewer GPUs than ranks: ranks wrap onto the visible devices (local_rank %% "
                         "device_count) instead of the run being refused; the line then says devices_shared")
    args = ap.parse_args()
    repeats = max(1, args.repeats)

    # `python bench.py --gpus N` launched bare: this process only starts the N rank processes (one per GPU, the launcher
    # the driver itself uses) as a CHILD and passes its exit code on.  Nothing here has touched the GPU: counting devices
    # does not initialise HIP, and a process that has is never replaced by another program.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py --gpus %d inside a job of %d rank(s): the two must agree" % (args.gpus, world))
    # N > 1: every rank process onto the CPUs next to ITS GPU, before the first HIP call of the process (sysfs only)
    affinity = None
    if world > 1:
        spec = importlib.util.spec_from_file_location("kpl_dist_early", os.path.join(ROOT, "keypoint-learning_amd", "dist.py"))
        # (dist.py imports torch, not torch.cuda: importing torch does not initialise the GPU)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        affinity = mod.pin_to_gpu_numa(local_rank)

    # the one-shot figure of the drop-in (reference_defaults.test_detector_process): a CHILD process, before this one has a GPU context
    child_rows = None
    if world == 1 and rank == 0 and not args.lean and not args.force_dist:
        try:
            child_rows = test_detector_child()
        except Exception as e:
            child_rows = [{"error": repr(e)}]

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    ndev = torch.cuda.device_count()
    devices_shared = False
    if local_rank >= ndev:
        if not args.share_devices:
            raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible device(s), %d rank(s)); "
                             "--share-devices lets ranks share a device (tests only)" % (local_rank, ndev, world))
        local_rank %= ndev
    devices_shared = args.share_devices and world > ndev
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                      # --force-dist outside torch.distributed.run: a group of one
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kpl = importlib.import_module("keypoint-learning_amd")
    from tools import synth

    nb = max(1, min(args.batch, 8))
    ng = max(1, min(args.groups, 4))
    thr = float(np.float32(0.85))       # TestDetector parses the threshold as float

    # ---- synthetic views, resident in HBM before the timed region ----------------------------------
    # keypoint output per view = one packed buffer [count, idx_0, idx_1, ...]; the buffers of a rank's
    # batch are rows of ONE tensor, which with N > 1 is the RCCL all-gather payload as it stands
    gather_cap = min(32768, args.nx * args.ny)   # keypoints per view that travel (a 200 k view has ~22 k)
    views, dets, d_in, d_scores, d_cnt, d_kp = [], [], [], [], [], []
    nv = nb * ng                           # views resident on this GPU
    for k in range(nv):
        seed = 1 + rank * nv + k
        xyz, nrm = synth.make_cloud(args.nx, args.ny, seed=seed)
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1000 + seed)
        views.append((xyz, nrm))
    n = views[0][0].shape[0]
    assert n >= gather_cap
    d_packed = torch.zeros(nv, n + 1, dtype=torch.int32, device=dev)
    for k, (xyz, nrm) in enumerate(views):
        det = kpl.KeypointLearningDetector(device=local_rank)
        mr = det.cloudResolution(xyz)       # kpl_cloud_resolution (input preparation, not timed)
        det.setNAnnulus(A)
        det.setNBins(B)
        det.setNonMaxima(True)
        det.setNonMaxRadius(float(np.float32(4.0 * mr)))
        det.setNonMaximaDrawsRemove(False)
        det.setPredictionThreshold(thr)
        det.setRadiusSearch(float(np.float32(6.0 * mr)))
        if not det.loadForest(FOREST):
            raise SystemExit("cannot load forest: " + det.lastError())
        dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det)
        d_in.append((dx, dn))
        d_scores.append(torch.empty(n, dtype=torch.float32, device=dev))
        d_cnt.append(d_packed[k, 0:1])
        d_kp.append(d_packed[k, 1:])
        views[k] = (xyz, nrm, mr)
    # every group on a stream of its own, none on the null stream: event waits that involve the legacy default stream
    # serialise the groups (measured: 1 532 instead of 1 785 Mpoints/s with the per-step gather, profiles/r03_notes.md)
    tstreams = [torch.cuda.Stream() for _ in range(ng)]
    stream = tstreams[0].cuda_stream
    p_scores = [t.data_ptr() for t in d_scores]
    p_kp = [t.data_ptr() for t in d_kp]
    p_cnt = [t.data_ptr() for t in d_cnt]
    caps = [n] * nb
    step_no = [0]

    def run_group(g):
        sl = slice(g * nb, (g + 1) * nb)
        if nb == 1:
            dets[g].computeDevice(p_scores[g], p_kp[g], n, p_cnt[g], tstreams[g].cuda_stream)
        else:
            kpl.compute_batch_device(dets[sl], p_scores[sl], p_kp[sl], caps, p_cnt[sl], tstreams[g].cuda_stream)

    gather_done = [None] * ng            # N > 1: the gather of a group's previous batch (it reads the buffers the next one writes)

    def step():
        g = step_no[0] % ng
        step_no[0] += 1
        if gather_done[g] is not None:
            tstreams[g].wait_event(gather_done[g])
        run_group(g)
        return g

    # ---- parity gate (rank 0): every view's keypoint list + scores must equal the oracle's --------
    torch.cuda.synchronize()            # (the buffers above were filled on torch's default stream; the groups run on others)
    for g in range(ng):
        run_group(g)
    # first view of this size: cell tables may have to grow.  EVERY detector is synced (no short
    # circuit: a detector that is not synced keeps its small tables) until all of them report OK
    for attempt in range(4):
        torch.cuda.synchronize()        # (syncStatus(None) waits for the NULL stream only: the groups' streams first)
        rcs = [d.syncStatus(None) for d in dets]
        if kpl.ERR_RETRY not in rcs:
            break
        torch.cuda.synchronize()
        for g in range(ng):
            run_group(g)
    else:
        raise SystemExit("cell tables still growing after 4 attempts")
    torch.cuda.synchronize()
    parity = None
    cpu = None
    if rank == 0:
        from oracle import kplo
        from tools import forest_yaml
        fa = forest_yaml.load_forest(FOREST)
        of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
        ncores = usable_cores()
        same_scores, same_kp, n_kp_total = True, True, 0
        oracle_results = []
        for k, (xyz, nrm, mr) in enumerate(views):
            r_feat, r_nms = float(np.float32(6.0 * mr)), float(np.float32(4.0 * mr))
            o_scores, o_kp = kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=ncores)
            g_scores = d_scores[k].cpu().numpy()
            g_kp = d_kp[k][:int(d_cnt[k].item())].cpu().numpy()
            same_scores &= bool(np.array_equal(g_scores.view(np.uint32), o_scores.view(np.uint32)))
            same_kp &= bool(np.array_equal(g_kp, o_kp))
            n_kp_total += int(len(o_kp))
            oracle_results.append((o_scores, o_kp))
        parity = {"scores_bit_exact": same_scores, "keypoints_identical": same_kp, "views_checked": nv,
                  "n_keypoints": n_kp_total}
        if not (same_scores and same_kp) and not args.no_parity:
            raise SystemExit("PARITY FAILURE vs oracle: %s" % parity)

    # ---- multi-GPU: the one exchange step = gather the keypoint lists ------------------------------
    gathered = [None]
    if use_dist:
        kd = importlib.import_module("keypoint-learning_amd.dist")

        comm_stream = torch.cuda.Stream()
        comm_spans = []                  # (start, end) events on the comm stream around every gather: collective.ms_per_step

        def full_step():
            g = step()
            # the collective runs on a stream of its own, ordered after that batch and before the group's NEXT batch
            # (two steps later): it never holds up the scoring of the other group
            batch_done = torch.cuda.Event()
            batch_done.record(tstreams[g])
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(batch_done)
                c_start = torch.cuda.Event(enable_timing=True)
                c_start.record(comm_stream)
                send = d_packed[g * nb:(g + 1) * nb, :gather_cap + 1].contiguous().view(-1)
                # the one exchange step of the path: all-gather of the packed keypoint lists (RCCL)
                gathered[0] = kd.gather_keypoints(send if args.backend == "nccl" else send.cpu())
                gather_done[g] = torch.cuda.Event(enable_timing=True)
                gather_done[g].record(comm_stream)
                comm_spans.append((c_start, gather_done[g]))
    else:
        full_step = step

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed settle: let clocks ramp and the host thread pool of the parity gate go to sleep
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.5:
        step()                      # no collective in here: ranks run different iteration counts
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        full_step()
    barrier()
    dets[0].enableTiming(True)
    rep_s, rep_enq = [], []
    if use_dist:
        del comm_spans[:]                # (the warm-up's gathers are not part of the figure)
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full_step()
        t_enq = time.perf_counter()
        barrier()
        t1 = time.perf_counter()
        rep_s.append(t1 - t0)
        rep_enq.append(t_enq - t0)
    timing = dets[0].getTiming()
    dets[0].enableTiming(False)
    # N > 1 (or --force-dist): what the exchange step itself takes on its stream, from the events around every gather of the
    # timed repetitions (pack + all-gather; it runs beside the other group's scoring, so this is not added to a step)
    collective_ms = None
    if use_dist and comm_spans:
        torch.cuda.synchronize()
        collective_ms = float(np.mean([a.elapsed_time(b) for a, b in comm_spans]))
    # the same loop at the OTHER customary step count (the driver runs --steps 20 --warmup 5, the tables of earlier rounds quoted
    # 200-step runs): a short run pays the fill of the two-deep pipeline and the clock ramp once per K steps
    sensitivity = None
    if world == 1 and not use_dist and not args.lean:
        other = 200 if args.steps != 200 else 20
        runs = []
        for _ in range(3):
            barrier()
            s0 = time.perf_counter()
            for _ in range(other):
                full_step()
            barrier()
            runs.append((time.perf_counter() - s0) * 1e3 / other)
        sensitivity = {"steps": other, "ms_per_step": round(sorted(runs)[1], 5),
                       "Mpoints_per_s": round(n * nb / sorted(runs)[1] / 1e3, 1)}
    # a status raised DURING the timed loop (a table that had to grow, a failed scan) would otherwise go unseen: every
    # detector must report OK now (the barrier above has drained every stream), else the numbers are not those of the path
    rcs = [d.syncStatus(None) for d in dets]
    if any(rc != kpl.OK for rc in rcs):
        raise SystemExit("bench.py: a timed step failed on the device: statuses %s (%s)" %
                         (rcs, "; ".join(d.lastError() for d, rc in zip(dets, rcs) if rc != kpl.OK)))
    # ... and the outputs of the LAST timed steps are the oracle's too: the timed steps run with what the handles measured in
    # the gate above (walk, accept words per point), not with the first call's defaults the gate itself saw
    if parity is not None and not args.no_parity:
        again_scores, again_kp = True, True
        for k, (o_scores, o_kp) in enumerate(oracle_results):
            again_scores &= bool(np.array_equal(d_scores[k].cpu().numpy().view(np.uint32), o_scores.view(np.uint32)))
            again_kp &= bool(np.array_equal(d_kp[k][:int(d_cnt[k].item())].cpu().numpy(), o_kp))
        parity["after_the_timed_steps"] = {"scores_bit_exact": again_scores, "keypoints_identical": again_kp,
                                           "feature_stage": {k: v for k, v in dets[0].getTiming().items()
                                                             if k in ("walk", "lanes_per_point", "accept_words")}}
        if not (again_scores and again_kp):
            raise SystemExit("PARITY FAILURE vs oracle after the timed steps: %s" % parity)
    per_rank = None
    if use_dist:        # every repetition: the slowest rank counts; every rank's own times travel too
        tt = torch.tensor(rep_s, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [[float(x) for x in t.tolist()] for t in allt]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rep_s = [float(x) for x in tt.tolist()]
    order = sorted(range(repeats), key=lambda k: rep_s[k])
    med = order[repeats // 2]
    elapsed, enq_s = rep_s[med], rep_enq[med]
    if use_dist:
        assert dist.get_world_size() == world == max(1, args.gpus), "the collective does not span --gpus ranks"
        assert per_rank is not None and len(per_rank) == world, "one timing row per rank"
        lists = kd.unpack_keypoints(gathered[0].view(world * nb, gather_cap + 1))
        assert len(lists) == world * nb and all(len(x) > 0 for x in lists)
        g_last = (step_no[0] - 1) % ng
        mine = lists[rank * nb]
        assert np.array_equal(mine.numpy(), d_kp[g_last * nb][:len(mine)].cpu().numpy())

    # the dominant kernel alone on the GPU: the same batch, one batch in flight
    torch.cuda.synchronize()
    extras = {}
    t_iso = t_single = None
    single_ms = detect_only_ms = 0.0
    if not args.lean:
        dets[0].enableTiming(True)
        for _ in range(20):
            run_group(0)
            torch.cuda.synchronize()
        t_iso = dets[0].getTiming()
        dets[0].enableTiming(False)

        # single view, single stream (latency mode): compute() and detect-only (index prebuilt)
        reps = max(20, args.steps // 4)
        ts0 = time.perf_counter()
        for _ in range(reps):
            dets[0].computeDevice(p_scores[0], p_kp[0], n, p_cnt[0], stream)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts0) * 1e3 / reps
        dets[0].buildIndexDevice(stream)
        dets[0].enableTiming(True)
        td0 = time.perf_counter()
        for _ in range(reps):
            dets[0].detectDevice(p_scores[0], p_kp[0], n, p_cnt[0], stream)
        torch.cuda.synchronize()
        detect_only_ms = (time.perf_counter() - td0) * 1e3 / reps
        t_single = dets[0].getTiming()
        dets[0].enableTiming(False)

    # ---- the drop-in class's own path: host buffers in PCL's layouts (PointXYZ 16 B, Normal 32 B), pageable,
    # one kpl_detect_keypoints per compute().  PCIe inclusive; never part of `value`.
    if not args.lean and rank == 0:
        xyz0, nrm0, mr0 = views[0]
        pcl_xyz = np.zeros((n, 4), dtype=np.float32)
        pcl_nrm = np.zeros((n, 8), dtype=np.float32)
        pcl_xyz[:, :3], pcl_nrm[:, :3] = xyz0, nrm0
        dets[0].setInputCloud(pcl_xyz)
        dets[0].setNormals(pcl_nrm)
        import ctypes as C
        host_ms = {}
        h_kp, h_kps, h_sc = np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.float32)
        h_cnt = C.c_int()
        lib, hd = dets[0]._lib, dets[0]._h
        dets[0]._push()
        for name in ("keypoints_only", "with_all_scores"):      # the C-ABI calls themselves, as the C++ class makes them
            ts = []
            for k in range(14):
                c0 = time.perf_counter()
                if name == "keypoints_only":
                    rc = lib.kpl_detect_keypoints(hd, pcl_xyz.ctypes.data, 16, pcl_nrm.ctypes.data, 32, n, h_kp.ctypes.data,
                                                  h_kps.ctypes.data, n, C.byref(h_cnt))
                else:
                    rc = lib.kpl_detect(hd, pcl_xyz.ctypes.data, 16, pcl_nrm.ctypes.data, 32, n, h_sc.ctypes.data,
                                        h_kp.ctypes.data, n, C.byref(h_cnt))
                ts.append((time.perf_counter() - c0) * 1e3)
                assert rc == 0, dets[0].lastError()
            host_ms[name] = float(np.median(ts[3:]))
        same = bool(np.array_equal(h_kp[:h_cnt.value], d_kp[0][:int(d_cnt[0].item())].cpu().numpy()))
        # the same bytes as plain pageable copies, for scale
        tx, tn = torch.from_numpy(pcl_xyz), torch.from_numpy(pcl_nrm)
        dx0, dn0 = torch.empty_like(tx, device=dev), torch.empty_like(tn, device=dev)
        ts = []
        for k in range(8):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            dx0.copy_(tx)
            dn0.copy_(tn)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - c0) * 1e3)
        h2d_ms = float(np.median(ts[2:]))
        # the same view through PINNED staging buffers of the handle (kpl_host_staging, what the class does with
        # setHostStaging(true)): packed 12-byte records filled by the caller beforehand, DMA uploads inside compute()
        sx, sn = dets[0].hostStaging(n, 12, 12)
        c0 = time.perf_counter()
        sx[:, :3] = xyz0
        sn[:, :3] = nrm0
        fill_ms = (time.perf_counter() - c0) * 1e3
        ts = []
        for k in range(14):
            c0 = time.perf_counter()
            rc = lib.kpl_detect_keypoints_staged(hd, h_kp.ctypes.data, h_kps.ctypes.data, n, C.byref(h_cnt))
            ts.append((time.perf_counter() - c0) * 1e3)
            assert rc == 0, dets[0].lastError()
        staged_ms = float(np.median(ts[3:]))
        same_staged = bool(np.array_equal(h_kp[:h_cnt.value], d_kp[0][:int(d_cnt[0].item())].cpu().numpy()))
        extras["host_buffer_path"] = {
            "what": "KeypointLearningDetector.compute() on host arrays in PCL layouts (16-B points, 32-B normals, pageable), "
                    "%d points: upload + index + detect + keypoint list back" % n,
            "compute_ms": round(host_ms["keypoints_only"], 4), "Mpoints_per_s": round(n / host_ms["keypoints_only"] / 1e3, 1),
            "compute_ms_with_all_scores_back": round(host_ms["with_all_scores"], 4),
            "upload_bytes": int(pcl_xyz.nbytes + pcl_nrm.nbytes),
            "same_bytes_as_plain_pageable_copies_ms": round(h2d_ms, 4),
            "keypoints_identical_to_device_path": same,
            "pinned_staging": {"what": "kpl_host_staging + kpl_detect_keypoints_staged: packed 12-byte xyz / normals in pinned "
                                       "buffers of the handle (filled beforehand, like setHostStaging(true) does in the setters), "
                                       "DMA uploads, normals overlapped with the first index kernels",
                               "compute_ms": round(staged_ms, 4), "Mpoints_per_s": round(n / staged_ms / 1e3, 1),
                               "upload_bytes": int(2 * 12 * n), "fill_ms_outside_compute": round(fill_ms, 4),
                               "keypoints_identical_to_device_path": same_staged}}
        dets[0].bindCloudDevice(d_in[0][0].data_ptr(), 12, d_in[0][1].data_ptr(), 12, n)   # back to the resident view

        # ---- the same batch in SORTED-search mode (kpl_params.neighbor_order = KPL_NEIGHBORS_SORTED): neighbors in FLANN's
        # sorted order, the one order a PCL build can be compared with bit for bit (tests/test_gpu_sorted.py holds the parity)
        for d in dets[:nb]:
            d.setSortedSearch(True)
        for _ in range(3):       # (kpl_sync_status: the deferred status, growth of the key array on RETRY -- and the hints of
            run_group(0)         # the sorted mode, list capacity and all-large, which a handle takes from its earlier calls)
            torch.cuda.synchronize()
            rcs = [d.syncStatus(None) for d in dets[:nb]]
        assert all(rc == kpl.OK for rc in rcs), rcs
        dets[0].enableTiming(True)
        c0 = time.perf_counter()
        for _ in range(20):
            run_group(0)
        torch.cuda.synchronize()
        sorted_ms = (time.perf_counter() - c0) * 1e3 / 20
        rcs = [d.syncStatus(None) for d in dets[:nb]]
        assert all(rc == kpl.OK for rc in rcs), rcs
        t_sorted = dets[0].getTiming()
        dets[0].enableTiming(False)
        for d in dets[:nb]:
            d.setSortedSearch(False)
        run_group(0)
        torch.cuda.synchronize()
        extras["sorted_search_mode"] = {"what": "one batch of %d views per step, one batch in flight, neighbor_order = sorted" % nb,
                                        "ms_per_step": round(sorted_ms, 5), "Mpoints_per_s": round(n * nb / sorted_ms / 1e3, 1),
                                        "feature_kernel_ms": round(t_sorted["feature_ms"] / max(t_sorted["calls"], 1), 5),
                                        "canonical_feature_kernel_ms_same_conditions": round(iso_ms_val(t_iso), 5) if t_iso else None}

        # ---- configs[0]: one small view alone on the GPU (the parity anchor of config 1, tests/golden/cheff000.npz)
        gold = os.path.join(ROOT, "tests", "golden", "cheff000.npz")
        if os.path.exists(gold):
            z = np.load(gold)
            cx, cn = np.ascontiguousarray(z["xyz"], dtype=np.float32), np.ascontiguousarray(z["nrm"], dtype=np.float32)
            dsm = kpl.KeypointLearningDetector(device=local_rank)
            cmr = dsm.cloudResolution(cx)
            dsm.setNAnnulus(A); dsm.setNBins(B); dsm.setNonMaxima(True); dsm.setNonMaximaDrawsRemove(False)
            dsm.setNonMaxRadius(float(np.float32(4.0 * cmr))); dsm.setPredictionThreshold(thr)
            dsm.setRadiusSearch(float(np.float32(6.0 * cmr)))
            dsm.loadForest(FOREST)
            m = len(cx)
            tcx, tcn = torch.from_numpy(cx).to(dev), torch.from_numpy(cn).to(dev)
            sc1 = torch.empty(m, dtype=torch.float32, device=dev)
            kp1 = torch.zeros(m + 1, dtype=torch.int32, device=dev)
            dsm.bindCloudDevice(tcx.data_ptr(), 12, tcn.data_ptr(), 12, m)
            dsm.computeDevice(sc1.data_ptr(), kp1[1:].data_ptr(), m, kp1[0:1].data_ptr(), stream)
            while dsm.syncStatus(stream) == kpl.ERR_RETRY:
                dsm.computeDevice(sc1.data_ptr(), kp1[1:].data_ptr(), m, kp1[0:1].data_ptr(), stream)
            dsm.enableTiming(True)
            reps1 = 200
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for _ in range(reps1):
                dsm.computeDevice(sc1.data_ptr(), kp1[1:].data_ptr(), m, kp1[0:1].data_ptr(), stream)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - c0) * 1e3 / reps1
            t1v = dsm.getTiming()
            extras["single_view_cfg1"] = {"view": "tests/golden/cheff000.npz", "points": m, "compute_ms": round(ms1, 5),
                                          "Mpoints_per_s": round(m / ms1 / 1e3, 2), "keypoints": int(kp1[0].item()),
                                          "phases_ms": {k: round(t1v[k] / max(t1v["calls"], 1), 5)
                                                        for k in ("index_ms", "feature_ms", "forest_ms", "nms_ms")}}

    # ---- the reference's OWN default operating point (src/main_test_detector.cpp:62-67,105-106: cheff001, radiusFeatures 20,
    # radiusNMS 4, threshold 0.85, 5 x 10), both neighbor orders, and what the drop-in's ONE compute() costs
    if not args.lean and rank == 0 and world == 1:
        try:
            extras["reference_defaults"] = reference_defaults(kpl, torch, dev, local_rank, child_rows)
        except Exception as e:               # (a missing fixture must not cost the headline)
            extras["reference_defaults"] = {"error": repr(e)}

    # ---- algorithmic bytes from the engine's own counters -------------------------
    # B_alg(i) = 24 (1 + K_f) + 16 K_n [s >= thr] + 8 sum depth + 8.  The scoring stage is two kernels:
    # the feature kernel (the dominant one) gathers xyz + normal of the point and of each feature
    # neighbor, 24 (1 + K_f) bytes; the forest kernel visits 8-byte nodes and writes the 4-byte score.
    b_alg_total = b_alg_feat = b_alg_forest = 0
    stats = []
    for d in dets[:nb]:                     # one batch = what one launch of the dominant kernel covers
        st = d.collectStats(stream)
        stats.append(st)
        b_alg_total += 24 * (st["n_scored"] + st["sum_kf"]) + 16 * st["sum_kn"] + 8 * st["sum_depth"] + 8 * st["n_scored"]
        b_alg_feat += 24 * (st["n_scored"] + st["sum_kf"])
        b_alg_forest += 8 * st["sum_depth"] + 4 * st["n_scored"]
    calls = max(timing["calls"], 1)
    # HIP events on the launching stream around the launch, over the timed region.  With two batches in flight that span is the
    # kernel's duration PLUS the time its dispatch waits for the other batch's kernels (round 6: the small kernels of a batch are
    # one-wave workgroups that now run inside this span instead of queueing behind it: span 0.59 -> 0.78 ms while a step got
    # SHORTER, 0.825 -> 0.80 ms, and rocprofv3's begin-to-end duration of the dispatch moved 0.574 -> 0.597); events given to
    # hipExtLaunchKernelGGL measure the same span (tried).  frac_rocprof / alone_on_gpu below are the kernel by itself.
    feat_ms = timing["feature_ms"] / calls
    forest_ms = timing["forest_ms"] / calls
    achieved = b_alg_feat / (feat_ms * 1e-3) if feat_ms > 0 else 0.0
    iso_ms = max(t_iso["feature_ms"] / max(t_iso["calls"], 1), 1e-9) if t_iso else None
    # measured counters of exactly these kernels (rocprofv3, profiles/counters.json) -- or null
    prof, why_not = load_profile(nb)
    traffic = valu_busy = ta_busy = hbm_counter_frac = waves_per_simd = valu_issue_frac = valu_model = lane_frac = None
    forest_prof = None
    rocprof_avg_ns = rocprof_tag = None
    if prof:
        pk = prof["kernels"].get("feature_kernel", {})
        traffic = pk.get("hbm_bytes")
        rocprof_avg_ns, rocprof_tag = pk.get("rocprof_avg_ns"), prof.get("tag")
        valu_busy = pk.get("valu_busy")
        ta_busy = pk.get("ta_busy")
        waves_per_simd = pk.get("waves_per_simd")
        valu_issue_frac = pk.get("valu_issue_frac")
        valu_model = pk.get("valu_model")
        lane_frac = pk.get("valu_active_lane_frac")
        forest_prof = prof["kernels"].get("forest_kernel")
        if traffic and feat_ms > 0:
            hbm_counter_frac = round(traffic / (feat_ms * 1e-3) / HBM_PEAK, 5)
    # what limits the dominant kernel, decided by the counters (never by opinion): VALU issue if the instructions
    # it executes need >= 70 % of its cycles at the MEASURED issue ceilings of this chip (tools/valu_ceiling.hip ->
    # tools/valu_model.py), the texture path if that is busier, else latency
    if valu_issue_frac is None:
        bound = "unknown (no matching counter profile)"
    elif valu_issue_frac >= 0.70 and valu_issue_frac >= (ta_busy or 0.0):
        bound = "valu"
    elif (ta_busy or 0.0) >= 0.70:
        bound = "texture-path"
    else:
        bound = "latency"
    # the forest kernel (nodes in LDS): LDS pipe, VALU issue, or the latency of its dependent LDS round trips
    forest_bound = "unknown (no matching counter profile)"
    if forest_prof and forest_prof.get("valu_issue_frac") is not None:
        fv, fl = forest_prof["valu_issue_frac"], forest_prof.get("lds_busy") or 0.0
        forest_bound = "lds" if fl >= 0.70 and fl >= fv else "valu" if fv >= 0.70 else "latency (dependent LDS reads)"

    # ---- CPU baseline: the oracle, timed on the host cores, bounded sample --------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        xyz, nrm, mr = views[0]
        r_feat, r_nms = float(np.float32(6.0 * mr)), float(np.float32(4.0 * mr))
        reps_c, t_cpu = 0, 0.0
        while t_cpu < 12.0 and reps_c < 32:        # ~12 s of single-thread work (the contract asks for 10-30 s)
            c0 = time.perf_counter()
            kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=1)
            t_cpu += time.perf_counter() - c0
            reps_c += 1
        c0 = time.perf_counter()
        kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=ncores)
        t_all = time.perf_counter() - c0
        cpu = {"value": round(n * reps_c / t_cpu / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": "port",
               "sample": "%d full passes of one %d-pt view of the batch (grid build + feature + forest + NMS), "
                         "oracle/kpl_oracle.c -O2 -ffp-contract=off, uniform grid not FLANN" % (reps_c, n),
               "all_cores": {"value": round(n / t_all / 1e6, 4), "cores": ncores}}

    if rank == 0:
        ms = elapsed * 1e3 / args.steps
        per = lambda t: {k: round(t[k] / max(t["calls"], 1), 5) for k in ("index_ms", "feature_ms", "forest_ms", "nms_ms")}
        out = {
            "metric": METRIC,
            "value": round(n * nb * world * args.steps / elapsed / 1e6, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1] (single %d-pt synthetic 2.5D view, 10-tree forest), annuli=5 bins=6 "
                                   "r_feat=6*mr r_nms=4*mr thr=0.85; one step = a batch of %d independent views "
                                   "per GPU (scored in one launch); %d batches in flight on %d HIP streams"
                                   % (n, nb, ng, ng),
                       "points_per_view": n, "views_per_step_per_gpu": nb, "views_per_launch": nb,
                       "batches_in_flight": ng,
                       "mr": [round(v[2], 6) for v in views[:nb]],
                       "forest": os.path.basename(FOREST), "timed": "index build + detect (compute()) of every view",
                       "parallelism": "views sharded, %d rank(s)" % world,
                       "exchange": ("one all-gather of the packed keypoint lists per step (%s)" %
                                    ("RCCL" if args.backend == "nccl" else args.backend)) if use_dist else "none (one rank)"},
            "repeats": {"n": repeats, "reported": "median", "ms_per_step": [round(x * 1e3 / args.steps, 5) for x in rep_s],
                        "steps_sensitivity": sensitivity},
            # N > 1: the median repetition's ms per step of EVERY rank (the reported one is their maximum), the size of the
            # RCCL communicator the gathers ran in, and where rank 0's process was pinned before its first HIP call
            "per_rank_ms_per_step": [round(r[med] * 1e3 / args.steps, 5) for r in per_rank] if per_rank else None,
            "collective": {"backend": "RCCL" if args.backend == "nccl" else args.backend, "world_size": dist.get_world_size(),
                           "ms_per_step": round(collective_ms, 5) if collective_ms is not None else None,
                           "what": "pack + all-gather of the keypoint lists, HIP events on the comm stream (it runs beside the "
                                   "other batch's scoring: not a part of ms_per_step unless it is longer than a step)",
                           "bytes_per_rank_per_step": int(nb * (gather_cap + 1) * 4)}
            if use_dist else None,
            "devices_shared": devices_shared,       # true only in tests that run more ranks than the box has GPUs
            "cpu_affinity": affinity,
            # SURVEY 8(d) contract figure: gather-model bytes of the dominant kernel / its launch time / 8 TB/s (most of
            # those bytes are L1 / L2 hits: the HBM traffic by counters is hbm_counter_frac of peak).  `bound` is set from
            # counters of exactly these kernel sources (profiles/counters.json, matched by hash; null when stale):
            # valu_issue_frac = the SIMD cycles the kernel's instructions need at the issue ceilings measured on this
            # chip / the kernel's cycles; ta_busy = the texture path; valu_busy = the old 4-cycles-per-instruction
            # upper bound, kept for comparison with earlier rounds.
            "roofline": {"bound": bound, "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic,
                         "kernel": "feature_kernel (histogram features, %d view(s) per launch)" % nb,
                         "kernel_ms": round(feat_ms, 5), "alg_bytes_per_launch": int(b_alg_feat),
                         "kernel_ms_is": "HIP events around the launch on its stream: the dispatch's duration + its wait behind the other batch",
                         # measured IN THIS RUN: achieved / frac / kernel_ms (HIP events on the launching stream) and
                         # alg_bytes_per_launch (the engine's own counters).  NOT measured in this run: everything that needs
                         # rocprofv3 -- traffic, valu_*, ta_busy, waves_per_simd, hbm_counter_frac's numerator and the forest
                         # kernel's counters -- replayed from profiles/counters.json, quoted only while its source hash matches.
                         # frac_rocprof = the same algorithmic bytes / the kernel's AVERAGE duration in the committed
                         # rocprofv3 --kernel-trace --stats summary (profiles/<tag>_kernel_stats.csv) / peak
                         "frac_rocprof": round(b_alg_feat / (rocprof_avg_ns * 1e-9) / HBM_PEAK, 5) if rocprof_avg_ns else None,
                         "rocprof_avg_ms": round(rocprof_avg_ns * 1e-6, 5) if rocprof_avg_ns else None,
                         "rocprof_summary": ("profiles/%s_kernel_stats.csv" % rocprof_tag) if rocprof_avg_ns else None,
                         "valu_issue_frac": valu_issue_frac,
                         "valu_issue_model": {k: valu_model[k] for k in ("valu_issue_frac_bounds", "class_cycles_per_instruction",
                                                                         "avg_cycles_per_instruction_at_ceiling", "static_fast_share")}
                         if valu_model else None,
                         "valu_busy": valu_busy, "valu_active_lane_frac": lane_frac, "ta_busy": ta_busy, "hbm_counter_frac": hbm_counter_frac,
                         "waves_per_simd": waves_per_simd,
                         "counters": {"file": "profiles/counters.json", "source_sha256": kernel_source_sha256()[:16],
                                      "matches_these_kernels": prof is not None, "note": why_not,
                                      "measured_in_this_run": False},
                         "alone_on_gpu": {"kernel_ms": round(iso_ms, 5),
                                          "frac": round(b_alg_feat / (iso_ms * 1e-3) / HBM_PEAK, 5)} if iso_ms else None,
                         # the forest kernel walks nodes that it has staged in LDS: its node bytes never come from HBM, so
                         # no fraction of the HBM peak is quoted for them.  LDS side: bytes the walk reads from LDS (8-byte node
                         # + 4-byte feature per visited node) against the guide's aggregate ds_read rate (~75 TB/s for b32,
                         # ~150 TB/s for b64 with every CU streaming); lds_busy / bank conflicts from counters.
                         "forest_kernel": {"kernel_ms": round(forest_ms, 5), "bound": forest_bound,
                                           "lds_bytes_per_launch": int(12 * sum(st["sum_depth"] for st in stats)),
                                           "lds_TBps": round(12 * sum(st["sum_depth"] for st in stats) / (forest_ms * 1e-3) / 1e12, 2) if forest_ms > 0 else None,
                                           "lds_peak_TBps": LDS_PEAK / 1e12,
                                           "lds_frac": round(12 * sum(st["sum_depth"] for st in stats) / (forest_ms * 1e-3) / LDS_PEAK, 4) if forest_ms > 0 else None,
                                           "hbm_bytes_by_counters": forest_prof.get("hbm_bytes") if forest_prof else None,
                                           "counters": forest_prof}},
            "cpu_baseline": cpu,
            "phases_ms": per(timing),
            "single_view": {"compute_ms": round(single_ms, 5), "Mpoints_per_s": round(n / single_ms / 1e3, 2),
                            "detect_only_ms": round(detect_only_ms, 5), "phases_ms": per(t_single)} if t_single else None,
            "host_enqueue_ms_per_step": round(enq_s * 1e3 / args.steps, 5),
            "alg_bytes_per_point": round(b_alg_total / max(sum(s["n_scored"] for s in stats), 1), 1),
            "pipeline_alg_GBps": round(b_alg_total / (ms * 1e-3) / 1e9, 2),
            "counters": stats[0],
            "parity": parity,
        }
        out.update(extras)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
