"""Builds libkpl.so (the C-ABI of include/kpl.h) for gfx950 with hipcc, in tree.

`python keypoint-learning_amd/build.py` or `build()` from __graft_entry__.  hipcc cross-compiles
without a GPU.  -ffp-contract=off is part of the specification of the kernels (bit-exact float
arithmetic); do not add -ffast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkpl.so")
SOURCES = ["kernels.hip", "organized_normals.hip", "api.cpp", "forest.cpp"]
HEADERS = ["kernels.h", "exact_math.h", "soft_pair.h", "organized_normals.h", "forest.h", os.path.join("..", "..", "include", "kpl.h"),
           os.path.join("..", "..", "include", "kpl_debug.h")]
TOOLS = {"TestDetector": ["test_detector_main.cpp"], "DetectViews": ["batch_views_main.cpp"]}

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-x", "hip"]


def source_hash():
    """sha256 over the sources and headers libkpl.so is built from: compiled into the library (kpl_source_hash) so that a
    binary can be told apart from one built from other sources (tests/test_abi.py)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    sha = source_hash()
    stamp = os.path.join(CSRC, ".build_hash")
    built_from = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if built_from != sha:
        force = True                       # (mtimes lie after a checkout; the hash does not)
    if force or _stale(LIB, deps):
        objs = []
        for s in srcs:
            o = os.path.splitext(s)[0] + ".o"
            if force or _stale(o, deps):
                cmd = [HIPCC] + FLAGS + ['-DKPL_SOURCE_SHA="%s"' % sha, "-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
            objs.append(o)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-lz", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(sha + "\n")
    # the same library with the test hooks of include/kpl_debug.h (api.cpp compiled with -DKPL_TEST_HOOKS): what
    # tests/test_gpu_status.py loads in child processes to force the failure paths; the shipped libkpl.so has no such symbol
    hooks = os.path.join(HERE, "..", "tests", "csrc", "libkpl_testhooks.so")
    if force or _stale(hooks, deps + [LIB]):
        ho = os.path.join(CSRC, "api_testhooks.o")
        cmd = [HIPCC] + FLAGS + ['-DKPL_SOURCE_SHA="%s"' % sha, "-DKPL_TEST_HOOKS", "-c", os.path.join(CSRC, "api.cpp"), "-o", ho]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs = [os.path.splitext(x)[0] + ".o" for x in srcs if not x.endswith("api.cpp")] + [ho]
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", hooks] + objs + ["-lz", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    for name, tsrcs in TOOLS.items():
        tpaths = [os.path.join(CSRC, s) for s in tsrcs]
        if not all(os.path.exists(p) for p in tpaths):
            continue
        exe = os.path.join(HERE, name)
        if force or _stale(exe, tpaths + [LIB, os.path.join(HERE, "..", "include", "KeypointLearning.h"), os.path.join(CSRC, "pcd_io.h")]):
            # plain g++: the tools are ordinary host programs (DetectViews also uses the HIP runtime API for its
            # own device buffers)
            cmd = ["g++", "-O2", "-std=c++14", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HERE, "..", "include"),
                   "-I", "/opt/rocm/include"] + tpaths + \
                  ["-o", exe, "-L", HERE, "-lkpl", "-L", "/opt/rocm/lib", "-lamdhip64"] + \
                  (["-lrccl", "-pthread"] if name == "DetectViews" else []) + \
                  ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    # test programs in C++ (tests/csrc): the hammer that scores one saved case again and again under co-running load
    # (tools/soak.sh, profiles/r04_notes.md) and the probe of the runtime's pageable-copy path
    tdir = os.path.join(HERE, "..", "tests", "csrc")
    hsrc, hexe = os.path.join(tdir, "hammer_case.cpp"), os.path.join(tdir, "hammer_case")
    if os.path.exists(hsrc) and (force or _stale(hexe, [hsrc, LIB, os.path.join(HERE, "..", "include", "kpl.h")])):
        cmd = ["g++", "-O2", "-std=c++14", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HERE, "..", "include"), "-I", "/opt/rocm/include",
               hsrc, "-o", hexe, "-L", HERE, "-lkpl", "-L", "/opt/rocm/lib", "-lamdhip64", "-pthread",
               "-Wl,-rpath,$ORIGIN/../../keypoint-learning_amd", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    msrc, mexe = os.path.join(tdir, "memset_probe.cpp"), os.path.join(tdir, "memset_probe")
    if os.path.exists(msrc) and (force or _stale(mexe, [msrc])):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-o", mexe, msrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    psrc, pexe = os.path.join(tdir, "pageable_copy_probe.cpp"), os.path.join(tdir, "pageable_copy_probe")
    if os.path.exists(psrc) and (force or _stale(pexe, [psrc])):
        cmd = [HIPCC, "-O2", "-o", pexe, psrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # stand-alone HIP diagnostics (not part of libkpl): the VALU issue-ceiling microbenchmark behind bench.py's
    # `valu_issue_frac` (tools/valu_ceiling.hip -> profiles/*_valu_ceiling.json)
    vsrc = os.path.join(HERE, "..", "tools", "valu_ceiling.hip")
    vexe = os.path.join(HERE, "..", "tools", "valu_ceiling")
    if os.path.exists(vsrc) and (force or _stale(vexe, [vsrc])):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-o", vexe, vsrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # exhaustive device-side check of exact_math.h's sqrt_rn against sqrtf (tests/test_gpu_exact_math.py)
    csrc_ = os.path.join(HERE, "..", "tools", "check_exact_math.hip")
    cexe = os.path.join(HERE, "..", "tools", "check_exact_math")
    if os.path.exists(csrc_) and (force or _stale(cexe, [csrc_, os.path.join(CSRC, "exact_math.h")])):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", CSRC, "-o", cexe, csrc_]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the soft assignment of the kernels (csrc/soft_pair.h) on the device over the reference-generated table
    # (tests/test_gpu_soft_pair.py)
    ssrc = os.path.join(HERE, "..", "tools", "check_soft_pair.hip")
    sexe = os.path.join(HERE, "..", "tools", "check_soft_pair")
    if os.path.exists(ssrc) and (force or _stale(sexe, [ssrc, os.path.join(CSRC, "exact_math.h"), os.path.join(CSRC, "soft_pair.h")])):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-I", CSRC, "-o", sexe, ssrc]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the reference-regeneration kit's driver built against THIS repo's drop-in header (tools/refgen/refgen_driver.cpp,
    # -DREFGEN_WITH_KPL): the kit's end-to-end self-test on the GPU box (tests/test_gpu_refgen.py)
    rsrc = os.path.join(HERE, "..", "tools", "refgen", "refgen_driver.cpp")
    rexe = os.path.join(HERE, "..", "tools", "refgen", "refgen_driver_kpl")
    if os.path.exists(rsrc) and (force or _stale(rexe, [rsrc, LIB, os.path.join(HERE, "..", "include", "KeypointLearning.h"),
                                                        os.path.join(HERE, "..", "include", "kpl_pcl_shim.h")])):
        cmd = ["g++", "-O2", "-std=c++14", "-DREFGEN_WITH_KPL", "-I", os.path.join(HERE, "..", "include"), rsrc, "-o", rexe,
               "-L", HERE, "-lkpl", "-Wl,-rpath,$ORIGIN/../../keypoint-learning_amd", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
