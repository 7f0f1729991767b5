"""Prints the VALU instructions of the innermost loop of a kernel that contains a given instruction
(build/exp/kernels.s is the device ISA of csrc/kernels.hip, see tools/valu_model.py for how it is made).

    python tools/loop_isa.py <mangled-name-fragment> <instruction-regex> [ways]
"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "build", "exp", "kernels.s")


def make():
    os.makedirs(os.path.dirname(S), exist_ok=True)
    c = os.path.join(ROOT, "keypoint-learning_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-x", "hip", "-I" + c, "-I" + os.path.join(ROOT, "include"), "-S",
                           "--cuda-device-only", "-o", S, os.path.join(c, "kernels.hip")], stderr=subprocess.DEVNULL)


def main():
    name, frag = sys.argv[1], sys.argv[2]
    ways = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    make()
    s = open(S).read()
    m = re.search(r"^(_Z\w*%s\w*):.*?\.amdhsa_kernel" % re.escape(name), s, re.S | re.M)
    lines = m.group(0).split("\n")
    idx = [i for i, l in enumerate(lines) if re.search(frag, l)]
    start = max(i for i, l in enumerate(lines[:idx[0]]) if l.startswith(".LBB"))
    end = min(i for i, l in enumerate(lines) if i > idx[-1] and "s_cbranch" in l)
    loop = lines[start:end + 1]
    valu = [l for l in loop if l.strip().startswith("v_")]
    print(m.group(1))
    print(len(valu), "VALU instructions in the loop;", len(valu) / ways, "per way")
    if "-v" in sys.argv:
        print("\n".join(l[:100] for l in loop))
    vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(m.group(1)), s)
    print("vgprs", vg.group(1) if vg else "?")


main()
