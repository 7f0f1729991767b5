"""tools/scratch_report.py [kernel-name-fragment]: where the scratch (private segment) instructions of a kernel sit relative to its
loops, from the device ISA of csrc/kernels.hip (a loop = a backward branch to a label).  Default: the dominant kernel,
feature_kernel<false, 2>, whose 96-VGPR cap leaves 8 bytes of scratch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "build", "exp", "kernels.s")


def main():
    frag = sys.argv[1] if len(sys.argv) > 1 else "14feature_kernelILb0ELi2EE"
    c = os.path.join(ROOT, "keypoint-learning_amd", "csrc")
    os.makedirs(os.path.dirname(S), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-x", "hip", "-I" + c, "-I" + os.path.join(ROOT, "include"), "-S",
                           "--cuda-device-only", "-o", S, os.path.join(c, "kernels.hip")], stderr=subprocess.DEVNULL)
    s = open(S).read()
    m = re.search(r"^(_Z\w*%s\w*):(.*?)\.amdhsa_kernel" % re.escape(frag), s, re.S | re.M)
    name, lines = m.group(1), m.group(2).split("\n")
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
    loops = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    priv = re.search(r"\.set %s\.private_seg_size, (\d+)" % re.escape(name), s)
    vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(name), s)
    print(name)
    print("vgprs %s, private segment %s bytes, %d ISA lines, %d loops (backward branches)" % (vg.group(1) if vg else "?", priv.group(1) if priv else "?", len(lines), len(loops)))
    inside_any = 0
    for i, l in enumerate(lines):
        if "scratch_" in l:
            inside = [lp for lp in loops if lp[0] <= i <= lp[1]]
            inside_any += bool(inside)
            print("  line %5d  %-44s %s" % (i, l.strip(), "INSIDE loop(s) %s" % inside if inside else "outside every loop"))
    print("scratch instructions inside a loop: %d" % inside_any)
    return 1 if inside_any else 0


if __name__ == "__main__":
    sys.exit(main())
