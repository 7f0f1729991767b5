"""VALU issue model of a kernel: how many SIMD cycles its vector instructions NEED at the measured issue ceilings.

Inputs
  * profiles/<tag>_valu_ceiling.json -- tools/valu_ceiling.hip run on the MI355X: cycles per wave-instruction and
    SIMD for each instruction kind at 1 / 2 / 4 / 8 resident waves (ipc_wall).  Three classes come out of it
    (>= 2 waves per SIMD):  FAST ~2.4 cycles (v_add/sub/mul_f32, v_add/sub_u32, v_and/or/xor_b32, v_mov_b32),
    SLOW ~4.2 (everything else, incl. v_fma_f32, v_cndmask, v_cmp, v_lshl*, v_min/max, v_cvt, v_floor, DPP, packed
    f32), TRANS ~8.3 (v_sqrt/rcp/rsq/exp/log/sin/cos_f32).
  * the dynamic class counters of the kernel (rocprofv3 --pmc): SQ_INSTS_VALU, _ADD_F32, _MUL_F32, _FMA_F32,
    _TRANS_F32, _CVT, _INT32.  ADD/MUL are FAST, FMA/CVT SLOW, TRANS TRANS; INT32 and the remainder (moves,
    selects, compares, min/max, floor, packed ops ...) mix FAST and SLOW kinds: they are split by the STATIC share of
    FAST opcodes among the integer resp. the remaining opcodes of the kernel's ISA (hipcc -S).
Output: needed_cycles_per_simd = sum over classes of count x cycles / 1024 SIMDs; valu_issue_frac = that / kernel
cycles (GRBM_GUI_ACTIVE / 8).  A fraction near 1 = the kernel is bound by VALU issue; the ceiling is measured, the
split of the two mixed classes is an estimate (both bounds are reported)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32",
        "v_or_b32", "v_xor_b32", "v_mov_b32"}
TRANS = {"v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
F32_COUNTED = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32"}
INT_PREFIXES = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshl", "v_lshr", "v_ashr",
                "v_mul_i32", "v_mul_u32", "v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32", "v_min_i32", "v_max_i32",
                "v_min_u32", "v_max_u32", "v_bfe", "v_bfi", "v_alignbit", "v_ffbh", "v_ffbl", "v_bcnt", "v_add_lshl",
                "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_add3", "v_xad", "v_not_b32", "v_mbcnt", "v_add_co", "v_sub_co",
                "v_addc_co", "v_subb_co", "v_add_i32", "v_sub_i32", "v_mad_u64", "v_lshl_add_u64")


def opcode(line):
    m = re.match(r"\s*(v_[a-z0-9_]+)", line)
    if not m:
        return None
    return re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m.group(1))


def static_shares(kernel_substr, asm_path=None):
    """(fast share among integer opcodes, fast share among the remaining non-f32-arithmetic opcodes, histogram)"""
    if asm_path is None:
        asm_path = "/tmp/kpl_kernels.s"
        src = os.path.join(ROOT, "keypoint-learning_amd", "csrc", "kernels.hip")
        if not os.path.exists(asm_path) or os.path.getmtime(asm_path) < os.path.getmtime(src):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-fno-fast-math", "-x", "hip", "--cuda-device-only", "-S", "-o", asm_path, src],
                                  stderr=subprocess.DEVNULL)
    text = open(asm_path).read()
    m = re.search(r"\n(_ZN[^\n:]*%s[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(kernel_substr), text, re.S)
    if not m:
        raise SystemExit("kernel %s not found in %s" % (kernel_substr, asm_path))
    hist = collections.Counter(op for op in map(opcode, m.group(2).splitlines()) if op)
    ints = {k: v for k, v in hist.items() if k.startswith(INT_PREFIXES)}
    rest = {k: v for k, v in hist.items() if k not in ints and k not in F32_COUNTED and k not in TRANS and not k.startswith("v_cvt")}
    share = lambda d: sum(v for k, v in d.items() if k in FAST) / max(sum(d.values()), 1)
    return share(ints), share(rest), hist


def class_cycles(ceiling, waves="w4"):
    k = ceiling["kinds"]
    cyc = lambda name: 1.0 / k[name][waves]["ipc_wall"]
    fast = sum(cyc(n) for n in ("v_add_f32", "v_mul_f32", "v_sub_f32", "v_add_u32", "v_and_b32", "v_mov_b32")) / 6.0
    slow = sum(cyc(n) for n in ("v_fma_f32", "v_cndmask_b32_sgpr", "v_cmp_lt_f32", "v_lshlrev_b32", "v_floor_f32", "v_cvt_i32_f32",
                                "v_alignbit_b32", "v_max_f32", "v_mul_i32_i24", "v_or_b32_dpp")) / 10.0
    trans = (cyc("v_sqrt_f32") + cyc("v_rcp_f32")) / 2.0
    return {"fast": fast, "slow": slow, "trans": trans}


def model(counters, cycles_per_xcd, ceiling, kernel_substr, simds=1024):
    """counters: SQ_INSTS_VALU*, per launch.  Returns a dict for profiles/counters.json."""
    c = class_cycles(ceiling)
    fi, fr, _ = static_shares(kernel_substr)
    tot = counters["SQ_INSTS_VALU"]
    add, mul, fma = counters.get("SQ_INSTS_VALU_ADD_F32", 0.0), counters.get("SQ_INSTS_VALU_MUL_F32", 0.0), counters.get("SQ_INSTS_VALU_FMA_F32", 0.0)
    trn, cvt, i32 = counters.get("SQ_INSTS_VALU_TRANS_F32", 0.0), counters.get("SQ_INSTS_VALU_CVT", 0.0), counters.get("SQ_INSTS_VALU_INT32", 0.0)
    rest = max(tot - add - mul - fma - trn - cvt - i32, 0.0)
    fixed = (add + mul) * c["fast"] + (fma + cvt) * c["slow"] + trn * c["trans"]
    mixed = lambda share_i, share_r: (i32 * (share_i * c["fast"] + (1 - share_i) * c["slow"]) +
                                      rest * (share_r * c["fast"] + (1 - share_r) * c["slow"]))
    need = (fixed + mixed(fi, fr)) / simds
    lo, hi = (fixed + mixed(1.0, 1.0)) / simds, (fixed + mixed(0.0, 0.0)) / simds
    return {"valu_issue_frac": round(need / cycles_per_xcd, 4),
            "valu_issue_frac_bounds": [round(lo / cycles_per_xcd, 4), round(hi / cycles_per_xcd, 4)],
            "needed_cycles_per_simd": round(need, 1), "kernel_cycles": round(cycles_per_xcd, 1),
            "class_cycles_per_instruction": {k: round(v, 3) for k, v in c.items()},
            "instructions_per_launch": {"total": tot, "add_f32": add, "mul_f32": mul, "fma_f32": fma, "trans_f32": trn, "cvt": cvt,
                                        "int32": i32, "other": rest},
            "static_fast_share": {"int32": round(fi, 3), "other": round(fr, 3)},
            "avg_cycles_per_instruction_at_ceiling": round(need * simds / tot, 3) if tot else None}


if __name__ == "__main__":
    fi, fr, hist = static_shares(sys.argv[1] if len(sys.argv) > 1 else "feature_kernelILb0")
    print("fast share: int %.3f other %.3f" % (fi, fr))
    for k, v in hist.most_common(40):
        print("  %-22s %4d %s" % (k, v, "FAST" if k in FAST else "TRANS" if k in TRANS else ""))
