#!/usr/bin/env python3
"""tools/isa_mix.py [out.json] -- the VALU instruction mix of the headline's two row-NTT kernels (ntt_pass_l9s_kernel<8, 2, true, false> and
<10, 0, false, false>: LcEncoding::encode of lcpc-ligero-pc/src/lib.rs:162-164 at 2^18 columns) from the gfx950 disassembly, priced with
the issue cost of each instruction class measured on an MI355X (profiles/r06_ubench_mix.txt = `tools/ubench_valu mix`, 4 waves per SIMD
like the kernels; profiles/r02_ubench_valu.txt for the rest), to state ONE ceiling for "how many wave instructions per second could this
stream issue": VERDICT r5 item 5.  Runs without a GPU (hipcc -S --cuda-device-only).

Classes and their measured cost in cycles per wave instruction per SIMD at the nominal 2.4 GHz (i.e. 1 / rate, whatever the clock did):
  mad64      v_mad_i64_i32 / v_mad_u64_u32, vector operands        4.79   (one accumulator or eight: the same)
  mad64_s    the same with an SGPR multiplicand (l9::mul_u)        4.46
  mul32      v_mul_lo_u32 / v_mul_hi_u32                           4.32
  vop3       every other 64-bit encoding (add3, alignbit, bfe, 64-bit shifts, lshl_add, carry adds, cndmask, lane moves)   4.30
  vop2       32-bit encodings, with or without a 32-bit literal    2.45
The counts are STATIC (every path of the kernel, the untaken zero-padding variants included); the dynamic total comes from
SQ_INSTS_VALU (profiles/pmc_latest.json), which bench.py combines with this file's mean cost: weighted peak = 1024 SIMDs x 2.4e9 /
mean cost, scaled by (sampled clock / the ~2.36 GHz the microbenchmark ran at, profiles/r01f_ubench_power.txt)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lcpc_amd", "csrc")
COST = {"mad64": 4.79, "mad64_s": 4.46, "mul32": 4.32, "vop3": 4.30, "vop2": 2.45}
UBENCH_CLOCK_GHZ = 2.36          # rocm-smi while the same microbenchmark kernels loop (profiles/r01f_ubench_power.txt: 2.33-2.39)
KERNELS = {"first pass, 8 stages": "ntt_pass_l9s_kernelILi8ELi2ELb1ELb0E", "last pass, 10 stages": "ntt_pass_l9s_kernelILi10ELi0ELb0ELb0E"}
VOP2_NAMES = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_lshlrev_b32", "v_lshrrev_b32",
              "v_ashrrev_i32", "v_not_b32", "v_min_u32", "v_max_u32", "v_mul_u32_u24", "v_mul_i32_i24"}
CARRY = ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32")


def classify(mn, rest):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base in ("v_mad_i64_i32", "v_mad_u64_u32"):
        srcs = rest.split(",")[2:4]
        return "mad64_s" if any(o.strip().startswith("s") for o in srcs) else "mad64"
    if base in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32"):
        return "mul32"
    if base in CARRY or base.startswith("v_cndmask") or base.startswith("v_cmp") or base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "vop3"
    if mn.endswith("_e32") or base in VOP2_NAMES and not mn.endswith("_e64"):
        return "vop2"
    return "vop3"


def disassemble():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "l9s.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               os.path.join(CSRC, "ntt_l9s.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def main():
    import bench
    lines = disassemble()
    res = {"kernel_stamp": bench.kernel_stamp(), "cost_cycles_at_nominal_2p4GHz": COST, "ubench_clock_GHz": UBENCH_CLOCK_GHZ,
           "source": "hipcc -S of lcpc_amd/csrc/ntt_l9s.hip (static counts, all paths) x profiles/r06_ubench_mix.txt / r02_ubench_valu.txt", "kernels": {}}
    for label, key in KERNELS.items():
        start = [i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") is False and re.match(r"^_ZN\S+:", l)][0]
        end = start
        while not lines[end].lstrip().startswith(".Lfunc_end"):
            end += 1
        cls, names = collections.Counter(), collections.Counter()
        for l in lines[start:end]:
            m = re.match(r"\s+(v_[a-z0-9_]+)\s+(.*)", l)
            if m:
                c = classify(m.group(1), m.group(2))
                cls[c] += 1
                names[re.sub(r"_(e32|e64)$", "", m.group(1))] += 1
        n = sum(cls.values())
        mean = sum(COST[c] * k for c, k in cls.items()) / n
        res["kernels"][label] = {"symbol": key, "valu_static": n, "by_class": dict(cls), "mean_cost_cycles": round(mean, 3),
                                 "weighted_peak_Ginst_per_s_at_ubench_clock": round(1024 * 2.4 / mean, 1),
                                 "top_mnemonics": dict(names.most_common(12))}
    means = [k["mean_cost_cycles"] for k in res["kernels"].values()]
    res["mean_cost_cycles"] = round(sum(means) / len(means), 3)
    res["weighted_peak_Ginst_per_s_at_ubench_clock"] = round(1024 * 2.4 / res["mean_cost_cycles"], 1)
    text = json.dumps(res, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
