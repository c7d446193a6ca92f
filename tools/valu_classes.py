#!/usr/bin/env python3
"""Price the VALU-issue bound with the MEASURED issue classes.

tools/ubench_issue.hip on MI355X (profiles/r02_ubench_issue.jsonl): with every SIMD saturated a wave-instruction costs ~2.3 cycles
when it is a plain VOP2 / VOP1 in the 32-bit (_e32) encoding with VGPR, inline-constant or literal operands - v_add/sub/subrev_u32,
v_and/or/xor_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, v_add/sub/mul/fmac_f32, v_add_u16, v_max_i16 - and ~4.15 cycles for
everything else (any VOP3 / _e64 encoding, packed 16-bit, DPP, SDWA, an SGPR operand, v_mul_*24, v_mul_lo, 32-bit min / max,
v_lshlrev_b32, conversions, v_perm, v_sad_*, v_lerp, compares, readlane / writelane).

usage: valu_classes.py <asm.s> [profiles/rNN_summary.json]
Classifies every VALU instruction of every kernel of the disassembly (hipcc -S --cuda-device-only) STATICALLY; with a PMC summary
the dynamic instruction count of the lockstep launch (SQ_INSTS_VALU) is split by the kernel's static class mix - an
approximation (loops weigh their body by trip count in the dynamic count and not in the static mix), stated as such - and
priced at 2.3 / 4.15 cycles.  Prints one JSON object; bench.py reads profiles/rNN_valu_classes.json."""
import json, re, subprocess, sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_add_u16", "v_max_i16", "v_not_b32", "v_subrev_f32"}
FAST_CYC, SLOW_CYC = 2.3, 4.15


def classify(line):
    t = line.split(";")[0].strip()
    m = re.match(r"^(v_[a-z0-9_]+)\s*(.*)$", t)
    if not m:
        return None
    op, args = m.group(1), m.group(2)
    if op.endswith("_dpp") or op.endswith("_sdwa") or op.endswith("_e64"):
        return "slow"
    base = op[:-4] if op.endswith("_e32") else op
    if base not in FAST:
        return "slow"
    ops = [a.strip() for a in args.split(",")]
    # a scalar register operand (s12, s[4:5], vcc, exec, m0 ...) makes the instruction issue at the slow rate
    for a in ops[1:]:
        if re.match(r"^(s\d+|s\[|vcc|exec|m0|ttmp|src_)", a):
            return "slow"
    return "fast"


def main():
    s = open(sys.argv[1]).read()
    out = {"what": "static VALU class mix per kernel (fast = plain 32-bit-encoded VOP1/VOP2 on VGPR / constant operands: ~2.3 cycles per wave-instruction "
                   "with the SIMD saturated; slow = everything else: ~4.15), tools/ubench_issue.hip + profiles/r02_ubench_issue.jsonl", "kernels": {}}
    for m in re.finditer(r"^(_ZN4h264\w+):(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name = m.group(1)
        try:   # the name the profiler prints: k_me<false>, k_deblock_pairs<false, false> ...
            name = subprocess.check_output(["c++filt", name], text=True).strip().split("(")[0].replace("void ", "").replace("h264::", "")
        except Exception:
            pass
        c = {"fast": 0, "slow": 0}
        for l in m.group(2).splitlines():
            k = classify(l)
            if k:
                c[k] += 1
        tot = c["fast"] + c["slow"]
        if tot:
            out["kernels"][name] = {"fast": c["fast"], "slow": c["slow"], "fast_fraction": round(c["fast"] / tot, 3),
                                    "cycles_per_valu": round((c["fast"] * FAST_CYC + c["slow"] * SLOW_CYC) / tot, 3)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
