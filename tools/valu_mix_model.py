#!/usr/bin/env python3
"""Mix-weighted VALU busy fraction of the integrator kernel.

Inputs: (1) the dynamic instruction counts per SQ_INSTS_VALU_* category of one launch (tools/pmc_passes.sh -> kernel_pmc.json),
(2) the issue cost of each opcode per SIMD at 4 waves per SIMD (profiles/round2_valu_issue.txt, tools/valu_issue_bench.hip),
(3) the STATIC opcode shares inside each category from the kernel's ISA (csrc/kernels.s, `make asm`), used to split a category
whose members have different costs (e.g. ADD_F32 = v_add_f32 / v_sub_f32 at 2.0 cycles and v_pk_add_f32 at 3.15).
Which counter counts which opcode was calibrated on the single-opcode loops of the microbenchmark (profiles/round2_pmc_category_calibration.txt).

usage: tools/valu_mix_model.py <kernel_pmc.json> <mangled-name substring> [kernels.s] -> prints the JSON with the model fields added
"""
import collections
import json
import re
import sys

# cycles per wave64 instruction per SIMD at W = 4 (profiles/round2_valu_issue.txt)
COST = {"fast": 2.00, "slow": 3.12, "fma": 2.89, "fmac": 2.07, "trans": 6.08, "f64": 3.14}
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_not_b32", "v_mov_b64"}
CATEGORY = {  # opcode -> PMC category (calibration run)
    "ADD_F32": {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32"},
    "MUL_F32": {"v_mul_f32", "v_pk_mul_f32", "v_mul_legacy_f32"},
    "FMA_F32": {"v_fma_f32", "v_fmac_f32", "v_pk_fma_f32", "v_div_fmas_f32", "v_mad_f32", "v_mac_f32"},
    "TRANS_F32": {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"},
    "INT32": {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_u32_u24", "v_lshl_add_u32", "v_add3_u32", "v_bfe_u32", "v_mad_u32_u24", "v_mbcnt_lo_u32_b32",
              "v_mbcnt_hi_u32_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_mul_i32_i24", "v_mad_i32_i24",
              "v_add_lshl_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_xad_u32", "v_bfi_b32", "v_alignbit_b32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32"},
    "INT64": {"v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_mad_i64_i32"},
    "CVT": {"v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f32_ubyte0"},
    "F64": {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_max_f64", "v_min_f64"},
}


def op_cost(base):
    if base in FAST:
        return COST["fast"]
    if base == "v_fma_f32":
        return COST["fma"]
    if base == "v_fmac_f32":
        return COST["fmac"]
    if base in CATEGORY["TRANS_F32"] or base in ("v_rcp_f64", "v_rsq_f64"):
        return COST["trans"]
    if base in CATEGORY["F64"]:
        return COST["f64"]
    return COST["slow"]


def static_mix(path, key):
    cnt = collections.Counter()
    inside = False
    with open(path) as fh:
        for ln in fh:
            if not inside:
                inside = ln.startswith("_ZN") and ":" in ln.split(";")[0] and key in ln
                continue
            if ln.startswith("\t.size") or ln.startswith(".Lfunc_end"):
                break
            s = ln.strip()
            if s.startswith("v_"):
                cnt[re.sub(r"_(e32|e64|sdwa|dpp)$", "", s.split()[0])] += 1
    return cnt


def main():
    pmc = json.load(open(sys.argv[1]))
    mix = static_mix(sys.argv[3] if len(sys.argv) > 3 else "distributed-path-tracer_amd/csrc/kernels.s", sys.argv[2])
    c = pmc["counters_per_launch"]
    dyn = {"ADD_F32": c["SQ_INSTS_VALU_ADD_F32"], "MUL_F32": c["SQ_INSTS_VALU_MUL_F32"], "FMA_F32": c["SQ_INSTS_VALU_FMA_F32"],
           "TRANS_F32": c["SQ_INSTS_VALU_TRANS_F32"], "INT32": c["SQ_INSTS_VALU_INT32"], "INT64": c["SQ_INSTS_VALU_INT64"], "CVT": c["SQ_INSTS_VALU_CVT"],
           "F64": c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]}
    total = c["SQ_INSTS_VALU"]
    dyn["other (mov, cndmask, cmp, min/max, readlane, shifts, div_scale/fixup)"] = total - sum(dyn.values())
    categorized = set().union(*CATEGORY.values())
    rows, cycles = {}, 0.0
    for cat, n in dyn.items():
        members = {op: k for op, k in mix.items() if (op in CATEGORY[cat] if cat in CATEGORY else op not in categorized)}
        w = sum(members.values())
        avg = sum(op_cost(op) * k for op, k in members.items()) / w if w else COST["slow"]
        rows[cat] = {"wave_instructions": n, "avg_cycles": round(avg, 3)}
        cycles += n * avg
    simd_cycles = 1024.0 * pmc["cycles_per_launch"]
    pmc["valu_mix_model"] = {"categories": rows, "avg_cycles_per_instruction": round(cycles / total, 3),
                             "note": "cycles per wave64 instruction per SIMD at 4 waves per SIMD from profiles/round2_valu_issue.txt; static opcode shares "
                                     "of csrc/kernels.s split each PMC category"}
    pmc["valu_busy_frac_mix_weighted"] = round(cycles / simd_cycles, 4)
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
