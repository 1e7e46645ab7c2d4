#!/usr/bin/env python3
"""Static instruction mix of one kernel in csrc/kernels.s (make asm), by issue-cost class of profiles/round2_valu_issue.txt.
usage: tools/isa_mix.py <substring of the mangled kernel name> [kernels.s]
Also lists v_cndmask_b32_e32 whose VCC was last written by a SALU instruction (12 cycles each on gfx950, see the table)."""
import collections
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_fmac_f32", "v_not_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_iflag_f32"}


def main():
    key = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else "distributed-path-tracer_amd/csrc/kernels.s"
    inside = False
    cnt = collections.Counter()
    vcc_writer = None
    slow_cnd = 0
    cnd_total = 0
    lines = []
    with open(path) as fh:
        for ln in fh:
            if not inside:
                if ln.startswith("_ZN") and ":" in ln.split(";")[0] and key in ln:
                    inside = True
                    name = ln.split(":")[0]
                continue
            if ln.startswith("\t.size") or ln.startswith(".Lfunc_end"):
                break
            s = ln.strip()
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            op = s.split()[0]
            lines.append(s)
            cnt[op] += 1
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
            if base == "v_cndmask_b32":
                cnd_total += 1
                if op.endswith("e32") or "vcc" in s.split(";")[0].split(",")[-1]:
                    if vcc_writer and vcc_writer.startswith("s_"):
                        slow_cnd += 1
            # who writes vcc
            body = s.split(";")[0]
            ops = body.split(None, 1)[1] if " " in body else ""
            dst = ops.split(",")[0].strip() if ops else ""
            if dst == "vcc" or (op.startswith("v_cmp") and op.endswith("e32")) or (base in ("v_div_scale_f32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_mad_u64_u32") and "vcc" in ops.split(",")[1:2].__str__()):
                vcc_writer = op
    tot_v = sum(v for k, v in cnt.items() if k.startswith("v_"))
    cls = collections.Counter()
    for k, v in cnt.items():
        if not k.startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", k)
        cls["fast (2.0)" if base in FAST else "trans (6.1)" if base in TRANS else "slow (3.1)"] += v
    print(name)
    print("static VALU:", tot_v, dict(cls), " SALU:", sum(v for k, v in cnt.items() if k.startswith("s_")), " DS:", sum(v for k, v in cnt.items() if k.startswith("ds_")),
          " VMEM:", sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))))
    print("v_cndmask total", cnd_total, " with SALU-written vcc:", slow_cnd)
    for k, v in cnt.most_common(45):
        print(f"  {k:28s} {v}")


if __name__ == "__main__":
    main()
