#!/bin/bash
# Dynamic VALU instruction mix of the integrator kernel by the SQ_INSTS_VALU_* category counters, plus a calibration of those
# categories on the single-instruction loops of tools/bin/valu_issue_bench (which counter counts which opcode).
# Usage (GPU box, repo root): tools/pmc_mix.sh <outdir> [spp]
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/mix}; SPP=${2:-64}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
B="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"
run() { name=$1; shift; ctr=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run cal_a "$A" $R/tools/bin/valu_issue_bench
run cal_b "$B" $R/tools/bin/valu_issue_bench
run k_a "$A" python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline
run k_b "$B" python3 $R/bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "k_render_pass" in k: k="k_render_pass"
            a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        print("==",d)
        for k,cs in agg.items():
            n=max(v[0] for v in cs.values())
            print(f"{k[:40]:40s} n={n:3d} "+" ".join(f"{c.replace('SQ_INSTS_VALU_','').replace('SQ_','')}={v[1]/v[0]:.4g}" for c,v in sorted(cs.items())))
PY
