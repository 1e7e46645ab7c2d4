#!/bin/bash
# Memory-side PMC passes (vector L1 / TA / address translation) for one command. Usage: tools/pmc_mem.sh <outdir> "<script + args>"
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/pmcmem}; CMD=$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run m1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run m2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run m3 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
# (the TA_* counters hang rocprofv3 on this pool: not collected)
run m5 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run g GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if "k_render_pass" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][0]+=1; agg[r["Counter_Name"]][1]+=float(r["Counter_Value"])
        for k,v in agg.items(): print(f"{k:40s} launches={v[0]} total={v[1]:.6g}")
PY
