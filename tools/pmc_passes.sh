#!/bin/bash
# Collect PMC counters for the integrator kernel in separate rocprofv3 passes (<= 8 SQ counters each).
# Usage (on the GPU box, from the repo root): tools/pmc_passes.sh <outdir> [spp] [script + args instead of bench.py, e.g. "tools/bench_scenes.py --only mesh6 --spp 8"]
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/pmc}; SPP=${2:-16}
CMD=${3:-"bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline"}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if "k_render_pass" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][0]+=1; agg[r["Counter_Name"]][1]+=float(r["Counter_Value"])
        for k,v in agg.items(): print(f"{k:28s} launches={v[0]} per_launch={v[1]/v[0]:.6g}")
PY
