#!/bin/bash
# Four PMC passes (SQ issue / wait / lanes, L1, L2 hit, clock) over the kernels of the queue-based pipeline: tools/pmc_wf_quick.sh <outdir (relative)> "<script + args>"
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/pmcwfq}; CMD=$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
run m1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run tcc3 TCC_HIT_sum TCC_MISS_sum
run g GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda:[0,0.0]); dur=collections.defaultdict(lambda:[0,0.0])
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            for tag in ("k_wf_traverse","k_wf_classify","k_wf_merge","k_wf_shade","k_intersect_batch","k_render_pass"):
                if tag in k: agg[(tag,r["Counter_Name"])][0]+=1; agg[(tag,r["Counter_Name"])][1]+=float(r["Counter_Value"])
    if d.rstrip("/").endswith("/g"):
        for f in glob.glob(d+"*/*_kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                for tag in ("k_wf_traverse","k_wf_classify","k_wf_merge","k_wf_shade"):
                    if tag in r["Kernel_Name"]: dur[tag][0]+=1; dur[tag][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6
for (tag,c),v in sorted(agg.items()): print(f"{tag:18s} {c:36s} launches={v[0]} total={v[1]:.6g}")
for tag,v in sorted(dur.items()): print(f"{tag:18s} kernel time under the profiler: launches={v[0]} total_ms={v[1]:.3f}")
PY
