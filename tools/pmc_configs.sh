#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, FETCH_SIZE / WRITE_SIZE in passes of their own) over the dominant kernel of each benchmark
# scene class of tools/bench_configs.py -> <outdir>/configs_pmc.json (copy to profiles/roundN_configs_pmc.json: bench.py replays it).
# Usage (GPU box, repo root): tools/pmc_configs.sh <outdir (relative)>
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/pmc_configs}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for cfg in config3_mesh82k_1080p_8b config4_atrium_1080p_8b config5_atrium_4k_16b jack_of_blades_1080p_8b; do
  run() { name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$cfg/$name -- python3 $R/tools/bench_configs.py --only $cfg --no-timing --spp-scale 0.5 > $OUT/$cfg.$name.log 2>&1
    echo "$cfg $name rc=$?"; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
  run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
  run tcc1 FETCH_SIZE
  run tcc2 WRITE_SIZE
  run tcc3 TCC_HIT_sum TCC_MISS_sum
  run g GRBM_GUI_ACTIVE
done
python3 $R/tools/pmc_configs_reduce.py $OUT
