#!/bin/bash
# The round's measurement set, on the GPU box from the repo root: tools/final_profiles_r3.sh <outdir (relative, e.g. gpurun_out/final)>
# Copy what it writes into profiles/round3_* afterwards (the list is printed at the end).
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/final}; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-psnr > $OUT/bench_under_rocprof.json 2> $OUT/stats.err); echo "rocprof stats rc=$?"
cp $(ls $OUT/stats/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
tools/pmc_passes.sh ${1:-gpurun_out/final}/pmc_headline 64 > $OUT/pmc_headline.txt 2>&1; echo "pmc headline rc=$?"
tools/pmc_configs.sh ${1:-gpurun_out/final}/pmc_configs > $OUT/pmc_configs.txt 2>&1; echo "pmc configs rc=$?"
if [ -f distributed-path-tracer_amd/exp/libptx_clk.so ]; then
  PTX_WAVEFRONT=0 PTX_LIB=$R/distributed-path-tracer_amd/exp/libptx_clk.so timeout -k 10 300 python tools/bench_scenes.py --spp 64 --only cornell,jack,plaza,mesh6 > $OUT/clk_fused.txt 2>&1; echo "clk rc=$?"
fi
if [ -f distributed-path-tracer_amd/exp/libptx_prof.so ]; then
  PTX_LIB=$R/distributed-path-tracer_amd/exp/libptx_prof.so timeout -k 10 300 python tools/wf_intersect_check.py --only-wavefront --reps 1 > $OUT/wf_prof.txt 2>&1; echo "wf prof rc=$?"
fi
timeout -k 10 300 python tools/wf_intersect_check.py > $OUT/wf_intersect.txt 2>&1; echo "intersect rc=$?"
timeout -k 10 300 python tools/bench_scenes.py --spp 64 --level7 > $OUT/scenes.txt 2>&1; echo "scenes rc=$?"
ls $OUT
