#!/bin/bash
# End-of-round measurement set (GPU box, repo root): tools/final_profiles.sh <outdir (relative)>
set -u
R=$PWD; REL=${1:-gpurun_out/final}; OUT=$R/$REL; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-psnr > $OUT/bench_under_rocprof.json 2> $OUT/stats.err)
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -5 $OUT/kernel_stats.csv
python tools/bench_scenes.py --spp 64 --level7 > $OUT/scenes.txt 2>&1; grep scene\" $OUT/scenes.txt
# the many-surface scene on the fused kernel, same box (the queue-based pipeline is its default)
PTX_WAVEFRONT=0 python tools/bench_scenes.py --spp 64 --only atrium > $OUT/scenes_atrium_fused.txt 2>&1; grep scene\" $OUT/scenes_atrium_fused.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_atrium -- python3 $R/tools/bench_scenes.py --spp 16 --only atrium,mesh6 > $OUT/stats_atrium.log 2>&1)
find $OUT/stats_atrium -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_atrium_mesh6.csv; head -8 $OUT/kernel_stats_atrium_mesh6.csv
python tools/wf_intersect_check.py > $OUT/wf_intersect.txt 2>&1; grep rays\" $OUT/wf_intersect.txt
tools/bin/gather_bench > $OUT/gather_bench.txt 2>&1; tail -3 $OUT/gather_bench.txt
tools/pmc_wf.sh $REL/pmc_wf "tools/bench_scenes.py --spp 16 --only atrium" > $OUT/pmc_wf.txt 2>&1; grep -c k_wf $OUT/pmc_wf.txt
tools/pmc_passes.sh $REL/pmc 64 > $OUT/pmc.txt 2>&1; tail -3 $OUT/pmc.txt
