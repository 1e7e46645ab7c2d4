#!/bin/bash
# End-of-round measurement set (GPU box, repo root): tools/final_profiles.sh <outdir>
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/final}; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-psnr > $OUT/bench_under_rocprof.json 2> $OUT/stats.err)
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -5 $OUT/kernel_stats.csv
python tools/bench_scenes.py --spp 64 --level7 > $OUT/scenes.txt 2>&1; grep scene\" $OUT/scenes.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_atrium -- python3 $R/tools/bench_scenes.py --spp 16 --only atrium,mesh6 > $OUT/stats_atrium.log 2>&1)
find $OUT/stats_atrium -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_atrium_mesh6.csv; head -4 $OUT/kernel_stats_atrium_mesh6.csv
tools/pmc_passes.sh ${1:-gpurun_out/final}/pmc 64 > $OUT/pmc.txt 2>&1; tail -3 $OUT/pmc.txt
