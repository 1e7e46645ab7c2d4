#!/bin/bash
# Render configs 4 / 5 / jack-of-blades through library variants: tools/cfg_ab.sh <outfile> <variant> ... ("base" = the shipped library)
OUT=$1; shift; mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ $v = base ]; then L=$PWD/distributed-path-tracer_amd/libptx_hip.so; else L=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo "== $v" >> $OUT
  PTX_LIB=$L timeout -k 10 400 python tools/bench_configs.py --only ${ONLY:-config4_atrium_1080p_8b,config5_atrium_4k_16b,jack_of_blades_1080p_8b} 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        for k,v in d.items(): print(k, v.get('msamples_per_s'), v.get('kernel_total_ms'), v.get('classify_total_ms'), v.get('shade_total_ms'), v.get('traverse_drain_frac'))
" >> $OUT || exit 1
done
