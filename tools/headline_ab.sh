#!/bin/bash
# Headline frame (Cornell 1080p, 256 spp, 8 bounces) through library variants, alternating: tools/headline_ab.sh <outfile> <variant> ... ("base" = the shipped library)
OUT=$1; shift; mkdir -p $(dirname $OUT); : > $OUT
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then L=$PWD/distributed-path-tracer_amd/libptx_hip.so; else L=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo -n "$v " >> $OUT
  PTX_LIB=$L timeout -k 10 300 python bench.py --no-configs --no-cpu-baseline --no-psnr --steps 4 --warmup 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $OUT || exit 1
done; done
