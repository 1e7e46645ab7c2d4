#!/bin/bash
# A/B of library variants on the atrium render: tools/wf_render_ab.sh <outfile> <spp> <variant> ... ("base" = the shipped library)
OUT=$1; SPP=$2; shift 2; mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ $v = base ]; then L=$PWD/distributed-path-tracer_amd/libptx_hip.so; else L=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo "== $v" >> $OUT
  PTX_LIB=$L timeout -k 10 300 python tools/wf_render_check.py --only atrium --spp $SPP ${WF_ARGS:-} 2>&1 | tail -1 | cut -c1-330 >> $OUT
done
