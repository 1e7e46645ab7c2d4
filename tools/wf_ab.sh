#!/bin/bash
# A/B of wavefront.hip build variants on the intersect check: tools/wf_ab.sh <outfile> <variant> ... ("base" = the shipped library)
OUT=$1; shift; mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ $v = base ]; then L=$PWD/distributed-path-tracer_amd/libptx_hip.so; else L=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo "== $v" >> $OUT
  PTX_LIB=$L timeout -k 10 200 python tools/wf_intersect_check.py --only-wavefront ${WF_ARGS:-} 2>&1 | grep -E '"rays"|WFPROF|WFCLK|WFMAX|WFHIST|Error|error' >> $OUT
done
