#!/bin/bash
# A/B of library variants on the atrium render (32 spp, 512 Mi-pair pool) and the batch-intersect check: tools/wf_ab2.sh <outfile> <variant> ... ("base" = the shipped library)
OUT=$1; shift; mkdir -p $(dirname $OUT); : > $OUT
for v in "$@"; do
  if [ $v = base ]; then L=$PWD/distributed-path-tracer_amd/libptx_hip.so; else L=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo "== $v" >> $OUT
  PTX_LIB=$L PTX_WF_PAIRS_M=${POOL:-512} timeout -k 10 200 python tools/wf_render_check.py --only atrium --spp ${SPP:-32} --only-wavefront ${RENDER_ARGS:-} 2>&1 | grep -E '"scene"|rror|fault' >> $OUT || exit 1
  if [ -z "${NO_ISECT:-}" ]; then PTX_LIB=$L timeout -k 10 200 python tools/wf_intersect_check.py --only-wavefront 2>&1 | grep -E '"rays"|rror|fault' >> $OUT || exit 1; fi
done
