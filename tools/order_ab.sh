#!/bin/bash
# Start order of the surfaces' queues in the traverse kernel (PTX_WF_ORDER: 0 surface order, 1 largest tree first, 2 smallest first)
OUT=$1; mkdir -p $(dirname $OUT); : > $OUT
for m in ${MODES:-0 1 2}; do
  echo "== PTX_WF_ORDER=$m" >> $OUT
  PTX_WF_ORDER=$m timeout -k 10 300 python tools/wf_render_check.py --only atrium --spp ${SPP:-32} --only-wavefront 2>&1 | grep -E '"scene"|rror|fault|differ' >> $OUT || exit 1
  PTX_WF_ORDER=$m timeout -k 10 200 python tools/wf_intersect_check.py 2>&1 | grep -E '"rays"|rror|fault|differ' >> $OUT || exit 1
done
