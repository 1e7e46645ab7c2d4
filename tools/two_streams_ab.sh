#!/bin/bash
# One slab at a time against two half-size slabs side by side on two streams (PTX_WF_TWO_STREAMS), per scene and frame size
OUT=$1; mkdir -p $(dirname $OUT); : > $OUT
for sc in jack atrium; do for size in 1920x1080 960x540 480x270; do for ts in 0 1; do
  echo "== $sc $size two_streams=$ts" >> $OUT
  if [ $ts = 1 ]; then export PTX_WF_TWO_STREAMS=1; else unset PTX_WF_TWO_STREAMS; fi
  timeout -k 10 300 python tools/wf_render_check.py --only $sc --size $size --spp 64 --only-wavefront 2>&1 | grep -E '"scene"|rror|fault' >> $OUT || exit 1
done; done; done
