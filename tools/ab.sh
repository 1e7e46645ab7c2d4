#!/bin/bash
# A/B of experiment builds on ONE box: tools/ab.sh <variant> [<variant> ...]   ("base" = the in-tree library)
# For each: image digests (must be identical across variants), the headline bench (2 steps), scene benchmarks.
SCENES=${SCENES:-atrium,jack,mesh6}
for v in "$@"; do
  if [ "$v" = base ]; then unset PTX_LIB; else export PTX_LIB=$PWD/distributed-path-tracer_amd/exp/libptx_$v.so; fi
  echo "=== $v"
  python tools/cmp_render.py 2>&1 | tail -2
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('cornell', j['value'], 'Msamples/s', j['roofline']['avg_launch_ms'], 'ms/launch')"
  [ -n "$SCENES" ] && python tools/bench_scenes.py --spp ${SPP:-8} --only $SCENES 2>&1 | grep scene\" | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(' ', j['scene'][:40], j['msamples_per_s'], 'Msamples/s', j['mrays_per_s'], 'Mrays/s')"
done
