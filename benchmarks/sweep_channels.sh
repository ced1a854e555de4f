#!/bin/bash
# raw single-GPU aggregation over channel widths: ms/step and forward launch ms
mkdir -p gpurun_out
for c in "$@"; do
  python bench.py --channels $c --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' | \
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C=%s step %.3f ms fwd %.3f ms frac %.3f' % ('$c', j['ms_per_step'], j['roofline']['launch_ms_avg'], j['roofline']['frac']))"
done | tee gpurun_out/sweep_channels.txt
