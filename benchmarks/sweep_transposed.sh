#!/bin/bash
# single-rank stand-in for the per-rank work of the channel-transposed scheme at world = 128/C
mkdir -p gpurun_out
: > gpurun_out/sweep_transposed.jsonl
for c in "$@"; do
  for k in 1 2; do
    python bench.py --channels $c --no-cpu-baseline --steps 10 --warmup 3 --force-partitioned --scheme transposed \
        --pipeline-chunks $k 2>/dev/null | grep '^{' >> gpurun_out/sweep_transposed.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/sweep_transposed.jsonl"):
    j = json.loads(l)
    print(j["config"]["workload"].split()[-1], j["ms_per_step"], j["roofline"]["launch_ms_avg"])
PY
