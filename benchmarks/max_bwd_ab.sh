#!/bin/bash
# A/B of the max-aggregation backward variants on the products-sized graph (one box, same process settings):
#   rows = one-launch walk gathering arg-max rows; mask = per-edge arg-max bit masks + the same walk
for graph in uniform local; do
for v in rows mask; do
thr=0; [ $v = rows ] && thr=99999999999
echo "graph=$graph variant=$v"
DGCN_MAXTHR=$thr GRAPH=$graph timeout 300 python - <<PY 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"bwd_launch_ms_avg\": [0-9.]*\|\"fwd_launch_ms_avg\": [0-9.]*"
import os, sys
from deep_gcns_torch_amd import ops
ops.MAX_MASK_MIN_TABLE_BYTES = int(os.environ["DGCN_MAXTHR"])
sys.argv = ["bench.py", "--aggr", "max", "--graph", os.environ["GRAPH"], "--no-cpu-baseline", "--no-extras", "--steps", "5", "--warmup", "2"]
exec(open("bench.py").read())
PY
done; done
