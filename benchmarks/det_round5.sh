python tests/guard_alloc/revgcn_sequence.py --winner 1 --rows modelfile_fused,modelfile_fused_graph --steps 6 --replays 10 2>&1 | grep "ms per"
for v in dynamic static; do
  unset DGCN_STATIC_ITEMS
  [ $v = static ] && export DGCN_STATIC_ITEMS=1
  echo "== bench $v"
  python tests/guard_alloc/bench_guarded.py --winner 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{\"metric\"'):
        r=json.loads(line)['extra']['revgcn_proteins']
        print({k:round(v['ms_per_step'],2) for k,v in r.items() if isinstance(v,dict) and 'ms_per_step' in v and 'fuse_models' in k})
"
done
