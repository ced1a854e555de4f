for v in base static nokeep; do
  unset DGCN_STATIC_ITEMS DGCN_NO_KEEP
  [ $v = static ] && export DGCN_STATIC_ITEMS=1
  [ $v = nokeep ] && export DGCN_NO_KEEP=1
  echo "== $v"; python tests/guard_alloc/first_divergence.py 8 2>&1 | grep "^step"
done
