out=gpurun_out/guard2; mkdir -p $out
for fm in keepva never; do
  for blocking in "" "--no-blocking"; do
    python tests/guard_alloc/run.py --mode back --free $fm $blocking --timeout 200 --log /tmp/g_$fm.log -- python tests/guard_alloc/selftest.py > $out/selftest_back_${fm}${blocking}.log 2>&1
    echo "free=$fm $blocking rc=$?"; tail -n 6 $out/selftest_back_${fm}${blocking}.log | cut -c1-300
  done
done
