out=gpurun_out/$1; mkdir -p $out
{
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15
} > $out/gpu_tests.log 2>&1
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --skip-dense-roofline --cpu-baseline off --workload streets1723 > $out/bench_streets1723.json 2> $out/bench_streets.err
timeout 600 tests/_build/test_gpu_lm_gtsam > $out/shim_test.log 2>&1
tail -3 $out/gpu_tests.log; tail -c 400 $out/bench.json; tail -2 $out/shim_test.log
