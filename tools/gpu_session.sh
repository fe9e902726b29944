out=gpurun_out/$1; mkdir -p $out
{
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15
} > $out/gpu_tests.log 2>&1
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/gpu_tests.log; tail -c 600 $out/bench.json
