out=gpurun_out/$1; mkdir -p $out
{
echo "== sharded smart + smart"; timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_smart_factors.py -q -k "smart" 2>&1 | tail -40
} > $out/log.txt 2>&1
