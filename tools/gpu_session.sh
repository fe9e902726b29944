out=gpurun_out/$1; mkdir -p $out
{
echo "== smart"; timeout 900 python -m pytest tests/test_gpu_smart_factors.py tests/test_gpu_sharding.py -q -k "smart or epi" 2>&1 | tail -30
echo "== shim"; timeout 600 tests/_build/test_gpu_lm_gtsam 2>&1 | grep -i "EPI\|FAIL\|PASSED" | head -20
} > $out/log.txt 2>&1
