out=gpurun_out/$1; mkdir -p $out
{
echo "== smart + shim"; timeout 900 python -m pytest tests/test_gpu_smart_factors.py tests/test_gpu_gtsam_shim.py -q -x 2>&1 | tail -40
echo "== shim log"; timeout 600 tests/_build/test_gpu_lm_gtsam 2>&1 | grep -i "smart\|FAIL\|PASSED" | head -60
} > $out/log.txt 2>&1
