out=gpurun_out/$1; mkdir -p $out
{
echo "== shim"; timeout 600 tests/_build/test_gpu_lm_gtsam 2>&1 | grep -i "GNC\|FAIL\|PASSED" | head -20
} > $out/log.txt 2>&1
