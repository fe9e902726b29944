out=gpurun_out/$1; mkdir -p $out
{
echo "== device analysis A/B + parity"; timeout 900 python -m pytest tests/test_gpu_device_analysis.py tests/test_gpu_parity.py tests/test_gpu_smart_factors.py -q -x 2>&1 | tail -12
echo "== bench"; GTG_DEBUG_TIMING=1 timeout 600 python bench.py --cpu-baseline off --skip-dense-roofline 2>&1 | grep -v "amdgpu" | grep "setup\]\|metric" | cut -c1-900 | tail -40
} > $out/log.txt 2>&1
