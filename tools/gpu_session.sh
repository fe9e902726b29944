out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" ; timeout 900 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
echo "== gpu tests"; timeout 1400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_device_analysis.py tests/test_gpu_sharding.py -q -x 2>&1 | grep -v amdgpu.ids | tail -5
for w in ladybug1723 venice1778 streets1723; do
  run python bench.py --cpu-baseline off --skip-dense-roofline --workload $w
  GTG_SCHUR_XCD=0 run python bench.py --cpu-baseline off --skip-dense-roofline --workload $w
done
GTG_SCHUR_WG=4 run python bench.py --cpu-baseline off --skip-dense-roofline --workload ladybug1723
GTG_SCHUR_WG=16 run python bench.py --cpu-baseline off --skip-dense-roofline --workload ladybug1723
} > $out/log.txt 2>&1
grep -E "passed|failed|Error" $out/log.txt | head
