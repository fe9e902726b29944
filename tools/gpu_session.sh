out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" ; timeout 900 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
echo "== gpu tests"; timeout 1400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dataflow_protocol.py tests/test_gpu_headline_parity.py -q -x 2>&1 | grep -v amdgpu.ids | tail -5
for w in ladybug1723 streets1723 sphere2500 w20000; do
  run python bench.py --cpu-baseline off --skip-dense-roofline --workload $w
done
} > $out/log.txt 2>&1
grep -E "passed|failed|Error" $out/log.txt | head
