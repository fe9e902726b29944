out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" ; timeout 900 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
for i in 1 2 3; do run python bench.py --cpu-baseline off --skip-dense-roofline --workload streets1723; done
GTG_CHOL=streams run python bench.py --cpu-baseline off --skip-dense-roofline --workload streets1723
GTG_HOST_ANALYSIS=1 run python bench.py --cpu-baseline off --skip-dense-roofline --workload streets1723
run python -m pytest tests/test_gpu_parity.py -q -x -k orderings
} > $out/log.txt 2>&1
grep -E "passed|failed|Error" $out/log.txt | head
