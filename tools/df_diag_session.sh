mkdir -p gpurun_out/r3j
run() { echo "== $*" ; timeout 500 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
run python tools/df_contention_diag.py 200 3 close sphere
run python tools/df_contention_diag.py 150 3 interleave sphere
run python -m pytest tests/test_gpu_dataflow_protocol.py -x -q
run python bench.py --cpu-baseline off --skip-dense-roofline
} > gpurun_out/r3j/diag.log 2>&1
cut -c1-1700 gpurun_out/r3j/diag.log
