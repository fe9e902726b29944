out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_s
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 4 --warmup 1 --cpu-baseline off --skip-dense-roofline > $REPO/$out/trace.log 2>&1
python $REPO/tools/rocprof_top.py $(find /tmp/prof_s -name "*.db" | head -1) $REPO/gpurun_out/r03_kernel_stats.csv | head -8
cd $REPO
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r03_gpu_tests.log 2>&1
tail -4 gpurun_out/r03_gpu_tests.log
