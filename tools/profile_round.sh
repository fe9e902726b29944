#!/bin/bash
# tools/profile_round.sh <tag>  -- run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + HBM PMC
# passes (separate runs, as MI355X_MICROARCH.md prescribes) for the default bench workload.  Writes gpurun_out/<tag>_*.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.log
tail -c 600 $OUT/${TAG}_bench.json
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
# (kernel tracing does not serialise dispatches: the two cooperating kernels of the dataflow Cholesky run as in production; should a
# tool version serialise them, the factorisation reports a wait timeout instead of hanging and the single-kernel form is profiled)
# (rocprofv3 of ROCm 7.0.2 on the GPU box has been seen to segfault in its own finalisation AFTER writing the database: the exit code
# is not the criterion, the database with the bench's JSON line in the log is)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 4 --warmup 1 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $OUT/${TAG}_trace.log 2>&1
if ! grep -q '"metric"' $OUT/${TAG}_trace.log || [ -z "$(find /tmp/prof_s -name '*.db' | head -1)" ]; then
  rm -rf /tmp/prof_s; GTG_DF_SINGLE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $REPO/bench.py --steps 4 --warmup 1 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $OUT/${TAG}_trace.log 2>&1
fi
python $REPO/tools/rocprof_top.py $(find /tmp/prof_s -name "*.db" | head -1) $OUT/${TAG}_kernel_stats.csv | head -12
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $OUT/${TAG}_fetch.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_f -name "*.db" | head -1) $OUT/${TAG}_pmc_fetch_size.csv > /dev/null
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $OUT/${TAG}_write.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_w -name "*.db" | head -1) $OUT/${TAG}_pmc_write_size.csv > /dev/null
python $REPO/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_pmc_cholesky_traffic.json
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
