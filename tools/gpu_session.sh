#!/bin/bash
# tools/gpu_session.sh [tag] -- one gpurun call's worth of checks in priority order (each step time-boxed, results under
# gpurun_out/<tag>_*): GPU test suite, fuzz parity against the oracle, bench line with the set-up breakdown, the multi-GPU
# exchange in both granularities on one GPU (two-shard tests), then the rocprofv3 / PMC passes of tools/profile_round.sh.
#   gpurun --timeout 600 -- 'bash tools/gpu_session.sh r02a'
TAG=${1:-session}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO; export TMPDIR=/tmp
(timeout 120 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > $OUT/${TAG}_gpu_tests.log; head -3 $OUT/${TAG}_gpu_tests.log
(timeout 120 python tools/gpu_fuzz.py --seeds 16 2>&1 | tail -70) > $OUT/${TAG}_gpu_fuzz.jsonl; tail -1 $OUT/${TAG}_gpu_fuzz.jsonl
GTG_DEBUG_TIMING=1 timeout 120 python bench.py --steps 8 --warmup 2 --cpu-baseline off --skip-dense-roofline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 400 $OUT/${TAG}_bench.json; grep "setup\]" $OUT/${TAG}_bench.err | tail -10 > $OUT/${TAG}_host_setup_breakdown.txt
(GTG_EXCHANGE_TILES=1 timeout 60 python -m pytest tests/test_gpu_sharding.py -q -p no:cacheprovider 2>&1 | tail -3) > $OUT/${TAG}_sharding_tiles.log
# stand-alone prototype of the device-side symbolic analysis against the host's lists (tools/device_analysis)
# (the probes are prebuilt by tools/build_probes.sh in the build container and travel with the snapshot)
(timeout 120 python tools/device_analysis/make_input.py ladybug1723 /tmp/da_l1723.bin && timeout 60 tools/device_analysis_proto.bin /tmp/da_l1723.bin) > $OUT/${TAG}_device_analysis_proto.log 2>&1
tail -2 $OUT/${TAG}_device_analysis_proto.log
# explicit hipGraph against stream / event issue on the shape of the Cholesky schedule (tools/graph_probe.hip)
(timeout 60 tools/graph_probe.bin 61 && timeout 60 tools/graph_probe.bin 239) > $OUT/${TAG}_graph_probe.jsonl 2>&1
tail -2 $OUT/${TAG}_graph_probe.jsonl
# chain kernel against a chip-filling bulk kernel, and the cheap remedies (tools/contention_probe.hip)
(timeout 60 tools/contention_probe.bin) > $OUT/${TAG}_contention_probe.json 2>&1
tail -1 $OUT/${TAG}_contention_probe.json
[ "$2" = "profile" ] && bash tools/profile_round.sh $TAG
true
