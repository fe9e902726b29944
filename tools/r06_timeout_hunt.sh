#!/bin/bash
# tools/r06_timeout_hunt.sh [tag] [runs] -- how often the first factorisations of a cold process run into a dependency-wait time-out, with the
# masked stream pair created on the helper thread (default) and by the first factorisation (GTG_SYNC_STREAM_PAIR=1), with / without parked streams
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r11a}; mkdir -p $out
N=${2:-12}
cd $REPO
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for cfg in default sync_pair no_parked both_off; do
  case $cfg in
    default) E="";;
    sync_pair) E="GTG_SYNC_STREAM_PAIR=1";;
    no_parked) E="GTG_NO_PARKED_STREAMS=1";;
    both_off) E="GTG_SYNC_STREAM_PAIR=1 GTG_NO_PARKED_STREAMS=1";;
  esac
  : > $out/$cfg.txt
  for i in $(seq $N); do
    env $E tests/_build/cold_start_probe /tmp/l1723.txt --prewarm >> $out/$cfg.txt 2>&1
    env $E tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 7 --warmup 0 >> $out/$cfg.bench.txt 2>> $out/$cfg.txt
  done
  echo "$cfg: time-outs $(grep -c 'dependency wait' $out/$cfg.txt) in $((2 * N)) processes; first try ms: $(grep 'gtg_try_lambda (first)' $out/$cfg.txt | awk '{print $4}' | tr '\n' ' ')"
done
