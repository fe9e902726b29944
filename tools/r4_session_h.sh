#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
for cfg in "4 3" "8 3" "8 4"; do
  set -- $cfg
  GTG_DF_SLOTS=$1 GTG_ND_DEPTH=$2 timeout 200 python tools/df_trace_pose.py w20000 > $out/trace_w20000_s$1_d$2.json 2> $out/trace_w20000_s$1_d$2.err
  tail -c 2500 $out/trace_w20000_s$1_d$2.json; echo
done
