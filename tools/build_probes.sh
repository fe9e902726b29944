#!/bin/sh
# Prebuilds the stand-alone GPU probes (hipcc cross-compiles without a GPU) as tools/*.bin: git-ignored, but they travel to
# the GPU box with the snapshot, so that no GPU-minute is spent compiling.   sh tools/build_probes.sh
set -e
cd "$(dirname "$0")/.."
for p in graph_probe contention_probe potrf_chain_probe potrf_micro_probe mfma_f64_peak mfma_f64_occupancy fma_f64_peak; do
  [ -f tools/$p.hip ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/$p.hip -o tools/$p.bin 2>/dev/null && echo built tools/$p.bin
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/device_analysis/proto.hip -o tools/device_analysis_proto.bin 2>/dev/null && echo built tools/device_analysis_proto.bin
