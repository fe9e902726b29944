#!/bin/bash
# tools/device_code_hash.sh [objdir] -- sha256 of the gfx950 code object embedded in every object file of the product library, and of the
# disassembled text of every kernel in it.  Two builds that print the same table run the same device code: host-side changes (and new
# translation units) can be checked against a tested build without a GPU.
#   bash tools/device_code_hash.sh > /tmp/before.txt ; <change, make> ; bash tools/device_code_hash.sh | diff /tmp/before.txt -
OBJ=${1:-$(dirname $0)/../gtsam_amd/lib/obj}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for o in $OBJ/*.o; do
  n=$(basename $o .o)
  $B/llvm-objcopy --dump-section .hip_fatbin=$T/$n.fat $o $T/$n.dummy 2>/dev/null || { echo "$n: no device code"; continue; }
  $B/clang-offload-bundler --unbundle --type=o --input=$T/$n.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$n.co 2>/dev/null || { echo "$n: no gfx950 bundle"; continue; }
  # the text section only: the code object's notes carry nothing time-dependent, but symbol order may change with unrelated additions
  $B/llvm-objdump -d --no-show-raw-insn $T/$n.co 2>/dev/null | grep -v "file format" | sed 's/^ *[0-9a-f]*://' > $T/$n.dis
  echo "$n: text $(sha256sum < $T/$n.dis | cut -c1-16) kernels $(grep -c '^[0-9a-f]* <.*>:$' $T/$n.dis)"
done
rm -rf $T
