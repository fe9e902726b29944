out=gpurun_out/$1; mkdir -p $out
{
for pf in "6 3" "4 3" "8 3" "6 2" "6 4" "4 2" "8 4" "12 3" "3 2"; do
  set -- $pf
  echo "== PIECE $1 FINAL $2"
  GTG_DF_PIECE=$1 GTG_DF_FINAL=$2 timeout 300 python bench.py --cpu-baseline off --skip-dense-roofline --steps 12 2>/dev/null | grep '^{'
done
for g in 240 232 216; do echo "== GRID $g"; GTG_DF_GRID=$g timeout 300 python bench.py --cpu-baseline off --skip-dense-roofline --steps 12 2>/dev/null | grep '^{'; done
} > $out/log.txt 2>&1
