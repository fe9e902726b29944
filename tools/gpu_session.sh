out=gpurun_out/$1; mkdir -p $out
{
for w in ladybug1723 venice1778; do
  echo "== $w sorted"; BENCH_SORT_LANDMARKS=1 timeout 600 python bench.py --cpu-baseline off --skip-dense-roofline --workload $w 2>&1 | grep -v amdgpu | tail -5
done
} > $out/log.txt 2>&1
