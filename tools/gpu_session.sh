out=gpurun_out/$1; mkdir -p $out
{
echo "== full gpu suite"; timeout 1400 python -m pytest tests/ -q -m gpu --durations=12 2>&1 | grep -v amdgpu.ids | tail -60
} > $out/log.txt 2>&1
tail -62 $out/log.txt
