out=gpurun_out/$1; mkdir -p $out
timeout 100 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 200 $out/bench.json
