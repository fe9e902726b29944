cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-baseline off --skip-dense-roofline > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-200
DB=$(find /tmp/prof -name "*.db" | head -1); echo $DB
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py $DB 40 4
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py $DB 100 2
