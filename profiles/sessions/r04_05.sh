# Round-4 session 5: scratch from the stream-ordered pool -- is the 1.6 s stall gone?  Four fresh bench processes + tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s05
mkdir -p $O; rm -rf $O/*
for i in 1 2 3 4; do
  P=0; [ $i -ge 3 ] && P=3
  GDN_PB_TRACE=1 GDN_PR_PLACE=$P python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_$i.json 2> $O/bench_$i.log
  python3 -c "
import json;d=json.load(open('$O/bench_$i.json'));print('run $i place $P', d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_parts'],'plan',d['config']['plan_build_s'],'graph',d['graph_build_s'])"
  grep 'pb_build_tiered\] edges' $O/bench_$i.log | sed 's/.*; //'
done
grep 'pb_build' $O/bench_1.log | head -14
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt
GDN_PB_TRACE=1 timeout 300 python3 tools/pr_oneshot.py 22 > $O/pr_oneshot.txt 2>&1; grep -v '^\[pb' $O/pr_oneshot.txt | tail -3; grep 'edges' $O/pr_oneshot.txt | tail -1
