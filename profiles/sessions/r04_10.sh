# Round-4 session 10: scratch CACHE (hipMalloc blocks kept by the process): fuzz sweeps, whole suite, stall check
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s10
mkdir -p $O; rm -rf $O/*
GDN_SCRATCH_POISON=1 timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_fuzz_poison.txt 2>&1; tail -3 $O/pytest_fuzz_poison.txt
timeout 1200 python3 -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
for i in 1 2 3 4; do
  GDN_PB_TRACE=1 GDN_PR_PLACE=0 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_$i.json 2> $O/bench_$i.log
  python3 -c "
import json;d=json.load(open('$O/bench_$i.json'));print('run $i', d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_parts'],'plan',d['config']['plan_build_s'],'graph',d['graph_build_s'])"
  grep 'pb_build_tiered\] edges' $O/bench_$i.log | sed 's/.*; //'
done
GDN_PB_TRACE=1 timeout 300 python3 tools/pr_oneshot.py 22 > $O/pr_oneshot.txt 2>&1; grep -v '^\[pb' $O/pr_oneshot.txt | tail -3; grep 'edges' $O/pr_oneshot.txt | tail -1
