# Round-4 session 4: partition kernels at 512 threads (no spills), exact-degree tier thresholds, phase times of the build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s04
mkdir -p $O; rm -rf $O/*
GDN_PB_TRACE=1 GDN_PR_PLACE=0 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_noplace.json 2> $O/bench_noplace.log
python3 -c "
import json;d=json.load(open('$O/bench_noplace.json'));print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_parts'],d['config']['plan_build_s'])"
grep 'pb_build' $O/bench_noplace.log | head -30
timeout 600 python3 -m pytest tests -m gpu -x -q -k "pr" > $O/pytest_pr.txt 2>&1; tail -3 $O/pytest_pr.txt
GDN_PB_TRACE=1 timeout 300 python3 tools/pr_oneshot.py 22 > $O/pr_oneshot.txt 2>&1; grep -v '^\[pb_order' $O/pr_oneshot.txt | tail -45
GDN_PB_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_place.json 2> $O/bench_place.log
python3 -c "
import json;d=json.load(open('$O/bench_place.json'));print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_parts'],d['config']['plan_build_s'])"
grep 'pb_build' $O/bench_place.log | head -30
