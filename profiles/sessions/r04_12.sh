# Round-4 session 12: partition kernels ranked by LDS atomics (and the ballot form for comparison)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s12
mkdir -p $O; rm -rf $O/*
for st in 0 1; do
  E=""; [ $st = 1 ] && E="GDN_PT_STABLE=1"
  env $E GDN_PB_TRACE=1 GDN_PR_PLACE=0 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-extras --no-bfs > $O/bench_$st.json 2> $O/bench_$st.log
  python3 -c "
import json;d=json.load(open('$O/bench_$st.json'));print('stable $st', d['ms_per_step'],d['roofline']['frac'],'plan',d['config']['plan_build_s'], 'l1', d['pr_last_l1_change'])"
  grep 'pt_split\|pt_radix\|pt_keygen\|pt_tiles\|wall' $O/bench_$st.log | grep -v order
  env $E GDN_PB_TRACE=1 python3 tools/pr_oneshot.py 22 2>&1 | grep 'scale 22 pb\|pt_split\|pt_radix\|wall' | tail -4
done
timeout 600 python3 -m pytest tests -m gpu -x -q -k "pr or spmv or fuzz" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
