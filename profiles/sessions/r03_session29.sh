# Round-3 session 29: the placement search in the FIRST processes of a fresh box (what the driver's bench meets)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s29
mkdir -p $O; rm -rf $O/*
for i in 1 2 3; do
  echo "=== process $i" >> $O/first.txt
  env GDN_PR_PLACE_TRACE=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_$i.json 2> $O/bench_$i.err
  grep "pr place" $O/bench_$i.err >> $O/first.txt
  python3 -c "
import json;d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]);print('bench', d['ms_per_step'], d['roofline']['frac'], 'plan_build_s', d['config']['plan_build_s'])" >> $O/first.txt
done
grep -v " try " $O/first.txt
