# Round-4 session 108: runs of empty rows filled by a grid (csr_from_keys): ingest tests, TC tests, forward plan build time, graph build times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s108
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_ingest.py tests/test_gpu_parity.py -q -x -m gpu -k "ingest or builder or symmetrize or tc or transpose or graph" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
export TC_AB_CORES=0,16384
for s in 21 23 24; do timeout 900 python3 tools/tc_core_ab.py $s 4 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -2; done
timeout 900 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-bfs > $O/bench.json 2> $O/bench.log; grep -o "built on device in [0-9.]* s" $O/bench.log | head -2; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('tc', d['tc']['ms']['median'], 'plan_build_s', d['tc']['plan_build_s'], 'orient_s', d['tc']['orient_s']); print('PR', d['ms_per_step'], d['roofline']['frac'])"
