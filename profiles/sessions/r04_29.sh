# Round-4 session 29: adaptive bucket width behind a run of light buckets: parity, sweeps, RMAT-24 A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s29
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
timeout 900 python3 tools/sssp_delta_sweep.py grid 4096 > $O/grid.txt 2>&1; cat $O/grid.txt | tail -8
timeout 600 python3 tools/sssp_delta_sweep.py uniform 23 8 > $O/uniform.txt 2>&1; cat $O/uniform.txt | tail -8
timeout 600 python3 tools/sssp_delta_sweep.py rmat 22 > $O/rmat.txt 2>&1; cat $O/rmat.txt | tail -8
timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_ADAPT 0 1 24 3 > $O/ab.txt 2>&1; grep -v round $O/ab.txt | tail -6
for L in 32768 524288; do GDN_SSSP_LIGHT_COOP=$L timeout 300 python3 tools/sssp_delta_sweep.py grid 4096 2>&1 | head -1; done
for L in 1024 16384; do GDN_SSSP_LIGHT_SMALL=$L timeout 300 python3 tools/sssp_delta_sweep.py grid 4096 2>&1 | head -1; done
