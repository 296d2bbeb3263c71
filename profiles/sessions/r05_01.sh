# Round-5 session 1: smoke; the new oracle-parity tests (reference-sum mode, RMAT-24 summation order, RMAT-27 BFS x3 + PR converged
# vs the OpenMP oracle); the counter list of this box; placement probe with addresses
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s01
mkdir -p $O; rm -rf $O/*
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_sum" -s > $O/t_refsum.txt 2>&1; tail -3 $O/t_refsum.txt
timeout 900 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "summation_order" -s > $O/t_sumorder.txt 2>&1; grep -E "PR RMAT|hub rows|passed|failed|Error|assert" $O/t_sumorder.txt | head -20
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "oracle" -s --durations=5 > $O/t_fullsize.txt 2>&1; grep -E "BFS RMAT|PR RMAT|GDN_PR_SUM|passed|failed|Error|assert|s call" $O/t_fullsize.txt | head -30
rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt
GDN_PR_PLACE=0 GDN_PR_PLACE_TRACE=1 timeout 600 python3 tools/pr_place_probe.py 27 2 > $O/place_probe.txt 2>&1; tail -16 $O/place_probe.txt
nproc; free -g | head -2
