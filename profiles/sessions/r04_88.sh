# Round-4 session 88: TC core with batched item grabs: tests, A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s88
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o tc -- python3 tools/tc_core_ab.py 23 6 > $O/run.txt 2>&1
grep RMAT $O/run.txt; grep "same total" $O/run.txt
