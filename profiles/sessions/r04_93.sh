# Round-4 session 93: TC core: two rows per group from K = 8192 on (registers beside the hash-set kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s93
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o tc -- python3 tools/tc_core_ab.py 23 6 > $O/run23.txt 2>&1
grep RMAT $O/run23.txt | tail -4; grep "same total" $O/run23.txt
for s in 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -4; grep "same total" $O/run$s.txt; done
