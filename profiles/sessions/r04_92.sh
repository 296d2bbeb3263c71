# Round-4 session 92: TC core: items partitioned by class (rows in order), low-priority stream queued behind the hash-set kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s92
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o tc -- python3 tools/tc_core_ab.py 23 6 > $O/run23.txt 2>&1
grep RMAT $O/run23.txt | tail -4; grep "same total" $O/run23.txt
GDN_TC_CORE_ASYNC=0 timeout 900 python3 tools/tc_core_ab.py 23 4 > $O/run23_serial.txt 2>&1
echo serial; grep RMAT $O/run23_serial.txt | tail -4
for s in 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 4 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -4; grep "same total" $O/run$s.txt; done
