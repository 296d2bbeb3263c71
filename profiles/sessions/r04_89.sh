# Round-4 session 89: TC core rows through buffer descriptors, loads pipelined: tests, A/B over K and the pair-test limit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s89
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for sm in 32 2 64 8; do
export GDN_TC_CORE_SMALL=$sm
timeout 900 rocprofv3 --kernel-trace -d $O/prof$sm -o tc -- python3 tools/tc_core_ab.py 23 4 > $O/run$sm.txt 2>&1
echo "small $sm"; grep RMAT $O/run$sm.txt | tail -4; grep "same total" $O/run$sm.txt
done
