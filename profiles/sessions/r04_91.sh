# Round-4 session 91: TC core: items sorted by length and handed out by class, core kernel beside the hash-set kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s91
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for as in 1 0; do
export GDN_TC_CORE_ASYNC=$as
timeout 900 rocprofv3 --kernel-trace -d $O/prof$as -o tc -- python3 tools/tc_core_ab.py 23 4 > $O/run$as.txt 2>&1
echo "async $as"; grep RMAT $O/run$as.txt | tail -4; grep "same total" $O/run$as.txt
done
