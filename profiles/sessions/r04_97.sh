# Round-4 session 97: TC: the hash-set kernel's waves take core items when their rows are done (GDN_TC_CORE_STEAL 1 / 0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s97
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,8192,12288,16384
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for st in 1 0; do
export GDN_TC_CORE_STEAL=$st
timeout 900 python3 tools/tc_core_ab.py 23 6 > $O/run23_$st.txt 2>&1
echo "steal $st"; grep RMAT $O/run23_$st.txt | tail -4; grep "same total" $O/run23_$st.txt
done
unset GDN_TC_CORE_STEAL
GDN_TC_CORE_ASYNC=0 timeout 900 python3 tools/tc_core_ab.py 23 6 > $O/run23_serial.txt 2>&1; echo "behind on the null stream"; grep RMAT $O/run23_serial.txt | tail -4
for s in 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -4; grep "same total" $O/run$s.txt; done
