# Round-4 session 102: TC core: a tail grid behind the hash-set kernel on the null stream (GDN_TC_CORE_TAIL 6 / 0 / 4 / 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s102
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,8192,12288,16384
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for t in 6 0 4 8; do
export GDN_TC_CORE_TAIL=$t
timeout 900 python3 tools/tc_core_ab.py 23 6 > $O/run23_$t.txt 2>&1
echo "tail $t"; grep RMAT $O/run23_$t.txt | tail -4; grep "same total" $O/run23_$t.txt
done
unset GDN_TC_CORE_TAIL
for s in 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -4; grep "same total" $O/run$s.txt; done
