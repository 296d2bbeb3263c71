# Round-4 session 95: TC core: one row per group from K = 12288 on, workgroups per CU 2 / 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s95
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=8192,12288,16384
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for w in default 2 3 4; do
if [ $w = default ]; then unset GDN_TC_CORE_WGS; else export GDN_TC_CORE_WGS=$w; fi
timeout 900 python3 tools/tc_core_ab.py 23 6 > $O/run23_$w.txt 2>&1
echo "wgs $w"; grep RMAT $O/run23_$w.txt | tail -3; grep "same total" $O/run23_$w.txt
done
unset GDN_TC_CORE_WGS
for s in 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -3; grep "same total" $O/run$s.txt; done
