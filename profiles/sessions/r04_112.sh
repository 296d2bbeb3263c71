# Round-4 session 112: the core's bit matrix stored as a triangle (16.8 MB instead of 32 at K = 16384)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s112
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,12288,16384
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for s in 23 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -3; grep "same total" $O/run$s.txt; done
