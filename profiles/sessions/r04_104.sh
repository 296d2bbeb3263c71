# Round-4 session 104: the hash-set kernel requests the first neighbours' bounds before it builds the set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s104
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,12288,16384
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for s in 23 21 22 24; do timeout 900 python3 tools/tc_core_ab.py $s 6 > $O/run$s.txt 2>&1; grep RMAT $O/run$s.txt | tail -3; grep "same total" $O/run$s.txt; done
for f in a u v; do GDN_TC_FORM=$f timeout 600 python3 tools/tc_notorch.py 21 3 2>&1 | tail -1; done
