# Round-6 session 56: the core's size again under the new sharing of a CU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s56
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 600 python3 tools/tc_knob_ab.py 23 6 "" "GDN_TC_CORE=16384" "GDN_TC_CORE=8192" > $O/23.txt 2>&1; tail -4 $O/23.txt | head -3
timeout 600 python3 tools/tc_knob_ab.py 22 6 "" "GDN_TC_CORE=16384" "GDN_TC_CORE=8192" > $O/22.txt 2>&1; tail -4 $O/22.txt | head -3
timeout 600 python3 tools/tc_knob_ab.py 21 6 "" "GDN_TC_CORE=12288" "GDN_TC_CORE=4096" > $O/21.txt 2>&1; tail -4 $O/21.txt | head -3
timeout 600 python3 tools/tc_knob_ab.py 20 6 "" "GDN_TC_CORE=12288" "GDN_TC_CORE=4096" > $O/20.txt 2>&1; tail -4 $O/20.txt | head -3
timeout 600 python3 tools/tc_knob_ab.py orkut 6 "" "GDN_TC_CORE=12288" "GDN_TC_CORE=4096" > $O/orkut.txt 2>&1; tail -4 $O/orkut.txt | head -3
timeout 600 python3 tools/tc_knob_ab.py 24 4 "" "GDN_TC_CORE=12288" > $O/24.txt 2>&1; tail -3 $O/24.txt | head -2
