# Round-6 session 72: the core of 12288 ranks (now two rows per group) against the sizes' defaults
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s72
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for g in 24 22 21 orkut; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "" "GDN_TC_CORE=12288" > $O/$g.txt 2>&1; tail -3 $O/$g.txt | head -2; done
