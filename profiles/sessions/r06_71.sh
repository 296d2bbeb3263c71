# Round-6 session 71: core kernel with two rows per group at K = 12288 under the 64-register bound as the default: TC tests, every graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s71
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for g in 23 orkut 21 22 24; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/$g.txt 2>&1; tail -2 $O/$g.txt | head -1; done
