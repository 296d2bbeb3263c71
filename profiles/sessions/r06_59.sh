# Round-6 session 59: the chunk position's clamp as one v_min: TC tests, times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s59
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for g in 23 orkut 21 22 24; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/$g.txt 2>&1; tail -3 $O/$g.txt | head -2; done
