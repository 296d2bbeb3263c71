# Round-6 session 50: heavy-row items of the triangle count kept in the plan; the BFS closing pass's level choice as bit-field blends: tests, times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s50
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs or tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/tc_knob_ab.py 23 8 "" > $O/r23.txt 2>&1; tail -3 $O/r23.txt | head -2
timeout 600 python3 tools/tc_knob_ab.py orkut 8 "" > $O/orkut.txt 2>&1; tail -3 $O/orkut.txt | head -2
timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs27.txt 2>&1; grep -E "^round|distances of" $O/bfs27.txt
