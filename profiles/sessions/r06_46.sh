# Round-6 session 46: top-down levels without the read in front of the atomic (graphs without hubs, early levels): tests, uniform 2^26, RMAT-27 unchanged
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s46
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/bfs_runs.py 27 3 1 > $O/bfs27.txt 2>&1; grep -E "^round" $O/bfs27.txt
export GDN_TEST_HOOKS=1
for b in "" 0 1; do
  if [ -z "$b" ]; then unset GDN_BFS_TD_BLIND; else export GDN_BFS_TD_BLIND=$b; fi
  echo "== GDN_BFS_TD_BLIND=$b"
  timeout 900 python3 tools/bfs_ab.py u26 "" > $O/u26_$b.txt 2>&1; grep -E "ms|top-down" $O/u26_$b.txt | head -12
done
