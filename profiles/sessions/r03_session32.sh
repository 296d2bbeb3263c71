# Round-3 session 32: the counter read-backs through the pinned mailbox against copy + stream synchronisation
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s32
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests -m gpu -q -x -k "bfs or BFS or fuzz or sssp or SSSP" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2; do
for cfg in "GDN_MAILBOX=0" "GDN_MAILBOX=1"; do
  for sc in 24 27; do
    echo "=== BFS RMAT-$sc $cfg" >> $O/t.txt
    env $cfg timeout 600 python3 tools/bfs_notorch.py $sc 2>&1 | grep "BFS RMAT" | awk '{print $5}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt
  done
  echo "=== SSSP RMAT-24 $cfg" >> $O/t.txt
  env $cfg REPS=6 timeout 300 python3 tools/sssp_trace.py 24 16 rand plan 2>&1 | grep "RMAT" | awk '{print $6}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt
done
done
cat $O/t.txt
