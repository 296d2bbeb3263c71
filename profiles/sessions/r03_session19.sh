# Round-3 session 19: hub heads of the bottom-up BFS step: tests, traces with / without, the bottom-up switch re-tuned
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s19
mkdir -p $O; rm -f $O/bfs.txt $O/bfs_small.txt
timeout 900 python3 -m pytest tests -m gpu -q -x -k "bfs or BFS or fuzz or dropin" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for cfg in "GDN_BFS_HUB_HEADS=0" "GDN_BFS_HUB_MIN=0" "GDN_BFS_HUB_HEADS=1"; do
  echo "=== $cfg" >> $O/bfs.txt
  env GDN_BFS_TRACE=1 $cfg timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT\|bottom-up\|binned\|dense\|plan:" | tail -40 >> $O/bfs.txt
done
grep "===\|BFS RMAT" $O/bfs.txt
for sc in 24 26; do
for cfg in "GDN_BFS_HUB_HEADS=0" "GDN_BFS_HUB_HEADS=1"; do
  echo "=== RMAT-$sc $cfg" >> $O/bfs_small.txt
  env $cfg timeout 600 python3 tools/bfs_notorch.py $sc 2>&1 | grep "BFS RMAT" >> $O/bfs_small.txt
done
done
cat $O/bfs_small.txt
