# Round-4 session 19: the bottom-up step (head records) from a smaller frontier share on: GDN_BFS_BU_EDGE_DIV
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s19
mkdir -p $O; rm -rf $O/*
for div in 3 6 8 12 20; do
  echo "== GDN_BFS_BU_EDGE_DIV=$div"
  GDN_BFS_BU_EDGE_DIV=$div GDN_BFS_TRACE=1 python3 tools/bfs_notorch.py 27 2>&1 | grep 'BFS RMAT\|level 3\|level 4' | head -12
done > $O/bu_div.txt 2>&1
cat $O/bu_div.txt
