# Round-5 session 19: the edge share from which the bottom-up step takes a level on skewed graphs: 1/8 (default) against 1/10, 1/12, 1/16, 1/24
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s19
mkdir -p $O; rm -rf $O/*
for s in 27 25 24 22; do
  timeout 400 python3 tools/bfs_ab.py $s "" "GDN_BFS_BU_EDGE_DIV=10" "GDN_BFS_BU_EDGE_DIV=12" "GDN_BFS_BU_EDGE_DIV=16" "GDN_BFS_BU_EDGE_DIV=24" "" 2> $O/trace$s.txt | sed "s/^/RMAT-$s /" | tee -a $O/ab.txt
done
