# Round-4 session 57: the slow source's heavy level on the bottom-up step again (GDN_BFS_BU_EDGE_DIV), now with the wave form + lone-head flag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s57
mkdir -p $O; rm -rf $O/*
for D in 3 6 8 12; do echo "== GDN_BFS_BU_EDGE_DIV=$D"; GDN_BFS_BU_EDGE_DIV=$D GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep -E "level 3|level 4|BFS RMAT" | head -12; done > $O/bu_div.txt 2>&1; cat $O/bu_div.txt
