# Round-3 session 21: full per-level trace of the BFS at RMAT-27 (all engines)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s21
mkdir -p $O; rm -f $O/*.txt
env GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 > $O/trace27.txt 2>&1
cat $O/trace27.txt | head -60
