# Round-4 session 60: grid of the wave-form bottom-up kernel (1024 = resident, 2048 default, 4096)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in g1024 g4096; do echo "== $v"; GARDENIA_HIP_LIB=gardenia_amd/lib/var_$v/libgardenia_hip.so timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"; done
echo "== default"; timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"
