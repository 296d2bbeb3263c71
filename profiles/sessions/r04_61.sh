# Round-4 session 61: wave-form bottom-up kernel with 512-thread workgroups (24 waves per CU instead of 16)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== 512 threads x 768"; GARDENIA_HIP_LIB=gardenia_amd/lib/var_w512/libgardenia_hip.so timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"
echo "== default 256 x 1024"; timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"
