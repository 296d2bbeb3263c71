# Round-4 session 53: timing-only ablation: the wave-form bottom-up step without its in-neighbour scans (wrong depths)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s53
mkdir -p $O; rm -rf $O/*
GDN_BFS_TRACE=1 GARDENIA_HIP_LIB=gardenia_amd/lib/var_noscan/libgardenia_hip.so timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep -E "bottom-up|BFS RMAT" | grep -v "hubs in" | head -8
