# Round-4 session 52: wave-form bottom-up step: rows per lane whose loads are in flight together (BFS_BU_UNR 2 / 4 / 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s52
mkdir -p $O; rm -rf $O/*
for v in bu2 bu8; do echo "== $v"; GARDENIA_HIP_LIB=gardenia_amd/lib/var_$v/libgardenia_hip.so timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"; done
echo "== default (4)"; timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"
