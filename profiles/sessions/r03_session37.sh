# Round-3 session 37: dispatches of one SSSP solve and one BFS after the per-workgroup closing flushes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s37
mkdir -p $O; rm -rf $O/*
REPS=3 rocprofv3 --kernel-trace --output-format csv -d $O/sssp -o sssp -- python3 tools/sssp_trace.py 24 16 rand plan > $O/sssp.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/bfs -o bfs -- python3 tools/bfs_notorch.py 27 > $O/bfs.log 2>&1
grep "RMAT" $O/sssp.log $O/bfs.log
