# Round-6 session 43: the closing pass of a BFS search as a plain fill (ablation 4: no bitmap read) -- what its reads and its logic cost
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s43
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_babl4/libgardenia_hip.so
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bfs_runs.py 27 4 1 > $O/bfs.txt 2>&1
grep -h "bfs_depth_finish" $O/trace/*/*_kernel_stats.csv | cut -d, -f1-4
