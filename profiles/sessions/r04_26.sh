# Round-4 session 26: SSSP solve time against the bucket width (lattice, uniform, R-MAT)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s26
mkdir -p $O; rm -rf $O/*
timeout 900 python3 tools/sssp_delta_sweep.py grid 4096 > $O/grid.txt 2>&1; cat $O/grid.txt | tail -9
timeout 600 python3 tools/sssp_delta_sweep.py uniform 23 8 > $O/uniform.txt 2>&1; cat $O/uniform.txt | tail -9
timeout 600 python3 tools/sssp_delta_sweep.py rmat 22 > $O/rmat.txt 2>&1; cat $O/rmat.txt | tail -9
