# Round-4 session 44: the non-R-MAT shapes (lattice, uniform, small world; 67 M edges) through every solver on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s44
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 tools/shapes.py large $O/shapes_large.json > $O/shapes.log 2>&1; tail -5 $O/shapes.log
