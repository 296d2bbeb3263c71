# Round-4 session 103: timeline of the three TC grids (count, core beside, core tail) at K = 16384 / 12288
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s103
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=16384,12288
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o tc -- python3 tools/tc_core_ab.py 23 3 > $O/run.txt 2>&1
grep RMAT $O/run.txt | tail -2
