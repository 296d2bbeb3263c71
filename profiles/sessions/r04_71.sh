# Round-4 session 71: medium-size parity sweep (tools/mid_sweep.py): seeded R-MAT 18 / 20 / 21 / 22, resident plans against the oracle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s71
mkdir -p $O; rm -rf $O/*
timeout 3000 python3 tools/mid_sweep.py 18,20,21 101,202,303 > $O/sweep.txt 2>&1; tail -12 $O/sweep.txt
timeout 1500 python3 tools/mid_sweep.py 22 404 >> $O/sweep.txt 2>&1; tail -2 $O/sweep.txt
