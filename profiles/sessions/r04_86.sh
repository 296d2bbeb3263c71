# Round-4 session 86: TC forward count with the core bit matrix: tests, then the A/B over the core size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s86
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "tc" -p no:cacheprovider > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python3 tools/tc_core_ab.py 21 6 > $O/ab21.txt 2>&1; cat $O/ab21.txt
timeout 900 python3 tools/tc_core_ab.py 23 8 > $O/ab23.txt 2>&1; cat $O/ab23.txt
