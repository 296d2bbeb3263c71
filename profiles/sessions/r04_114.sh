# Round-4 session 114: smoke + TC / ingest tests on the final binary (after comment-only edits)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s114
mkdir -p $O; rm -rf $O/*
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_ingest.py tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu -k "tc or triangle or builder" -p no:cacheprovider 2>&1 | tail -1
TC_AB_CORES=0,16384 timeout 600 python3 tools/tc_core_ab.py 23 4 2>&1 | grep RMAT | tail -2
