# Round-5 session 11: run-time knobs of the SSSP schedule (RMAT-24, U[1,255], delta 16) on one plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s11
mkdir -p $O; rm -rf $O/*
timeout 600 python3 tools/sssp_knob_sweep.py 24 "" "GDN_SSSP_DENSE_IN=12" "GDN_SSSP_DENSE_IN=48" "GDN_SSSP_DENSE_IN=96" "GDN_SSSP_DENSE_OUT=4" "GDN_SSSP_DENSE_OUT=16" "GDN_SSSP_DENSE_OUT=32" "GDN_SSSP_DENSE_PRE=96" "GDN_SSSP_DENSE_PRE=384" "delta=8" "delta=32" "delta=64" "GDN_SSSP_ADAPT=0" "GDN_SSSP_DENSE_IN=48,GDN_SSSP_DENSE_OUT=16" "" > $O/sweep24.txt 2>&1; cat $O/sweep24.txt
GDN_SSSP_TRACE=1 timeout 300 python3 tools/sssp_knob_sweep.py 24 "" 2>&1 | grep "\[sssp\]" | tail -22 > $O/trace24.txt; cat $O/trace24.txt
