# Round-4 session 28: traces of the RMAT-24 U[1,255] solve with and without the adaptive bucket width
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s28
mkdir -p $O; rm -rf $O/*
GDN_SSSP_ADAPT=0 GDN_SSSP_TRACE=1 REPS=2 python3 tools/sssp_trace.py 24 16 rand > $O/trace_adapt0.txt 2>&1
GDN_SSSP_ADAPT=1 GDN_SSSP_TRACE=1 REPS=2 python3 tools/sssp_trace.py 24 16 rand > $O/trace_adapt1.txt 2>&1
grep "sssp\]" $O/trace_adapt0.txt | tail -22; echo ====; grep "sssp\]" $O/trace_adapt1.txt | tail -22
