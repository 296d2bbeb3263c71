# Round-4 session 31: SSSP RMAT-24 with the light phases on the cooperative grid from the start (GDN_SSSP_COOP=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s31
mkdir -p $O; rm -rf $O/*
timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_COOP 0 1 24 3 > $O/ab.txt 2>&1; grep -v round $O/ab.txt | tail -6
GDN_SSSP_COOP=1 GDN_SSSP_TRACE=1 REPS=2 python3 tools/sssp_trace.py 24 16 rand > $O/trace_coop1.txt 2>&1; grep "sssp\]" $O/trace_coop1.txt | tail -14
