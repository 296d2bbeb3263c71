# Round-3 session 35: dispatches of one BC solve (RMAT-24, resident plan) and of the delta PageRank
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s35
mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --output-format csv -d $O/bc -o bc -- python3 tools/bc_notorch.py 24 plan > $O/bc.log 2>&1
grep -v "^W2026\|^E2026" $O/bc.log | tail -5
rocprofv3 --kernel-trace --output-format csv -d $O/prd -o prd -- python3 tools/prdelta_notorch.py 25 > $O/prd.log 2>&1
grep -v "^W2026\|^E2026" $O/prd.log | tail -5
