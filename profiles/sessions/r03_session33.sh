# Round-3 session 33: CC tests, timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s33
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests -m gpu -q -x -k "cc or CC or fuzz or shapes or dropin" > $O/pytest.txt 2>&1; grep "passed\|failed" $O/pytest.txt | tail -2
for sc in 20 22 24 26 27; do timeout 300 python3 tools/cc_notorch.py $sc 2>&1 | grep "RMAT" | head -2 >> $O/cc.txt; done
cat $O/cc.txt
