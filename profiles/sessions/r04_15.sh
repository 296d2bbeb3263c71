# Round-4 session 15: CC without the reverse graph (outside-c filter), SSSP prep after the unrolled tile pass
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s15
mkdir -p $O; rm -rf $O/*
python3 tools/cc_notorch.py 24 > $O/cc.txt 2>&1; cat $O/cc.txt
GDN_CC_OUTSIDE=0 python3 tools/cc_notorch.py 24 2>&1 | grep 'out-edges' 
python3 tools/sssp_prep.py 24 > $O/sssp_prep.txt 2>&1; grep scale $O/sssp_prep.txt
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cc or sssp or fuzz" > $O/pytest.txt 2>&1; grep -E 'FAILED|passed|failed|Error' $O/pytest.txt | head
