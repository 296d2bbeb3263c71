# Round-4 session 17: CC out-edges-only with the edge-parallel target sweep; SSSP prep phases
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s17
mkdir -p $O; rm -rf $O/*
python3 tools/cc_notorch.py 24 > $O/cc.txt 2>&1; cat $O/cc.txt
GDN_PB_TRACE=1 python3 tools/sssp_prep.py 24 > $O/sssp_prep.txt 2>&1; grep 'scale\|pb_build_out' $O/sssp_prep.txt | sed -n '10,20p;/u1_255/p' | head -24
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cc or fuzz" > $O/pytest.txt 2>&1; grep -E 'FAILED|passed|failed|Error' $O/pytest.txt | head
