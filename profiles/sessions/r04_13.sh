# Round-4 session 13: SSSP's layout + tiers from the out-CSR builder
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s13
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests -m gpu -x -q -k "sssp or fuzz" > $O/pytest.txt 2>&1; grep -E 'FAILED|passed|failed|Error' $O/pytest.txt | head
GDN_PB_TRACE=1 GDN_SSSP_TRACE=1 python3 tools/sssp_prep.py 24 > $O/sssp_prep.txt 2>&1; grep -v '^\[sssp\] *[0-9]' $O/sssp_prep.txt | grep 'scale\|pb_build_out\|plan:' | head -40
GDN_PB_BUILDER=old python3 tools/sssp_prep.py 24 2>&1 | grep scale
