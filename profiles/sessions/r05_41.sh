# Round-5 session 41: TC forward count with the walks' bounds packed per edge (GDN_TC_NBOUND: no gather of the neighbour's row offsets):
# parity, then the A/B at RMAT-23 / 22 / 24 with K = 16384 and 12288
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tc_" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "triangle or config4" 2>&1 | tail -3
timeout 900 python3 tools/tc_knob_ab.py 23 8 "GDN_TC_NBOUND=0" "" "GDN_TC_NBOUND=0,GDN_TC_CORE=12288" "GDN_TC_CORE=12288" "GDN_TC_CORE=8192" "GDN_TC_NBOUND=0,GDN_TC_CORE=0" "GDN_TC_CORE=0" 2>&1 | tee gpurun_out/r05s41_tc23.txt
timeout 900 python3 tools/tc_knob_ab.py 22 8 "GDN_TC_NBOUND=0" "GDN_TC_NBOUND=1" "GDN_TC_NBOUND=1,GDN_TC_CORE=12288" 2>&1 | tee gpurun_out/r05s41_tc22.txt
timeout 900 python3 tools/tc_knob_ab.py 24 6 "GDN_TC_NBOUND=0" "" "GDN_TC_CORE=12288" 2>&1 | tee gpurun_out/r05s41_tc24.txt
