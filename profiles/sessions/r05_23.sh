# Round-5 session 23: the TC core kernel split over the XCDs (GDN_TC_CORE_XCD = length classes every XCD walks for its own rows of
# the bit matrix -- an experiment: the knob was removed again after this session, profiles/r05_tc_xcd_split.txt); CC with the reverse graph built inside the call; parity first, then the A/B at RMAT-23 / 22 / 24
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tc_ or cc_" 2>&1 | tail -5
timeout 900 python3 tools/tc_knob_ab.py 23 8 "" "GDN_TC_CORE_XCD=1" "GDN_TC_CORE_XCD=2" "GDN_TC_CORE_XCD=3" "GDN_TC_CORE_XCD=4" 2>&1 | tee gpurun_out/r05s23_tc23.txt
timeout 900 python3 tools/tc_knob_ab.py 23 8 "GDN_TC_CORE_ASYNC=0" "GDN_TC_CORE_ASYNC=0,GDN_TC_CORE_XCD=2" "GDN_TC_CORE_ASYNC=0,GDN_TC_CORE_XCD=4" 2>&1 | tee gpurun_out/r05s23_tc23_alone.txt
timeout 900 python3 tools/tc_knob_ab.py 22 8 "" "GDN_TC_CORE_XCD=2" "GDN_TC_CORE_XCD=3" 2>&1 | tee gpurun_out/r05s23_tc22.txt
