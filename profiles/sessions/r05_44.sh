# Round-5 session 44: the TC knobs again at the new default K (12288 at RMAT-23, 8192 on the Orkut-like stand-in)
mkdir -p gpurun_out
timeout 900 python3 tools/tc_knob_ab.py 23 8 "" "GDN_TC_CORE_WGS=3" "GDN_TC_CORE_WGS=1" "GDN_TC_LIGHT=64" "GDN_TC_LIGHT=256" "GDN_TC_CORE_SMALL=16" "GDN_TC_CORE_SMALL=64" 2>&1 | tee gpurun_out/r05s44_tc23.txt
timeout 900 python3 tools/tc_knob_ab.py orkut 8 "" "GDN_TC_CORE_WGS=3" "GDN_TC_LIGHT=64" "GDN_TC_LIGHT=256" "GDN_TC_CORE_SMALL=16" "GDN_TC_CORE=4096" 2>&1 | tee gpurun_out/r05s44_orkut.txt
