# Round-5 session 51: a map of the device memory by allocation order: plain writes, reads and a 4096-stream scatter per 2 GB chunk (tools/hbm_region_map.py)
mkdir -p gpurun_out
timeout 1500 python3 tools/hbm_region_map.py 2 5 8 > gpurun_out/r05s51_map.txt 2>&1; head -3 gpurun_out/r05s51_map.txt; tail -5 gpurun_out/r05s51_map.txt
