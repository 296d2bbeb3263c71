# Round-5 session 42: the core's size K by graph size on the final binary (RMAT-21 / 22 / 23 and the Orkut-like stand-in)
mkdir -p gpurun_out
for g in 21 22 23 orkut; do
  timeout 900 python3 tools/tc_knob_ab.py $g 8 "GDN_TC_CORE=8192" "GDN_TC_CORE=12288" "GDN_TC_CORE=16384" 2>&1 | tee -a gpurun_out/r05s42_tc_k.txt
done
