# Round-5 session 12: the TC core size and its workgroups per CU swept again on the final kernels (RMAT-23)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s12
mkdir -p $O; rm -rf $O/*
TC_AB_CORES=8192,12288,16384 timeout 400 python3 tools/tc_core_ab.py 23 6 > $O/k.txt 2>&1; cat $O/k.txt | grep -v amdgpu
for w in 1 3; do GDN_TC_CORE_WGS=$w TC_AB_CORES=12288,16384 timeout 300 python3 tools/tc_core_ab.py 23 6 2>&1 | grep -v amdgpu | sed "s/^/WGS=$w /" >> $O/wgs.txt; done; cat $O/wgs.txt
for l in 64 256; do GDN_TC_LIGHT=$l TC_AB_CORES=16384 timeout 300 python3 tools/tc_core_ab.py 23 6 2>&1 | grep -v amdgpu | sed "s/^/LIGHT=$l /" >> $O/light.txt; done; cat $O/light.txt
