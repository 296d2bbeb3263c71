# Round-6 session 68: the wave bound the hash-set kernel is compiled for (4 / 5 / 6 per SIMD) and core workgroups per CU, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s68
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base wpe4 wpe6 wpe6u3; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for g in 23 orkut 21; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "" "GDN_TC_CORE_WGS=6" > $O/${g}_$v.txt 2>&1; echo "$v: $(tail -3 $O/${g}_$v.txt | head -2 | cut -c1-125 | tr '\n' '|')"; done
done
