# Round-6 session 69: the core kernel at K = 12288 with two rows per group (70 registers: one wave per SIMD beside the hash-set kernel; held to 64 with spills: two)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s69
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base g3 g3w8 base; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  timeout 600 python3 tools/tc_knob_ab.py 23 8 "" "GDN_TC_CORE_ASYNC=0" > $O/23_$v.txt 2>&1; echo "$v: $(tail -3 $O/23_$v.txt | head -2 | cut -c1-125 | tr '\n' '|')"
done
