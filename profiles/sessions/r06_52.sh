# Round-6 session 52: hash-set kernel held to 96 registers (dynamic LDS + five waves per SIMD asked for) against 101; core workgroups per CU 2 / 3 / 4, the hash-set kernel starting first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s52
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base dyn5; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  echo "== $v"
  for g in 23 orkut 21 24; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "GDN_TC_CORE_WGS=2" "GDN_TC_CORE_WGS=3" "GDN_TC_CORE_WGS=4" > $O/${g}_$v.txt 2>&1; tail -4 $O/${g}_$v.txt | head -3; done
done
