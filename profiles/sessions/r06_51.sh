# Round-6 session 51: with the items in the plan the hash-set kernel starts FIRST (its four waves per SIMD at 101 registers leave room for one core wave, not two): three chunks per step (93 registers) against four
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s51
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base unr3; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  echo "== $v"
  for g in 23 orkut 21 22 24; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "" "GDN_TC_CORE_WGS=3" > $O/${g}_$v.txt 2>&1; tail -3 $O/${g}_$v.txt | head -2; done
done
