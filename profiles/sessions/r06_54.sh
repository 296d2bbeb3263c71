# Round-6 session 54: the hash-set kernel at 93 registers (row offsets in scalar registers, walks as first + length) with static LDS and with dynamic LDS / five waves asked for; counts (tests)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s54
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in base dyn5; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  echo "== $v"
  for g in 23 orkut 21 24; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "GDN_TC_CORE_WGS=2" "GDN_TC_CORE_WGS=3" "GDN_TC_CORE_WGS=4" > $O/${g}_$v.txt 2>&1; tail -4 $O/${g}_$v.txt | head -3; done
done
