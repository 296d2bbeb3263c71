# Round-6 session 41: registers of the hash-set kernel (101 at four chunks per step, 93 at three, 87 at two) against the core kernel's room beside it (workgroups per CU 1 / 2 / 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s41
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base unr3 unr2; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  echo "== $v"
  timeout 600 python3 tools/tc_knob_ab.py 23 6 "" "GDN_TC_CORE_WGS=1" "GDN_TC_CORE_WGS=3" "GDN_TC_CORE_ASYNC=0" > $O/r23_$v.txt 2>&1; tail -5 $O/r23_$v.txt | head -4
  timeout 600 python3 tools/tc_knob_ab.py orkut 6 "" "GDN_TC_CORE_WGS=3" > $O/orkut_$v.txt 2>&1; tail -3 $O/orkut_$v.txt | head -2
done
