# Round-6 session 32: mask-free look-ups, second form (scalar way out per chunk) against the masked ones (var_old)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s32
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in base old; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  timeout 600 python3 tools/tc_knob_ab.py 23 6 "" "GDN_TC_CORE_ASYNC=0" > $O/r23_$v.txt 2>&1; tail -5 $O/r23_$v.txt | head -4
  timeout 600 python3 tools/tc_knob_ab.py orkut 6 "" > $O/orkut_$v.txt 2>&1; tail -2 $O/orkut_$v.txt | head -1
done
