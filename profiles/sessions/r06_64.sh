# Round-6 session 64: the next 64 neighbours' walk bounds requested before the current group is walked (variant nb): counts, same-box A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s64
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_nb/libgardenia_hip.so timeout 1500 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest_nb.txt 2>&1; tail -2 $O/pytest_nb.txt
for rep in 1 2; do for v in base nb; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for g in 23 orkut 21 24; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/${g}_${v}_$rep.txt 2>&1; echo "$v $rep: $(tail -2 $O/${g}_${v}_$rep.txt | head -1 | cut -c1-125)"; done
done; done
