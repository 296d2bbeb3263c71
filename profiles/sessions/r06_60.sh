# Round-6 session 60: same box A/B of the chunk clamp (minimum against compare + select) and of the 24-bit multiply in the bucket hash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s60
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for rep in 1 2; do for v in base oldclamp mul24; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for g in 23 orkut 21; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/${g}_${v}_$rep.txt 2>&1; echo "$v $rep: $(tail -2 $O/${g}_${v}_$rep.txt | head -1 | cut -c1-120)"; done
done; done
