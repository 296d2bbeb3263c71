# Round-6 session 70: rows per group of the core kernel under a 64-register bound (K = 4096 / 8192 / 12288 / 16384: 6 / 3 / 2 / 2 against 4 / 2 / 1 / 1), one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s70
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base gall g123 base; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for g in 23 orkut 21 22 24 20; do timeout 600 python3 tools/tc_knob_ab.py $g 6 "" > $O/${g}_$v.txt 2>&1; echo "$v: $(tail -2 $O/${g}_$v.txt | head -1 | cut -c1-125)"; done
done
