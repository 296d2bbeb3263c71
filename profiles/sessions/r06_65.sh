# Round-6 session 65: the hash-set kernel's list-length threshold (chunk stream / packed path) and slice again on the last code, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s65
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base long32 long64 long96 slice1024 slice256 base; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for g in 23 orkut 21; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/${g}_$v.txt 2>&1; echo "$v: $(tail -2 $O/${g}_$v.txt | head -1 | cut -c1-125)"; done
done
