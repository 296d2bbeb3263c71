# records-only phase B (GDN_PB_DBG=34: no main stream, no epilogue; timing only) in the ablation builds; MIDVAR=0: all tiers dword form
export GDN_PB_MIDVAR=0
for v in exp abl1 abl2 abl3; do
  export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so
  for dbg in 34 42; do
  echo "=== variant $v dbg $dbg"
  GDN_PB_DBG=$dbg timeout 600 python tools/pr_notorch.py 27 2>&1 | grep "no-torch"
  done
done
