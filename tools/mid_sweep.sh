# phase-B ablation with the record tiers (timing only; GDN_PB_DBG / GDN_PB_MIDVAR need the EXPERIMENTS build)
export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_exp/libgardenia_hip.so
for cfg in "0 1" "0 2" "0 3" "0 4"; do
  set -- $cfg
  echo "=== GDN_PB_DBG=$1 GDN_PB_MIDVAR=$2"
  GDN_PB_DBG=$1 GDN_PB_MIDVAR=$2 timeout 600 python tools/pr_notorch.py 27 2>&1 | grep "no-torch\|check"
done
