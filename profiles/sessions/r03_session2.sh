# Round-3 session 2: the GPU suite on the new tier picker / NaN-safe SpMV / drop-in binaries, then the tier sweep with
# linear thresholds (variant build: up to 8 mid tiers).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s2
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
V=gardenia_amd/lib/var_mid8/libgardenia_hip.so
for rep in 1 2; do
for cfg in "2 4" "2 1" "3 1" "4 1" "5 1" "6 1" "7 1" "8 1"; do
  set -- $cfg
  echo "=== GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2 rep $rep" >> $O/tier_sweep.txt
  GARDENIA_HIP_LIB=$V GDN_PB_TRACE=1 GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2 timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep -v "^\[pb_build\] .*keys" >> $O/tier_sweep.txt
done
done
grep "===\|no-torch\|mid tiers\|pick_tiers" $O/tier_sweep.txt
