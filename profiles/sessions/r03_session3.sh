# Round-3 session 3: GPU suite again (barrier / multi / one-shot SSSP / drop-in changes), tier count replicated
# (2 / 3 / 4 mid tiers, 5 interleaved repetitions, fresh process each), stagger 0 vs 4352 with 4 plans each in one process.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s3
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
V=gardenia_amd/lib/var_mid8/libgardenia_hip.so
for rep in 1 2 3 4 5; do
for cfg in "2 4" "3 1" "4 1"; do
  set -- $cfg
  echo "=== GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2 rep $rep" >> $O/tier_reps.txt
  GARDENIA_HIP_LIB=$V GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2 timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep "no-torch" >> $O/tier_reps.txt
done
done
cat $O/tier_reps.txt
GARDENIA_HIP_LIB=$V GDN_PB_MID=4 GDN_PB_MID_MIN16=1 timeout 900 python3 tools/pr_stagger.py 27 4 0 4352 > $O/stagger_mid4.txt 2>&1
tail -4 $O/stagger_mid4.txt
