# Round-3 session 8: plan arrays re-homed into virtual ranges of SHUFFLED physical chunks (GDN_EXPERIMENTS build, A/B of the placement spread)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s8
mkdir -p $O; rm -f $O/vmm.txt
V=gardenia_amd/lib/var_exp/libgardenia_hip.so
for rep in 1 2 3 4; do
for cfg in "GDN_PB_VMM=0" "GDN_PB_VMM=64" "GDN_PB_VMM=2" "GDN_PB_VMM=512" "GDN_PB_VMM=64 GDN_PB_VMM_WHAT=3" "GDN_PB_VMM=64 GDN_PB_VMM_WHAT=15"; do
  echo "=== $cfg rep $rep" >> $O/vmm.txt
  ( S=$SECONDS; env GARDENIA_HIP_LIB=$V $cfg timeout 600 python3 tools/pr_notorch.py 27 2 2>&1 | grep -i "no-torch\|error\|fail"; echo "wall $((SECONDS - S)) s" ) >> $O/vmm.txt
done
done
grep "===\|no-torch" $O/vmm.txt | paste - - | sed 's/no-torch process: scale 27//; s/(best of 3 batches; first batch [0-9.]*)//' | sort
grep "wall" $O/vmm.txt | sort | uniq -c | head -30
