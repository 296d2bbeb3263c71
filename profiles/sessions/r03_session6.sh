# Round-3 session 6: read-once streams of the PageRank plan in UNCACHED device memory (GDN_EXPERIMENTS build, A/B)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s6
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
V=gardenia_amd/lib/var_exp/libgardenia_hip.so
for rep in 1 2 3; do
for u in 0 1 2 3 4 8 15; do
  echo "=== GDN_PB_UNCACHED=$u rep $rep" >> $O/uncached.txt
  GARDENIA_HIP_LIB=$V GDN_PB_UNCACHED=$u timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep "no-torch\|check:" >> $O/uncached.txt
done
done
grep "===\|no-torch" $O/uncached.txt | paste - - | sed 's/no-torch process: scale 27//; s/(best of 3 batches; first batch [0-9.]*)//' | sort
grep "check:" $O/uncached.txt | sort | uniq -c
