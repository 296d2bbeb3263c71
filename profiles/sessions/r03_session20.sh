# Round-3 session 20: compile-time variants of the bottom-up BFS step, one box (tools/build_variant.sh with VARIANT_FILE=gdn_bfs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s20
mkdir -p $O; rm -f $O/*.txt
for v in ${VARIANTS:-base}; do
  lib=gardenia_amd/lib/var_$v/libgardenia_hip.so
  [ $v = base ] && lib=gardenia_amd/lib/libgardenia_hip.so
  for sc in 24 26 27; do
  for cfg in "GDN_BFS_HUB_HEADS=0" "GDN_BFS_HUB_HEADS=1"; do
    echo "=== $v RMAT-$sc $cfg" >> $O/bfs.txt
    env GARDENIA_HIP_LIB=$PWD/$lib $cfg timeout 600 python3 tools/bfs_notorch.py $sc 2>&1 | grep "BFS RMAT" >> $O/bfs.txt
  done
  done
done
python3 - <<'PY'
import re
cur=None; acc={}
for l in open("gpurun_out/r03s20/bfs.txt"):
    if l.startswith("==="): cur=l.strip()[4:]; continue
    m=re.search(r": ([0-9.]+) ms",l)
    if m: acc.setdefault(cur,[]).append(float(m.group(1)))
for k,v in acc.items(): print(k, " ".join("%.3f"%x for x in v))
PY
