# Round-3 session 16: TC -- share of the packed (short-list) path, TC_LONG variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s16
mkdir -p $O; rm -f $O/tc.txt
for lib in lib/libgardenia_hip.so lib/var_abl6/libgardenia_hip.so lib/var_long16/libgardenia_hip.so lib/var_long24/libgardenia_hip.so lib/var_long32/libgardenia_hip.so; do
for sc in 19 21 23; do
  for f in f a; do
    echo "=== TC RMAT-$sc form $f $lib" >> $O/tc.txt
    GARDENIA_HIP_LIB=gardenia_amd/$lib GDN_TC_FORM=$f timeout 600 python3 tools/tc_notorch.py $sc 2 2>&1 | grep RMAT | tail -1 >> $O/tc.txt
  done
done
done
paste - - < $O/tc.txt | awk '{print $3,$5,$6, $(NF-10), $(NF-9)}'
