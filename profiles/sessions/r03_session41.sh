# Round-3 session 41: slice sizes of BC's blocked levels (RMAT-24: ~200 chunks of 2^15 live sources for 256 CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s41
mkdir -p $O; rm -rf $O/*
for cfg in "X=0" "GDN_BC_LOG_CHUNK=14" "GDN_BC_LOG_CHUNK=14 GDN_BC_LOG_BIN=13" "GDN_BC_LOG_CHUNK=13 GDN_BC_LOG_BIN=13" "GDN_BC_LOG_BIN=13" "X=0"; do
  for sc in 24 26; do
    echo "$cfg | RMAT-$sc: $(env $cfg timeout 300 python3 tools/bc_notorch.py $sc plan 2>&1 | grep 'BC plan' | awk '{print $6}' | tr '\n' ' ')" >> $O/t.txt
  done
done
cat $O/t.txt
