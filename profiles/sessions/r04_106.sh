# Round-4 session 106: the blocked PageRank layout on shapes without hubs under slice-geometry knobs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s106
mkdir -p $O; rm -rf $O/*
for s in uniform small_world; do
timeout 900 python3 tools/pr_shape_knobs.py $s "" "GDN_PB_SLICES_LOG=8" "GDN_PB_SLICES_LOG=10" "GDN_PB_SLICES_LOG=11" "GDN_PB_LOG_CHUNK=15 GDN_PB_LOG_BIN=13" "GDN_PB_LOG_CHUNK=13 GDN_PB_LOG_BIN=15" "GDN_PB_LOG_CHUNK=15 GDN_PB_LOG_BIN=15" "GDN_PB_LOG_CHUNK=13 GDN_PB_LOG_BIN=13" "GDN_PB_PAD=32" "GDN_PB_PAD=64" > $O/$s.txt 2>&1; cat $O/$s.txt | tail -22
done
