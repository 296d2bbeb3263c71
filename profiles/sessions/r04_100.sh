# Round-4 session 100: the hash-set kernel's whole-row limit (GDN_TC_LIGHT) beside the core
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s100
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=12288,16384
for l in 512 128 256 1024 2048; do
export GDN_TC_LIGHT=$l
timeout 900 python3 tools/tc_core_ab.py 23 5 > $O/run23_$l.txt 2>&1
echo "light $l"; grep RMAT $O/run23_$l.txt | tail -2
done
