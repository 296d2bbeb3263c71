# Round-4 session 107: chunks per step of the hash-set kernel (TC_UNR 2 / 4 / 8, variant builds) beside the core
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s107
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,12288,16384
for v in default unr2 unr8; do
if [ $v = default ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
timeout 900 python3 tools/tc_core_ab.py 23 5 > $O/run23_$v.txt 2>&1
echo "variant $v"; grep RMAT $O/run23_$v.txt | tail -3; grep "same total" $O/run23_$v.txt
done
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_unr2/libgardenia_hip.so
for w in 3 4; do GDN_TC_CORE_WGS=$w timeout 900 python3 tools/tc_core_ab.py 23 5 > $O/run23_unr2_w$w.txt 2>&1; echo "unr2 wgs $w"; grep RMAT $O/run23_unr2_w$w.txt | tail -2; done
