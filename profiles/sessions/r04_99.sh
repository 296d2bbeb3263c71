# Round-4 session 99: the hash-set kernel beside the core with 4 KB sets (TC_CAP 512) and 4 / 5 / 6 waves per SIMD (variant builds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s99
mkdir -p $O; rm -rf $O/*
export TC_AB_CORES=0,12288,16384
for v in default cap512eu4 cap512eu5 cap512eu6; do
if [ $v = default ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
timeout 900 python3 tools/tc_core_ab.py 23 5 > $O/run23_$v.txt 2>&1
echo "variant $v"; grep RMAT $O/run23_$v.txt | tail -3; grep "same total" $O/run23_$v.txt
done
unset GARDENIA_HIP_LIB
timeout 1200 python3 -m pytest tests/test_gpu_configs.py -q -x -p no:cacheprovider -k "tc or triangle" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
