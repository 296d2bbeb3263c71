# Round-3 session 15: TC with the scalar chunk iterator and the cheaper ballots; multiplicative vs xor bucket hash (variant build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s15
mkdir -p $O; rm -f $O/tc.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py tests/test_reference_dropin.py tests/test_host_mains.py -m gpu -q -k "tc or TC or triangle or shapes or fuzz or dropin or mains" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for lib in gardenia_amd/lib/libgardenia_hip.so gardenia_amd/lib/var_tcx/libgardenia_hip.so; do
for sc in 19 21 23; do
  for f in f a; do
    echo "=== TC RMAT-$sc form $f lib $lib" >> $O/tc.txt
    GARDENIA_HIP_LIB=$lib GDN_TC_FORM=$f timeout 600 python3 tools/tc_notorch.py $sc 3 2>&1 | grep RMAT >> $O/tc.txt
  done
done
done
cat $O/tc.txt | cut -c1-200
