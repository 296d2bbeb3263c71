# Round-3 session 12: the forward triangle count -- tests, timing against round 2's form, counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s12
mkdir -p $O; rm -f $O/tc.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py tests/test_reference_dropin.py tests/test_host_mains.py -m gpu -q -k "tc or TC or triangle or shapes or fuzz or dropin or mains" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for sc in 19 21 23; do
  for f in f a; do
    echo "=== TC RMAT-$sc form $f" >> $O/tc.txt
    GDN_TC_FORM=$f timeout 600 python3 tools/tc_notorch.py $sc 4 >> $O/tc.txt 2>&1
  done
done
cat $O/tc.txt
