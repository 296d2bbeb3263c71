# Round-3 session 13: TC plan API + forward default: tests, timings, counters of the forward count, full bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s13
mkdir -p $O; rm -f $O/tc.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bench_sharded.py tests/test_reference_dropin.py tests/test_host_mains.py tests/test_gpu_fuzz.py -m gpu -q -k "tc or triangle or bench_line or dropin or mains or fuzz" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for sc in 21 22 23 24; do
  echo "=== TC RMAT-$sc default" >> $O/tc.txt
  timeout 600 python3 tools/tc_notorch.py $sc 3 >> $O/tc.txt 2>&1
done
cat $O/tc.txt
bash tools/pmc_generic.sh tc4 tc_count tools/tc_notorch.py 23 1 > $O/tc_pmc_forward.txt 2>&1
grep -v "^$" $O/tc_pmc_forward.txt | tail -32
rm -rf gpurun_out/pmc_tc4
timeout 1500 python3 bench.py > $O/bench.json 2> $O/bench.log
python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r03s13/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["roofline"]["frac"])
print(json.dumps(j["tc"])[:3000])
PY
