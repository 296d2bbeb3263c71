# One session on ONE box, final code of round 6: the unprofiled default bench line, the rocprofv3 --kernel-trace --stats summary of the
# same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE: separate runs, --kernel-trace only, collection restricted to the two
# kernels of the iteration so that the graph build is not serialised) -- the same-session triple VERDICT r5 asked for (weak 10).
# tools/pmc_summary_r06.py turns gpurun_out/r06 into profiles/r06_pb_kernel_stats.{md,csv}, r06_pb_bench_same_session.json, pr_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06; rm -rf gpurun_out/r06/*
( date -u +"%Y-%m-%dT%H:%M:%SZ"; hostname; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > gpurun_out/r06/session.txt 2>&1
timeout 1500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-refsum > gpurun_out/r06/bench_under_rocprof.json 2> gpurun_out/r06/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "pb_(accumulate|expand)_kernel|mp_reduce_f64" --output-format csv -d gpurun_out/r06/$c -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs --no-extras --no-refsum > /dev/null 2> gpurun_out/r06/$c.log
done
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs --no-refsum > gpurun_out/r06/bench_after.json 2>> gpurun_out/r06/bench.log
cat gpurun_out/r06/session.txt; find gpurun_out/r06 -name "*.csv" | head -12
