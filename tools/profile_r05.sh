# One session on ONE box: the unprofiled default bench line, the rocprofv3 --kernel-trace --stats summary of the same
# command and the two PMC passes (FETCH_SIZE / WRITE_SIZE: separate runs, --kernel-trace only -- gpurun refuses mixed trace
# domains and the two counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  The PMC passes only with
# PMC=1 (8-9 minutes each at RMAT-27: every dispatch of the graph build is serialised under counter collection; the iteration
# kernels did not change in round 5, profiles/pr_traffic.json of round 4's session stands).  tools/pmc_summary.py r05_pb turns
# gpurun_out/r05 into profiles/ (without the PMC passes: kernel statistics and the bench lines only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
( date -u +"%Y-%m-%dT%H:%M:%SZ"; hostname; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ; rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 ) > gpurun_out/r05/session.txt 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras > gpurun_out/r05/bench_under_rocprof.json 2> gpurun_out/r05/trace.log
[ -n "$PMC" ] && timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r05/fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs --no-extras > /dev/null 2> gpurun_out/r05/fetch.log
[ -n "$PMC" ] && timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r05/write -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs --no-extras > /dev/null 2> gpurun_out/r05/write.log
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > gpurun_out/r05/bench_after.json 2>> gpurun_out/r05/bench.log
cat gpurun_out/r05/session.txt; find gpurun_out/r05 -name "*.csv" | head -20
# (the counter traffic of the other blocks -- tools/traffic.sh -- was taken in round 4 on kernels this round did not change:
# profiles/{bfs,spmv,tc,sssp_u255,cc,cc_out}_traffic.json stand; sssp_unit now runs as a BFS and carries no counter figure)
