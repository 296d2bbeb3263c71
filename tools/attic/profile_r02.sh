# One session on ONE box: the unprofiled default bench line, the rocprofv3 --kernel-trace --stats summary of the same
# command and the two PMC passes (FETCH_SIZE / WRITE_SIZE: separate runs, --kernel-trace only -- gpurun refuses mixed trace
# domains and the two counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  tools/pmc_summary.py r02
# turns gpurun_out/r02_* into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
( date -u +"%Y-%m-%dT%H:%M:%SZ"; hostname; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ; rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 ) > gpurun_out/r02/session.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras > gpurun_out/r02/bench_under_rocprof.json 2> gpurun_out/r02/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs --no-extras > /dev/null 2> gpurun_out/r02/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02/write -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs --no-extras > /dev/null 2> gpurun_out/r02/write.log
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > gpurun_out/r02/bench_after.json 2>> gpurun_out/r02/bench.log
cat gpurun_out/r02/session.txt; find gpurun_out/r02 -name "*.csv" | head -20
