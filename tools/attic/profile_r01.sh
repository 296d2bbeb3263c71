# rocprofv3 passes of the default bench (RMAT-27, PB layout).  Kernel trace and PMC counters are
# collected in SEPARATE runs (gpurun refuses mixed trace domains; FETCH_SIZE and WRITE_SIZE do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pb_trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/prof_pb_bench.json 2> gpurun_out/prof_pb_bench.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pb_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs > /dev/null 2> gpurun_out/prof_pb_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pb_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-bfs > /dev/null 2> gpurun_out/prof_pb_write.log
find gpurun_out/prof_pb_* -name "*.csv" | head -20
