# Round-5 session 38: the tests of this round's new device code under the allocation fence (GDN_ALLOC_FENCE=1: every device buffer ends at
# the end of its own block of whole pages, a kernel that leaves a buffer faults): BFS (outer hubs, compact records, deferred depths), the
# fuzz variants, the range generator and the sharded bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GDN_ALLOC_FENCE=1
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs or cc_ or sssp_equal" 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "rmat_build" 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_gpu_bench_sharded.py -x -q -m gpu -k "two_ranks or eight" 2>&1 | tail -3
