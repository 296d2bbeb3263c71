# Round-6 session 13: the tests added since the last whole-suite run, the per-shard compute table on the final rule, then the
# same-session triple (bench line + kernel statistics + counter traffic): tools/profile_r06.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s13
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_bench_sharded.py -x -q > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reserved or reference or ticketed or sharded" > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
timeout 1500 python3 tools/shard_compute.py --n 1,2,4,8 --out $O/shard_compute.json > $O/shard.out 2> $O/shard.log; tail -1 $O/shard.out
timeout 3000 bash tools/profile_r06.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
