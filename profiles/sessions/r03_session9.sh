# Round-3 session 9: the GPU suite on the final code, then ONE session of bench + rocprofv3 stats + FETCH/WRITE counters (tools/profile_r03.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03s9
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r03s9/pytest.txt 2>&1
tail -5 gpurun_out/r03s9/pytest.txt
bash tools/profile_r03.sh > gpurun_out/r03s9/profile.log 2>&1
tail -5 gpurun_out/r03s9/profile.log
tail -c 1500 gpurun_out/r03/bench.json
