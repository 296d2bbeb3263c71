# Round-3 session 14 (final): the whole GPU suite, then the same-session profile (bench + rocprofv3 stats + FETCH/WRITE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03s14
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r03s14/pytest.txt 2>&1
tail -5 gpurun_out/r03s14/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03s14/smoke.txt 2>&1; tail -2 gpurun_out/r03s14/smoke.txt
bash tools/profile_r03.sh > gpurun_out/r03s14/profile.log 2>&1
tail -c 600 gpurun_out/r03/bench.json
