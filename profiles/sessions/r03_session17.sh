# Round-3 session 17: final same-session profile (bench + rocprofv3 stats + FETCH/WRITE) on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_r03.sh > gpurun_out/r03_profile.log 2>&1
tail -c 400 gpurun_out/r03/bench.json
