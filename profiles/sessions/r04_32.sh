# Round-4 session 32: whole GPU suite, then the profile session of the final code (tools/profile_r04.sh: bench line, rocprofv3
# kernel statistics of the same command, FETCH_SIZE / WRITE_SIZE passes, counter traffic of every bench block)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s32
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -30 $O/profile.log
