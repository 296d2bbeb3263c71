# Round-5 session 56: the whole GPU suite as the driver runs it, on the final code; smoke
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s56
mkdir -p $O; rm -rf $O/*
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1700 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_all.txt 2>&1; grep -E "FAILED|passed|failed|Error" $O/pytest_all.txt | head; grep -E "s call" $O/pytest_all.txt | head -8
# ... and the default bench line + its rocprofv3 kernel statistics on the same box (tools/profile_r05.sh without PMC passes)
timeout 2400 bash tools/profile_r05.sh > gpurun_out/r05_profile.log 2>&1; tail -2 gpurun_out/r05_profile.log
