# Round-4 session 101: whole GPU suite + the profile session (tools/profile_r04.sh) after the TC core
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s101
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -12 $O/profile.log
