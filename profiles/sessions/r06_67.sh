# Round-6 session 67 (last): the same-session triple of the headline on the LAST library of the round (bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes): tools/profile_r06.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s67
mkdir -p $O; rm -rf $O/*
timeout 3000 bash tools/profile_r06.sh > $O/profile.log 2>&1; tail -3 $O/profile.log
