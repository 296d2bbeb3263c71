# Round-6 session 26: what one address serves in device-scope atomic adds (tools/atomic_probe.hip); counter traffic of the triangle count on the striped counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s26
mkdir -p $O; rm -rf $O/*
timeout 300 tools/_bin/atomic_probe > $O/atomic_probe.txt 2>&1; cat $O/atomic_probe.txt
timeout 2400 bash tools/traffic.sh r06t1 tc 23 2>&1 | tail -1
