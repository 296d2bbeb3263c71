# Round-6 session 36: the flat scan with four steps of gathers in flight and the next rows offsets requested ahead
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s36
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs_flat.txt 2>&1; grep -E "^round|level [345] bottom|traced" $O/bfs_flat.txt
for s in 24 22 26; do timeout 600 python3 tools/traffic_run.py bfs $s 6 2>&1 | tail -1; done
