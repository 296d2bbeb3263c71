# Round-5 session 2: RMAT-27 BFS x3 + PR converged vs the OpenMP oracle (test fixed); per-channel counter passes over slow / fast placements
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s02
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "oracle" -s --durations=5 > $O/t_fullsize.txt 2>&1; grep -E "BFS RMAT|PR RMAT|GDN_PR_SUM|passed|failed|Error|assert|s call" $O/t_fullsize.txt | head -30
bash tools/pr_channels.sh $O/ch 27 6 > $O/channels.log 2>&1; tail -120 $O/channels.log
ls -la $O/ch/pj | head; find $O/ch/pj -name "*.json" -size +1k | head -3
