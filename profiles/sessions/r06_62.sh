# Round-6 session 62: an upper bound of what overlapping consecutive PageRank iterations could buy (two independent chains on two streams)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06s62
timeout 900 python3 tools/pr_overlap_bound.py 27 20 > gpurun_out/r06s62/overlap.txt 2>&1; tail -5 gpurun_out/r06s62/overlap.txt
