# Round-3 session 44: PageRank on R-MAT scale 28 (268 M vertices, 4.3 G edges: what 288 GB of HBM is for)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s44
mkdir -p $O; rm -rf $O/*
env GDN_PR_PLACE_TRACE=1 timeout 900 python3 tools/pr_notorch.py 28 2 > $O/pr28.txt 2>&1; grep -v " try " $O/pr28.txt | tail -8
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
