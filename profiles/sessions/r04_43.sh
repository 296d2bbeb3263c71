# Round-4 session 43: PageRank phase A with an octet per lane (one U and one G load per 8 edges), same plan, runtime knob
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s43
mkdir -p $O; rm -rf $O/*
GARDENIA_HIP_LIB=gardenia_amd/lib/var_exp/libgardenia_hip.so timeout 600 python3 tools/pr_ab.py GDN_PB_AVAR 0 16 27 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
