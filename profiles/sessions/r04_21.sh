# Round-4 session 21: lane-interleaved mid-tier record streams (phase B form 2): parity, then A/B on RMAT-27
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s21
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "pr or pagerank or PageRank or PR" > $O/pytest_pr.txt 2>&1; tail -3 $O/pytest_pr.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "200001" > $O/pytest_fuzz.txt 2>&1; tail -3 $O/pytest_fuzz.txt
timeout 600 python3 tools/pr_ab_plan.py GDN_PB_REC_IL 0 1 27 4 > $O/ab_il2.txt 2>&1; tail -4 $O/ab_il2.txt
GARDENIA_HIP_LIB=gardenia_amd/lib/var_il4/libgardenia_hip.so timeout 600 python3 tools/pr_ab_plan.py GDN_PB_REC_IL 0 1 27 4 > $O/ab_il4.txt 2>&1; tail -4 $O/ab_il4.txt
timeout 300 python3 tools/pr_ab_plan.py GDN_PB_REC_IL 0 1 24 4 > $O/ab_il2_s24.txt 2>&1; tail -3 $O/ab_il2_s24.txt
