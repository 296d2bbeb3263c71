# Round-4 session 9: memset -> kernel ordering probe; the fuzz tests and the whole suite on the kernel-zeroed arenas
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s09
mkdir -p $O; rm -rf $O/*
tools/_bin/memset_probe > $O/memset_probe.txt 2>&1; grep 'kernel\|ok\|BROKEN' $O/memset_probe.txt | head -40
GDN_SCRATCH_POISON=1 timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/pytest_fuzz_poison.txt 2>&1; tail -3 $O/pytest_fuzz_poison.txt
timeout 1200 python3 -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
