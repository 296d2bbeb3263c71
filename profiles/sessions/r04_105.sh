# Round-4 session 105 (last): smoke, whole GPU suite as the driver runs it, the profile session (tools/profile_r04.sh), a default-mode sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s105
mkdir -p $O; rm -rf $O/*
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -9 $O/profile.log
export OMP_NUM_THREADS=4
for i in 1 2 3 4; do ( FUZZ_PLANS=1 timeout 1200 python3 tests/aids/fuzz_parity.py 250 $((51000000 + i * 1000)) > $O/fuzz$i.txt 2>&1; echo "fuzz $i: $(tail -1 $O/fuzz$i.txt | cut -c1-110)" ) & done; wait
