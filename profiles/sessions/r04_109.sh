# Round-4 session 109 (last): smoke, whole GPU suite as the driver runs it, the profile session (tools/profile_r04.sh), TC + ingest tests under the allocation fence
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s109
mkdir -p $O; rm -rf $O/*
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -8 $O/profile.log
GDN_ALLOC_FENCE=1 timeout 1200 python3 -m pytest tests/test_ingest.py tests/test_gpu_parity.py -q -x -m gpu -k "ingest or builder or symmetrize or tc" -p no:cacheprovider > $O/fence.txt 2>&1; echo "fence: $(tail -1 $O/fence.txt)"
