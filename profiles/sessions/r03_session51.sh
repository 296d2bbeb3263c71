# Round-3 session 51: what bounds phase B of the SSSP sweeps (with record tiers): LDS / TA / wait counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=3
bash tools/pmc_generic.sh sssp_acc sssp_pb_accumulate tools/sssp_trace.py 24 16 rand plan > gpurun_out/pmc_sssp_acc.txt 2>&1
grep -v "^W2026\|^E2026" gpurun_out/pmc_sssp_acc.txt | tail -40
