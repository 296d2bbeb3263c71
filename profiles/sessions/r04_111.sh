# Round-4 session 111: counters of the two TC kernels (every set in its own pass; kernels run one after the other under --pmc)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tc5
bash tools/pmc_generic.sh tc5 tc_co tools/tc_notorch.py 23 1 > gpurun_out/pmc_tc5_summary.txt 2>&1
cat gpurun_out/pmc_tc5_summary.txt | tail -70
