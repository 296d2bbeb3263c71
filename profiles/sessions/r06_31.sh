# Round-6 session 31: counters of tc_count_kernel with the masked look-ups (var_old) and the mask-free ones (the working tree): why fewer instructions are slower
export GDN_TEST_HOOKS=1
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_old/libgardenia_hip.so
bash tools/pmc_generic.sh tc_old tc_count_kernel tools/tc_knob_ab.py 23 2 "" > $GRAFT_REPO_ROOT/gpurun_out/pmc_tc_old.txt 2>&1
unset GARDENIA_HIP_LIB
bash tools/pmc_generic.sh tc_new tc_count_kernel tools/tc_knob_ab.py 23 2 "" > $GRAFT_REPO_ROOT/gpurun_out/pmc_tc_new.txt 2>&1
cd "$GRAFT_REPO_ROOT"; paste -d'|' <(cut -c44- gpurun_out/pmc_tc_old.txt) <(cut -c76- gpurun_out/pmc_tc_new.txt)
