# Round-6 session 29: TC defaults moved (forward count from 2^22 DAG edges, core of 8192 ranks from 2^19 vertices): tests, small scales; what the walks cost when L2 serves the lists (ablation 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s29
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for s in 17 18 19 20 21; do timeout 600 python3 tools/tc_knob_ab.py $s 8 "" "GDN_TC_FORM=a" > $O/tc_small_$s.txt 2>&1; tail -3 $O/tc_small_$s.txt | head -2; done
export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_abl8/libgardenia_hip.so
for mode in beside alone; do
  spec=""; [ $mode = alone ] && spec="GDN_TC_CORE_ASYNC=0"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl8.$mode -- python3 tools/tc_knob_ab.py 23 5 "$spec" > $O/abl8.$mode.txt 2>&1
  grep "count median" $O/abl8.$mode.txt | tail -1; grep -h "tc_count_kernel\|tc_core_count" $O/abl8.$mode/*/*_kernel_stats.csv | cut -d, -f1-4
done
