# Round-3 session 43: is the placement spread a TLB matter?  address-translation counters of phase A / B for four plans of one process
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s43
mkdir -p $O; rm -rf $O/*
export GDN_PR_PLACE=0
python3 tools/pr_replan.py 27 4 > $O/plain.txt 2>&1; grep round $O/plain.txt
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $O/pmc1 -o t -- python3 tools/pr_replan.py 27 4 > $O/pmc1.txt 2>&1; grep round $O/pmc1.txt
rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o t -- python3 tools/pr_replan.py 27 4 > $O/pmc2.txt 2>&1; grep round $O/pmc2.txt
ls $O/pmc1 $O/pmc2
