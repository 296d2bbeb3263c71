# usage: bash tools/pmc_pb.sh <tag> [scale]    (on the GPU box; gpurun_out/pmc_<tag>/)
# Hardware counters of the two kernels of the PageRank iteration on the FINAL code: every counter set in a pass of its own
# (--pmc with --kernel-trace only), collection restricted to the two kernels (--kernel-include-regex: the graph build is not
# serialised under the counters), over the torch-free driver tools/attic/pr_notorch.py <scale> 2 (squished PB plan).
# tools/pmc_pb_summary.py <tag> turns the passes into profiles/<tag>_phaseB_counters.md.
tag=$1; scale=${2:-27}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_$tag
mkdir -p $O; rm -rf $O/*
( date -u +"%Y-%m-%dT%H:%M:%SZ"; hostname ) > $O/session.txt 2>&1
timeout 600 python3 tools/attic/pr_notorch.py $scale 2 > $O/unprofiled.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/attic/pr_notorch.py $scale 2 > $O/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "pb_(accumulate|expand)_kernel" --output-format csv -d $O/p$i -- python3 tools/attic/pr_notorch.py $scale 2 > $O/p$i.log 2>&1
  echo "pass $i ($set): rc $?" >> $O/passes.txt
done
cat $O/passes.txt; tail -4 $O/unprofiled.txt
