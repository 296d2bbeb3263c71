# usage: bash tools/pr_channels.sh <outdir> [scale] [configs]   (on the GPU box; one rocprofv3 pass per counter set, --kernel-trace only)
out=$1; scale=${2:-27}; configs=${3:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
i=0
for set in "TCC_EA0_WRREQ_sum CH_WRREQ_max CH_WRREQ_min TCC_EA0_RDREQ_sum CH_RDREQ_max CH_RDREQ_min TCC_REQ_sum CH_REQ_max CH_REQ_min" \
           "TCC_EA0_WRREQ_STALL_sum CH_WRSTALL_max CH_WRSTALL_min TCC_TAG_STALL_sum CH_TAGSTALL_max TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum CH_WRDRAMCREDIT_max CH_WRDRAMCREDIT_min TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum CH_RDDRAMCREDIT_max CH_RDDRAMCREDIT_min" \
           "TCC_BUSY_sum CH_BUSY_max CH_BUSY_min TCC_EA0_WRREQ_LEVEL_sum CH_WRLEVEL_max CH_WRLEVEL_min TCC_EA0_RDREQ_LEVEL_sum CH_RDLEVEL_max CH_RDLEVEL_min" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_EA_BUSY GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"; do
  i=$((i+1))
  rocprofv3 -E tools/pmc_channels.yaml --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/pr_channels.py $scale $configs > $out/p$i.log 2>&1
  grep -E "^config|rror" $out/p$i.log | tr '\n' ';'; echo
done
# one pass with the raw (per-instance) counters in JSON: are the instances reported one by one?
rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_RDREQ --kernel-trace --output-format json -d $out/pj -- python3 tools/pr_channels.py $scale 2 > $out/pj.log 2>&1
python3 tools/pr_channels_summary.py $out $configs > $out/summary.txt 2>&1; cat $out/summary.txt
