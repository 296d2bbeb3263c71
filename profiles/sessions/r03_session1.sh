# Round-3 session 1 (one box): phase-B counters of the shipped kernel, the tier-boundary sweep, the phase-B
# decomposition by timing-only ablations, and the allocation-stagger A/B.  Outputs under gpurun_out/r03s1/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s1
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
V=gardenia_amd/lib/var_mid6/libgardenia_hip.so
# 1. baseline + stagger A/B (default build)
timeout 600 python3 tools/pr_stagger.py 27 2 0 256 4352 69888 2101504 > $O/stagger.txt 2>&1
tail -7 $O/stagger.txt
# 2. tier sweep (variant: up to 6 mid tiers)
for cfg in "2 4" "2 2" "3 2" "4 2" "4 1" "6 1" "6 0"; do
  set -- $cfg
  echo "=== GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2" >> $O/tier_sweep.txt
  GARDENIA_HIP_LIB=$V GDN_PB_TRACE=1 GDN_PB_MID=$1 GDN_PB_MID_MIN16=$2 timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep -v "^\[pb_build\] .*keys" >> $O/tier_sweep.txt
done
grep "===\|no-torch\|mid tiers" $O/tier_sweep.txt
# 3. decomposition of phase B on the default tiers (timing-only: wrong results)
for d in 0 2 8 10 32 34 40 42; do
  echo "=== GDN_PB_DBG=$d" >> $O/ablation.txt
  GARDENIA_HIP_LIB=$V GDN_PB_DBG=$d timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep "no-torch" >> $O/ablation.txt
done
cat $O/ablation.txt
# 4. counters of phase B, shipped build (separate passes, --kernel-trace only)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -- python3 tools/pr_notorch.py 27 2 > $O/pmc_p$i.log 2>&1
  tail -1 $O/pmc_p$i.log
done
python3 - > $O/pmc_summary.txt <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03s1/pmc/p*/")):
    for f in glob.glob(d + "**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "pb_" in k[0]:
                print("%-42s %-32s n=%3d avg=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
cat $O/pmc_summary.txt
rm -rf $O/pmc
