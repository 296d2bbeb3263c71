# usage: bash tools/pmc_generic.sh <name> <kernel-name-substring> <python script + args...>
# PMC passes (each counter set in its own run, --kernel-trace only) over one script; prints per-kernel averages.
name=$1; kern=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$name
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$name/p$i -- python3 "$@" > gpurun_out/pmc_$name/p$i.log 2>&1
done
python3 - "$name" "$kern" <<'PY'
import csv, glob, collections, sys
name, kern = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob("gpurun_out/pmc_%s/p*/" % name)):
    for f in glob.glob(d + "**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if kern in k[0]:
                print("%-42s %-32s n=%3d avg=%.4g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
