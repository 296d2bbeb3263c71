# usage: bash tools/pr_place_pmc.sh <outdir>   (on the GPU box)
# Counters of phase A on the CANDIDATES of the plan's own placement search: twelve fresh allocations of vals in one process, each
# run 3 times as pb_expand_kernel<1> (tagged launches of the search; the trace line of a candidate gives its time), one rocprofv3
# pass per counter set.  Completes profiles/r05_pb_channels.md (whose passes 6-8 did not fit their session).
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
export GDN_PR_PLACE_TRACE=1 GDN_PR_PLACE_STOP=0 GDN_PR_PLACE_VALS=12
i=0
for set in "GRBM_GUI_ACTIVE GRBM_EA_BUSY GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-bfs --no-extras > $out/p$i.json 2> $out/p$i.log
done
python3 - "$out" <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p[0-9]*/")):
    cc = glob.glob(d + "**/*_counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*_kernel_trace.csv", recursive=True)
    log = open(d.rstrip("/") + ".log").read()
    ms = [float(x) for x in re.findall(r"vals +fresh \d+: ([0-9.]+) ms", log)]
    if not cc or not kt:
        print(d, "no output"); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt[0]))}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc[0])):
        if r["Kernel_Name"].startswith("void pb_expand_kernel<1>"):
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = list(per)
    names = sorted(per[ids[0]]) if ids else []
    print("==", d, "%d tagged expand dispatches, %d candidates in the trace" % (len(ids), len(ms)))
    print("  %-22s %8s %8s  %s" % ("", "trace ms", "kernel ms", "  ".join("%14s" % n[-14:] for n in names)))
    # 9 baseline dispatches (begin: 2 x 3, rebase: 3), then 3 per candidate: the first of each triple is untimed
    def row(label, grp, t):
        k = sum(dur[i] for i in grp) / len(grp)
        print("  %-22s %8s %8.3f  %s" % (label, t, k, "  ".join("%14.5g" % (sum(per[i][n] for i in grp) / len(grp)) for n in names)))
    if len(ids) >= 9:
        row("where the builder put it", ids[7:9], "")
    for c in range(len(ms)):
        grp = ids[9 + 3 * c + 1: 9 + 3 * c + 3]
        if len(grp) == 2:
            row("candidate %d" % c, grp, "%.3f" % ms[c])
PY
