# Round-3 session 31: GPU-busy time inside one BFS (kernel durations from rocprofv3 --kernel-trace) against its wall time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s31
mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --output-format csv -d $O/bfs -o bfs -- python3 tools/bfs_notorch.py 27 > $O/bfs.log 2>&1
grep "BFS RMAT" $O/bfs.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03s31/bfs/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the searches are the tail of the trace: split at bfs_init-like fills? print the last 120 dispatches compactly
names = [r["Kernel_Name"].split("(")[0][:34] for r in rows]
# find the last 4 occurrences of the first kernel of a search: gdn_fill (dist = INF)
idx = [i for i, n in enumerate(names) if "fill" in n.lower()]
starts = idx[-4:]
for si, s in enumerate(starts):
    e = starts[si + 1] if si + 1 < len(starts) else len(rows)
    seg = rows[s:e]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("search %d: %d dispatches, span %.3f ms, kernels busy %.3f ms" % (si, len(seg), (t1 - t0) / 1e6, busy / 1e6))
    if si == 0:
        prev = None
        for r in seg:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (st - prev) / 1e3 if prev else 0
            print("   %-36s %8.1f us  (gap before %6.1f us)" % (r["Kernel_Name"].split("(")[0][:36], (en - st) / 1e3, gap))
            prev = en
PY
