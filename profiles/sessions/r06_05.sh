# Round-6 session 5: the placement of `vals` (last paragraph): is memory that no build has touched uniformly fast?
# Five fresh processes each: (a) vals in a 4 GB block reserved as the FIRST device call of the process (bench.py --reserve-gb 4),
# no search; (b) vals where hipMalloc puts it behind the build, no search; (c) three processes with the search (the default).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s05
mkdir -p $O; rm -rf $O/*
Q="--no-extras --no-bfs --no-cpu --no-refsum --steps 20 --warmup 5"
for i in 1 2 3 4 5; do
  GDN_PR_PLACE=0 GDN_PR_PLACE_TRACE=1 timeout 600 python3 bench.py $Q --reserve-gb 4 > $O/reserved_$i.json 2> $O/reserved_$i.log
  GDN_PR_PLACE=0 timeout 600 python3 bench.py $Q > $O/raw_$i.json 2> $O/raw_$i.log
done
for i in 1 2 3; do
  timeout 600 python3 bench.py $Q > $O/search_$i.json 2> $O/search_$i.log
  timeout 600 python3 bench.py $Q --reserve-gb 4 > $O/reserved_search_$i.json 2> $O/reserved_search_$i.log
done
python3 - <<'PY'
import json, glob
O = "gpurun_out/r06s05"
for kind in ("reserved", "raw", "search", "reserved_search"):
    rows = []
    for f in sorted(glob.glob("%s/%s_[0-9].json" % (O, kind))):
        try:
            r = json.loads([l for l in open(f) if l.startswith("{")][-1])
            rows.append((r["roofline"]["kernel_ms_parts"][0], r["roofline"]["kernel_ms_parts"][1], r["ms_per_step"], r["config"]["plan_build_s"]))
        except Exception as e:
            rows.append(("failed", str(e)))
    print(kind, " | ".join("A %.3f B %.3f step %.3f plan %.2fs" % x if len(x) == 4 else str(x) for x in rows))
PY
grep -h "reserved at process start" $O/reserved_1.log | head -2
