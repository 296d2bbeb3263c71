# Round-5 session 7: Shiloach-Vishkin variants (rounds as launches, fused kernel), placement test, bench with the early-stopping search
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s07
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shiloach or placement or test_cc" > $O/t_parity.txt 2>&1; tail -3 $O/t_parity.txt
GDN_PR_PLACE_TRACE=1 timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; grep "place\]" $O/bench.err | tail -8; python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05s07/bench.json") if l.startswith("{")][-1])
print("PR ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "parts", d["roofline"]["kernel_ms_parts"], "plan_build_s", d["config"]["plan_build_s"])
t = d["traversal"]
print("cc", t["cc_with_reverse_graph"]["ms"], t["cc_out_edges_only"]["ms"], t.get("cc_sv_fused_kernel"))
PY
