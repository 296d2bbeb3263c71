# Round-3 session 36: per-workgroup closing flushes / counter adds: tests, then BFS / SSSP / BC timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s36
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -m gpu -q -x -k "bfs or BFS or fuzz or sssp or SSSP or bc or BC or worklist or shapes or dropin" > $O/pytest.txt 2>&1; grep "passed\|failed" $O/pytest.txt | tail -2
for sc in 24 26 27; do timeout 300 python3 tools/bfs_notorch.py $sc 2>&1 | grep "BFS RMAT" | awk '{print $2, $5}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt; done
for k in "16 rand" "1 unit"; do REPS=6 timeout 300 python3 tools/sssp_trace.py 24 $k plan 2>&1 | grep "RMAT" | awk '{print $3, $4, $6}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt; done
REPS=4 timeout 300 python3 tools/sssp_trace.py 26 16 rand plan 2>&1 | grep "RMAT" | awk '{print $1, $6}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt
timeout 300 python3 tools/bc_notorch.py 24 plan 2>&1 | grep "BC plan" | awk '{print $3, $6}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt
timeout 300 python3 tools/bc_notorch.py 24 2>&1 | grep "BC" | awk '{print $2, $3, $5, $6}' | tr '\n' ' ' >> $O/t.txt; echo >> $O/t.txt
cat $O/t.txt
