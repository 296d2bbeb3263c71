run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu --no-bfs --layout pb --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_parts'])"; }
run GDN_X=0
