# BFS direction threshold sweep (GDN_BFS_ALPHA_DENSE) on RMAT-27
for a in 8 16 32 64 128; do echo "== alpha_dense $a"; GDN_BFS_ALPHA_DENSE=$a timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 2>&1 >/dev/null | grep "BFS from" | sed 's/.*ms.: //' | cut -c1-60; done
