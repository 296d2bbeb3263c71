# Round-4 session 14: why the SSSP solve is slower on the new layout: tiers off / on, old / new builder
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s14
mkdir -p $O; rm -rf $O/*
for b in old new; do for t in 0 4; do
  echo "== builder $b tiers $t"
  GDN_SSSP_TRACE=1 GDN_PB_TRACE=1 GDN_PB_BUILDER=$b GDN_SSSP_TIERS=$t python3 tools/sssp_prep.py 24 2>&1 | grep 'scale\|plan:\|pb_build\] edges\|dense sweep' | sed -n '1,3p;/scale/p' | cut -c1-220 | head -12
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
