# Round-3 session 34: CC sampling round 1 in two launches: size of the head
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s34
mkdir -p $O; rm -rf $O/*
for h in 0 1024 16384 262144 1048576; do
  for sc in 24 26; do
    echo "head $h: $(env GDN_CC_HEAD=$h timeout 300 python3 tools/cc_notorch.py $sc 2>&1 | grep 'RMAT' | head -2 | awk '{print $1, $4 $5 $6, $7, $8}' | tr '\n' ' ')" >> $O/cc.txt
  done
done
cat $O/cc.txt
