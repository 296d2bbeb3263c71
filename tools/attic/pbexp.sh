#!/bin/bash
# experiment driver: one no-torch PageRank timing run per environment setting (tools/pr_notorch.py)
SCALE=${SCALE:-27}
run() { echo "== $*"; env "$@" timeout 300 python tools/pr_notorch.py $SCALE 2>&1 | grep "no-torch"; }
for i in 1 2 3; do
run GDN_PB_HUB_ROWS=0
run GDN_PB_HUB_ROWS=1
done
