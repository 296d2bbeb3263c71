#!/bin/bash
# experiment driver: one no-torch PageRank timing run per environment setting (tools/pr_notorch.py)
SCALE=${SCALE:-27}
run() { echo "== $*"; env "$@" timeout 300 python tools/pr_notorch.py $SCALE 2>&1 | grep -v "^hub tier" | tail -6; }
run GDN_PB_HUB_ROWS=0
run GDN_PB_HUB_ROWS=1 GDN_PB_TRACE=1
run GDN_PB_HUB_ROWS=0
run GDN_PB_HUB_ROWS=1
