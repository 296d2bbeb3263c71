#!/bin/bash
# experiment driver: one no-torch PageRank timing run per environment setting (tools/pr_notorch.py)
SCALE=${SCALE:-27}
run() { echo "== $*"; env "$@" timeout 300 python tools/pr_notorch.py $SCALE 2>&1 | tail -2; }
run GDN_X=0
run GDN_PB_PAD=32 GDN_PB_LOG_GROUP=3
run GDN_PB_PAD=64 GDN_PB_LOG_GROUP=5
run GDN_PB_PAD=64 GDN_PB_LOG_GROUP=6
run GDN_PB_PAD=16 GDN_PB_LOG_GROUP=3
