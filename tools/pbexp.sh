#!/bin/bash
# experiment driver: one no-torch PageRank timing run per environment setting (tools/pr_notorch.py)
SCALE=${SCALE:-27}
run() { echo "== $*"; env "$@" timeout 300 python tools/pr_notorch.py $SCALE 2>&1 | tail -2; }
run GDN_X=0
run GDN_PB_AVAR=1
run GDN_PB_SPLIT=2
run GDN_PB_LOG_BIN=13
