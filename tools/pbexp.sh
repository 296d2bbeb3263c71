#!/bin/bash
# experiment driver: one no-torch PageRank timing run per environment setting (tools/pr_notorch.py)
SCALE=${SCALE:-27}
run() { echo "== $*"; env "$@" timeout 300 python tools/pr_notorch.py $SCALE 2>&1 | tail -3; }
run GDN_PB_HUBS=0
run GDN_PB_HUB_MIN=1
run GDN_PB_HUB_MIN=2
run GDN_PB_HUB_MIN=3
run GDN_PB_HUB_MIN=4
run GDN_PB_HUB_MIN=6
