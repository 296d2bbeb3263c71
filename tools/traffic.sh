# usage: bash tools/traffic.sh <tag> <workload> <scale>      (on the GPU box; writes gpurun_out/<tag>/<workload>/...)
# HBM bytes per solve of one bench workload: FETCH_SIZE and WRITE_SIZE in passes of their own (--kernel-trace only: gpurun
# refuses mixed trace domains and the two counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), each
# at two solve counts (2 and 6): the difference is the traffic of 4 solves, builds cancel.
tag=$1; w=$2; scale=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag/$w
mkdir -p $O; rm -rf $O/*
( date -u +"%Y-%m-%dT%H:%M:%SZ"; hostname; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
for reps in 2 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${c}_$reps -- python3 tools/traffic_run.py $w $scale $reps > $O/${c}_$reps.log 2>&1
  done
done
python3 tools/traffic_run.py $w $scale 6 > $O/plain.log 2>&1
tail -1 $O/plain.log
