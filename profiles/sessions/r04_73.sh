# Round-4 session 73: the randomised sweep under knobs that force this round's other paths (600 fresh graphs each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s73
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
H="GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2"
run() { name=$1; first=$2; shift 2; ( env FUZZ_PLANS=1 "$@" timeout 2400 python3 tests/aids/fuzz_parity.py 600 $first > $O/$name.txt 2>&1; echo "$name: $(grep -E 'MISMATCH|fuzz parity' $O/$name.txt | tail -1)" ) & }
run hostloop 21000001 GDN_SSSP_SMALL=0 GDN_SSSP_COOP=0 $H
run adapt_all 22000001 GDN_SSSP_ADAPT_AFTER=0 GDN_SSSP_LIGHT_SMALL=1000000000 GDN_SSSP_LIGHT_COOP=1000000000 GDN_SSSP_LIGHT_HOST=1000000000 $H
run coop 23000001 GDN_SSSP_COOP=1 GDN_SSSP_SMALL=0 GDN_BFS_COOP=1 $H
run window 24000001 GDN_BFS_BU_FORM=window $H
wait
run plain_streams 25000001 GDN_PB_REC_IL=0 GDN_PB_V_IL=0 GDN_SSSP_REC_IL=0 GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PB_HUB_MIN=8 GDN_PB_MID_CAP=300 $H
run old_builder 26000001 GDN_PB_BUILDER=old GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PB_HUB_MIN=8 GDN_PB_MID_CAP=300 $H
run cand 27000001 GDN_SSSP_DENSE_IN=100000 GDN_SSSP_DENSE_OUT=1000000 $H
run small_far 28000001 GDN_SSSP_SMALL_FAR=1 GDN_SSSP_SMALL=1 $H
wait
