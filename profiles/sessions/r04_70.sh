# Round-4 session 70: the two aborted modes of session 69 from where they stopped (borderline PageRank stop / cancelling SpMV row now arbitrated)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s70
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( timeout 3000 python3 tests/aids/fuzz_parity.py 1968 11000533 > $O/plain.txt 2>&1; grep -E "^\(|MISMATCH|fuzz parity" $O/plain.txt ) &
( GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 timeout 3000 python3 tests/aids/fuzz_parity.py 1635 12000866 > $O/blocked.txt 2>&1; grep -E "^\(|MISMATCH|fuzz parity" $O/blocked.txt ) &
wait
