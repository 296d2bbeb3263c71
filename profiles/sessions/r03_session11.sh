# Round-3 session 11: randomised parity sweep (tests/aids/fuzz_parity.py) on the final code: default options with the plans,
# then with the opt-in forms (binned SSSP passes, binary-search TC)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s11
mkdir -p $O
OMP_NUM_THREADS=8 FUZZ_PLANS=1 timeout 1500 python3 tests/aids/fuzz_parity.py 500 9100 > $O/fuzz_default.txt 2>&1; echo "rc $?" >> $O/fuzz_default.txt
tail -3 $O/fuzz_default.txt
OMP_NUM_THREADS=8 FUZZ_PLANS=1 GDN_SSSP_BINS=1 GDN_SSSP_DENSE_IN=1000 GDN_TC_FORM=bs timeout 900 python3 tests/aids/fuzz_parity.py 200 9700 > $O/fuzz_optin.txt 2>&1; echo "rc $?" >> $O/fuzz_optin.txt
tail -3 $O/fuzz_optin.txt
