# Round-6 session 17: shards with FULL-size bins spread over whole rounds (GDN_PB_SLICES_LOG=9 on the experiments build: a shard's
# bins are 2^14 rows like the whole graph's, N = 8: 512 bins = two full rounds instead of 793 = 3.1) against the shipped rule
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s17
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_exp/libgardenia_hip.so GDN_TEST_HOOKS=1
timeout 1200 python3 tools/shard_compute.py --n 4,8 --out $O/shard_default.json > $O/a.out 2> $O/a.log
GDN_PB_SLICES_LOG=9 timeout 1200 python3 tools/shard_compute.py --n 4,8 --out $O/shard_s9.json > $O/b.out 2> $O/b.log
GDN_PB_SLICES_LOG=9 GDN_PB_BALANCE_SHARDS=1 timeout 1200 python3 tools/shard_compute.py --n 4,8 --out $O/shard_s9b.json > $O/c.out 2> $O/c.log
GDN_PB_SLICES_LOG=11 timeout 1200 python3 tools/shard_compute.py --n 8 --out $O/shard_s11.json > $O/d.out 2> $O/d.log
python3 - <<'PY'
import json
O = "gpurun_out/r06s17"
for n in ("shard_default", "shard_s9", "shard_s9b", "shard_s11"):
    try:
        for s in json.load(open("%s/%s.json" % (O, n)))["shards"]:
            print("%-14s N %d rank %d bins %4d log_blk %d whole A %.3f B %.3f | ticketed A %.3f B %.3f | frac %.3f plan %.2f" % (n, s["n"], s["rank"], s["bins"], s["log_blk"], s["whole_launch"]["phase_a_ms"], s["whole_launch"]["phase_b_ms"], s["ticketed"]["phase_a_ms"], s["ticketed"]["phase_b_ms"], s["frac_of_peak"], s["plan_build_s"]))
    except Exception as e:
        print(n, "failed", e)
PY
