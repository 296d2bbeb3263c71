# First thing to run on a lease with >= 2 GPUs (none has been available in rounds 1-5):
#   bash tools/first_multi_gpu_lease.sh [outdir]
# Order: cheapest and most diagnostic first, every step under its own timeout, nothing re-execs after touching a GPU.
out=${1:-gpurun_out/multi}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p $out
n=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices: $n" | tee $out/devices.txt
rocm-smi --showtopo > $out/topo.txt 2>&1
if [ "$n" -lt 2 ]; then echo "one device: nothing to do"; exit 0; fi
# 1. links: peer copies and ncclAllGather of 32 / 64 / 252 MB (the squished contrib vector of RMAT-27 is 252 MB)
timeout 300 tools/_bin/xgmi_probe > $out/xgmi_probe.txt 2>&1; tail -20 $out/xgmi_probe.txt
# 2. the seven two-device tests (gdn_pr_multi / gdn_spmv_multi with peer copies and RCCL, sharded TC): bit-equality with one device
timeout 900 python3 -m pytest tests/test_gpu_multi_devices.py -x -q -m gpu > $out/t_multi_devices.txt 2>&1; tail -3 $out/t_multi_devices.txt
# 3. bench.py, one rank per GPU over RCCL, small first (a wrong geometry shows at scale 22 in seconds), then the headline
for g in 2 4 8; do
  [ "$g" -le "$n" ] || continue
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29500 + g)) \
      bench.py --gpus $g --scale 22 --steps 5 --warmup 2 --no-bfs --no-cpu --no-extras > $out/bench_s22_n$g.json 2> $out/bench_s22_n$g.err
  tail -c 600 $out/bench_s22_n$g.json; echo
  # (the default generates every rank's destination range on its own; --gen whole = the shard cut out of the whole graph: same L1 change)
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29550 + g)) \
      bench.py --gpus $g --scale 22 --steps 5 --warmup 2 --gen whole --no-bfs --no-cpu --no-extras > $out/bench_s22_whole_n$g.json 2> $out/bench_s22_whole_n$g.err
  python3 -c "import json,sys; a,b=(json.loads([l for l in open(f) if l.startswith('{')][-1])['pr_last_l1_change'] for f in sys.argv[1:3]); print('range vs whole L1 change:', a, b, 'EQUAL' if abs(a-b) <= 1e-12*abs(b) else 'DIFFERENT')" $out/bench_s22_n$g.json $out/bench_s22_whole_n$g.json
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $out/bench_n1.json 2> $out/bench_n1.err
for g in 2 4 8; do
  [ "$g" -le "$n" ] || continue
  timeout 1200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29600 + g)) \
      bench.py --gpus $g --steps 20 --warmup 5 > $out/bench_n$g.json 2> $out/bench_n$g.err
  python3 - $out/bench_n1.json $out/bench_n$g.json <<'PY'
import json, sys
a, b = (json.loads([l for l in open(f) if l.startswith("{")][-1]) for f in sys.argv[1:3])
print("N=%d: %.3f ms/step, %.1f G edges/s, x%.2f of N=1 (%.0f%% efficiency); exchange: %s" % (
    b["n_gpus"], b["ms_per_step"], b["value"] / 1e9, b["value"] / a["value"], 100 * b["value"] / a["value"] / b["n_gpus"],
    b["config"]["partition"][:160]))
PY
done
