"""bench.py's N > 1 code path on ONE device (--share-device: both ranks use cuda:0, collectives through gloo): the dense
and the compact contrib exchange must give the single-GPU run's L1 change to the last bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(extra, launcher=None):
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py"), "--scale", "20", "--steps", "4", "--warmup",
                                                 "1", "--no-bfs", "--no-cpu"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_on_one_device_match_single():
    single = _bench([])
    res = {}
    for ex, extra in (("dense", []), ("compact", []), ("compact", ["--no-squish"]), ("dense", ["--no-squish"])):
        launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port())]
        r = res[ex + " ".join(extra)] = _bench(["--gpus", "2", "--share-device", "--exchange", ex] + extra, launcher)
        assert r["n_gpus"] == 2 and ex + " exchange" in r["config"]["partition"]
        assert ("relabelled before the vertex-range cut" in r["config"]["layout"]) == (not extra)
        assert abs(r["pr_last_l1_change"] - single["pr_last_l1_change"]) <= 1e-12 * single["pr_last_l1_change"]
    assert "roofline" in single and single["roofline"]["frac"] > 0
