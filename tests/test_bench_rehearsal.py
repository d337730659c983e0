"""bench.py's two multi-GPU modes, rehearsed end to end on ONE GPU (the reference shares one Arc<State>
between all workers, src/main.rs:108-112):
  * --one-context --gpus 2: one process, one flgpu context over devices [0, 0] -- two full shard contexts on the one card, the
    batch cut by flgpu_plan_shards;
  * two ranks under torch.distributed.run with --backend gloo (both on GPU 0): the per-rank mode the driver launches with nccl,
    including the table-blob and CMYK-table broadcasts.
Each run must print one JSON line with a positive value, the workload's shard accounting and the objects the single-GPU line has."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "3", "--warmup", "1", "--batch", "64", "--extra-steps", "0", "--verify-images", "4", "--latency-requests", "0", "--config0-runs", "0"]


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_one_context_over_two_shards_of_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-context", "--gpus", "2", "--cpu-images", "8"] + QUICK,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and "error" not in line
    assert line["config"]["devices"] == [0, 0] and sum(line["config"]["shard_images"]) == 128
    assert min(line["config"]["shard_images"]) >= 60                       # byte-balanced: equal pictures, equal shards
    assert line["roofline"]["kernel"] == "resample_mfma_kernel" and 0 < line["roofline"]["frac"] < 1
    assert "24 bits" in line["dtype"]                                      # the full-width arithmetic is what the bench runs
    assert line["cpu_baseline"] and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert "traffic" in line["roofline"]                                   # (a number once profiles/traffic.json has this workload)


def test_two_gloo_ranks_share_the_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cpu-images", "0"] + QUICK,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and "error" not in line and line["scaling"] == "weak"
    assert line["verified_images"] == 4
    assert line["icc_lut_broadcast"]["ok"]                                 # rank 0's baked CMYK table reached rank 1 and converts identically
    assert line["config"]["images_per_gpu_per_step"] == 64
