"""GPU: bench.py's OWN N>1 code path on the one-GPU test box (round-2 VERDICT item 7): two ranks launched exactly as the driver
launches them (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...`), both on cuda:0 with gloo
carrying the collectives (SRBH_BENCH_SHARED_DEVICE / SRBH_BENCH_BACKEND test hooks) -- so that `_max_over_ranks`, the barrier-bracketed
timing, `comm.comm_ms` / `exposed_comm_ms` (GradReducer under the bench) and `Mosaic.reduce_to_` inside bench_predict are not executed
for the first time on the 8-GPU box.  Hardware scaling itself stays unmeasured until the driver's SCALE run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(extra, num_block="1", **more_env):
    env = dict(os.environ, **more_env)
    env.update(SRBH_BENCH_SHARED_DEVICE="1", SRBH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               # two processes cannot both own every CU: the persistent trunk kernel needs all its workgroups co-resident, so
               # the shared-GPU test runs the per-layer launch sequence (bit-identical, tests/test_gpu_rrdbnet.py)
               SRBH_PERSISTENT="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + (["--num-block", num_block] if num_block else []) + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_feature_two_ranks():
    d = _launch(["--steps", "3", "--warmup", "1", "--batch", "8", "--no-extras"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-3
    assert "cpu_baseline" not in d                    # N>1: no CPU leg


def test_bench_train_two_ranks_reports_comm():
    d = _launch(["--workload", "train", "--steps", "3", "--warmup", "2", "--batch", "4", "--no-extras"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    c = d["comm"]
    assert c["buckets"] >= 1 and c["grad_bytes"] > 80e6 and c["comm_ms"] > 0 and c["exposed_comm_ms"] >= 0
    assert d["final_loss"] == d["final_loss"]


def test_bench_train_two_ranks_graph_mode():
    """SRBH_TRAIN_GRAPH=1 at N > 1: forward + backward replayed as one graph per rank, the buckets all-reduced after it"""
    d = _launch(["--workload", "train", "--steps", "3", "--warmup", "5", "--batch", "4", "--no-extras"], SRBH_TRAIN_GRAPH="1")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert "one HIP graph" in d["config"]["parallelism"] and d["comm"]["buckets"] >= 1
    assert d["final_loss"] == d["final_loss"]


def test_bench_predict_two_ranks_merges_row_bands():
    d = _launch(["--workload", "predict", "--steps", "2", "--warmup", "1", "--batch", "64"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["tiles"] == 4494 + 536 and d["value"] > 0 and d["p50_city_latency_ms"] > 0


def test_bench_epoch_two_ranks_drop_last():
    """BASELINE configs[3] driver (`--workload epoch`, harness.train_epoch) as two data-parallel ranks: 70 tiles at batch 4 x 2 ranks
    = 8 steps, 64 tiles seen by the whole job (the ragged 6 dropped, train.py:97), gradients averaged through GradReducer."""
    d = _launch(["--workload", "epoch", "--epoch-tiles", "70", "--warmup", "2", "--batch", "4"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["steps"] == 8
    assert "70 synthetic train tiles" in d["config"]["workload"] and "drawn on the device" in d["data"]
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 8 / 1e3)) / d["value"] < 1e-3       # value = tiles seen / wall time
    assert d["comm"]["buckets"] >= 1 and d["final_loss"] == d["final_loss"]


def test_bench_exact_driver_command_two_ranks_with_extras():
    """The command the driver runs at N > 1 -- `bench.py --gpus 2 --steps K --warmup W`, NO other flag: 23 blocks, batch 32, the
    `train_step` and `predict` sub-objects ON -- as two gloo ranks sharing the one GPU (round-4 VERDICT item 5a).  Asserts that the
    one line parses, that `train_step.comm` exists (GradReducer ran under the extras), that `predict` merged the ranks' row bands and
    that the trailing `summary` carries the data-parallel fields a SCALE record needs.  Only the city count of the predict extra is
    bounded (SRBH_BENCH_PREDICT_CITIES test hook: 4 instead of 30) to keep two 23-block ranks on one GPU inside the test budget."""
    d = _launch(["--steps", "3", "--warmup", "1"], num_block=None, SRBH_BENCH_PREDICT_CITIES="4")
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 64 and d["value"] > 0
    assert "cpu_baseline" not in d
    t, p = d["train_step"], d["predict"]
    assert "error" not in t, t
    assert "error" not in p, p
    assert t["n_gpus"] == 2 and t["config"]["global_batch"] == 128 and t["value"] > 0
    c = t["comm"]
    assert c["buckets"] >= 1 and c["grad_bytes"] > 80e6 and c["comm_ms"] > 0 and c["exposed_comm_ms"] >= 0
    assert c["backend"] == "gloo" and c["world_size_seen"] == 2
    assert p["n_gpus"] == 2 and p["config"]["cities"] == 4 and p["value"] > 0 and "x2" in p["config"]["parallelism"]
    assert p["roofline"]["bound"] == "mfma" and 0 < p["roofline"]["frac"] < 1
    s = d["summary"]["dp_train"]
    assert s["n_gpus"] == 2 and s["world_size_seen"] == 2 and s["backend"] == "gloo"
    assert s["tiles_per_s"] == t["value"] and s["comm_ms"] == c["comm_ms"] and s["buckets"] == c["buckets"]
