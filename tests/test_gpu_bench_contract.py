"""GPU: bench.py keeps the driver's contract — one JSON line from rank 0 with the metric,
the roofline object of the query kernel and (N = 1) the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C1", "--steps", "3", "--warmup", "2", "--reads", "60000", "--batch", "4096"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "phases"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["unit"] == "reads/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 3 * 4096 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert rf["bytes_per_probe"] == 128 and rf["line_rate_Gprobes_per_s"] > 0
    ph = d["phases"]
    assert ph["head"]["reads"] + ph["steady"]["reads"] == 3 * 4096 and d["config"]["reads_timed"] == 3 * 4096
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "reads/s" and cb["sample"]


def test_two_ranks_on_the_one_gpu_end_where_one_rank_ends():
    """The N > 1 path end to end on the one available GPU (two processes share it): sharded fill
    merged by reduce-scatter + all-gather, batches with striped queries, striped streaming
    windows, the shared-memory exchange.  The run's counters (tiles, hits, misses, inserted
    bases, IDs, inserts) and the filter's population must equal the single-rank run's — not
    just agree between the ranks."""
    common = ["--config", "C1", "--steps", "4", "--reads", "120000", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    # round 6: no launcher around it — `bench.py --gpus 2` starts its two ranks itself (bench.py launch_ranks)
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--verify-ranks"] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1
    assert a["aux"]["pop"] == b["aux"]["pop"]
    assert a["aux"]["counters"] == b["aux"]["counters"]
    assert a["aux"]["counters"]["inserts"] > 1000 and b["aux"]["timed"]["batches"] > 10
    # round 5: the merge of the sharded fill is the product's (csrc/host/gr_ranks.cpp: staged here, the ranks share the
    # device; RCCL where every rank has its own), and the ranks' striped windows apply inserts inside their launches
    assert b["aux"]["comm"]["world"] == 2 and b["aux"]["comm"]["merge"].startswith("staged"), b["aux"]["comm"]
    assert b["aux"]["timed"]["stream_inserts"] > 0, b["aux"]["timed"]
