"""bench.py at N > 1, as the driver will start it on the 8-GPU node (BASELINE.json configs[3]),
rehearsed here with two ranks sharing the box's one MI355X (MHX_DIST_BACKEND=gloo carries the bytes; sketching, shard
export and the merge run on the GPU exactly as under RCCL).  Checks the contract keys of the one JSON line, the gate
(`parity_on_sample`: sharded == unsharded), the exchange timing keys, and that the strong-scaling read set gives the
same sketch whatever the number of ranks."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def _bench(*argv, gloo=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if gloo:
        env["MHX_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    for key in CONTRACT:
        assert key in line, key
    return line


SMALL = ["--genome", "1000000", "--steps", "2", "--warmup", "1"]


@pytest.mark.parametrize("ksm", [(21, 1000, 1), (27, 5000, 3)])
def test_weak_scaling_line_two_ranks(ksm):
    k, s, m = ksm
    line = _bench("--gpus", "2", "--reads", "200000", "--k", str(k), "--s", str(s), "--m", str(m), *SMALL)
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["scaling"] == "weak"
    assert line["config"]["total_bases"] == 2 * 200000 * 150
    assert line["parity_on_sample"] is True, line["parity_kind"]
    ex = line["config"]["exchange_ms"]
    assert ex and all(key in ex for key in ("export_ms", "sizes_ms", "pack_ms", "gather_ms", "merge_ms", "total_ms"))
    assert len(line["config"]["exchange_entries_per_rank"]) == 2
    assert line["sketch_len"] == s
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


def test_strong_scaling_read_set_is_the_same_whatever_n(tmp_path):
    one, two = tmp_path / "one.npz", tmp_path / "two.npz"
    l1 = _bench("--gpus", "1", "--total-reads", "400000", "--no-cpu-baseline", "--dump-sketch", str(one), *SMALL)
    l2 = _bench("--gpus", "2", "--total-reads", "400000", "--dump-sketch", str(two), *SMALL)
    assert l1["scaling"] == l2["scaling"] == "strong"
    assert l1["config"]["total_bases"] == l2["config"]["total_bases"] == 400000 * 150
    assert l2["config"]["ranks"] == 2 and l2["parity_on_sample"] is True
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["hashes"], b["hashes"]) and np.array_equal(a["counts"], b["counts"])
    assert len(a["hashes"]) == 1000
