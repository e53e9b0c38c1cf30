"""``bench.py --gpus N`` must start N ranks by itself: proven on the GPU-less box with ``--dry-run-dist``
(gloo, a stand-in step through ``stheno_amd.dist.sharded_logpdf`` -- the launch, rendezvous, barrier / MAX-over-ranks timing,
all-gather and the one JSON line are the code the MI355X run uses; reference semantics of the sharded batch:
``tests/model/test_cases.py:134-155``)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import ROOT


def _run(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=env,
                         cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]            # ONE JSON line, whatever the number of ranks
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_spawns_that_many_ranks(n):
    out = _run("--gpus", str(n), "--dry-run-dist", "--steps", "3", "--warmup", "1")
    assert out["dry_run"] is True and out["n_gpus"] == n and out["world_size"] == n and out["backend"] == "gloo"
    assert out["steps"] == 3 and out["warmup"] == 1 and out["value"] > 0 and out["ms_per_step"] > 0
    b = out["batched"]
    assert b["n_gpus"] == n and b["scaling"] == "strong" and b["unit"] == "GPs/s"
    assert (b["allgather_us"] is not None) == (16 % n == 0)      # (the stand-in batch of 16 splits evenly over 2 ranks, raggedly over 3)


def test_eight_ranks_with_a_batch_they_do_not_divide():
    """The shape of the first real 8-GPU run: eight ranks, a batch that is no multiple of eight -- shards of 3 and 2
    GPs, the padded all-gather of ``dist.sharded_logpdf``, one JSON line."""
    out = _run("--gpus", "8", "--dry-run-dist", "--dry-run-gps", "20", "--steps", "2", "--warmup", "1")
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["world_size"] == 8 and out["backend"] == "gloo"
    b = out["batched"]
    assert b["n_gpus"] == 8 and b["value"] > 0 and b["allgather_us"] is None      # (the timed all-gather is the even-split one)
    assert "20 stand-in GPs" in b["metric"]


def test_one_rank_dry_run_needs_no_process_group():
    out = _run("--dry-run-dist", "--steps", "2", "--warmup", "0")
    assert out["n_gpus"] == 1 and out["world_size"] == 0 and out["batched"]["allgather_us"] is None


def test_rank_count_mismatch_is_refused():
    env = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-dist"], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, **env), cwd=ROOT)
    assert res.returncode != 0 and "launcher started 1 ranks" in (res.stderr + res.stdout)
