"""bench.py's multi-rank control flow on CPU: the driver launches `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` on an 8-GPU node, a path no 1-GPU lease ever executes.  With SDT_BENCH_STUB=1 the train step is replaced by a tiny
all-reduce and SDT_BENCH_BACKEND=gloo swaps the transport; everything else -- environment parsing, process-group set-up, the
barrier + timed loop, max over ranks, the single JSON line on rank 0, tear-down -- is the production code."""
import json
import os
import subprocess
import sys

from conftest import REPO
from test_dp_gloo import _free_port


def _run(nproc, extra=(), plain=False):
    env = dict(os.environ, SDT_BENCH_STUB="1", SDT_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(REPO, "bench.py")
    if nproc == 1:
        cmd = [sys.executable, bench, "--steps", "5", "--warmup", "2"]
    elif plain:  # as the driver types it on a multi-GPU node: no launcher, bench.py starts its own ranks
        cmd = [sys.executable, bench, "--gpus", str(nproc), "--steps", "5", "--warmup", "2"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench, "--gpus", str(nproc), "--steps", "5", "--warmup", "2"]
    out = subprocess.run(cmd + list(extra), env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_rank_control_flow_under_gloo():
    j = _run(2)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["stub"] is True
    assert j["config"]["global_batch"] == 64 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    # value = clips of ALL ranks / the slowest rank's wall time
    assert abs(j["value"] - 2 * 32 * 5 / (j["ms_per_step"] * 5e-3)) < 1e-6 * j["value"]
    assert j["median_ms_per_step"] > 0 and j["ms_per_step_uninstrumented"] > 0


def test_bench_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` typed as a plain command (it used to die on an assertion in 2 s)."""
    j = _run(2, plain=True)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and j["stub"] is True
    pr = j["per_rank"]
    assert pr["ranks"] == 2 and pr["backend"] == "gloo" and len(pr["elapsed_s"]) == 2 and len(pr["median_ms_per_step"]) == 2
    assert abs(max(pr["elapsed_s"]) * 1e3 / 5 - j["ms_per_step"]) < 1e-6 * j["ms_per_step"]


def test_bench_single_rank_defaults():
    j = _run(1)
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 32 and j["vs_baseline"] is None
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "dtype", "data", "median_ms_per_step"):
        assert k in j
