"""CPU: `python bench.py --gpus N` with no launcher around it starts N ranks itself (bench.launch_ranks -> torch.distributed.run on 127.0.0.1),
the ranks rendezvous (gloo here: no GPU), run the stub step of --dry-run, take the barrier + max-over-ranks path, rank 0 prints ONE JSON
line with n_gpus = N and the launcher's exit code comes back.  A rehearsal of the launch path only: value is null, nothing is measured."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ, PPMS_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    e.pop("RANK", None), e.pop("WORLD_SIZE", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, cwd="/tmp", env=e, timeout=600)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n,T,scaling,par", [(2, 5, "weak", "replicas x2"), (8, 40, "strong", "frames sharded 5/GPU x8"), (8, 5, "weak", "replicas x8")])
def test_bench_gpus_n_launches_n_ranks(n, T, scaling, par):
    r = _bench("--gpus", str(n), "--T", str(T), "--steps", "2", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == n and out["dry_run"] and out["value"] is None and out["backend"] == "gloo"
    assert out["scaling"] == scaling and out["config"]["parallelism"] == par
    assert out["frames_over_ranks"] == (T if scaling == "strong" else n * T)          # an all-reduce over all N ranks ran
    if scaling == "weak":                # replicas: the extra frame-sharded window (config 4: T = 40) is planned on the same ranks
        sh = out["sharded"]
        assert sh["T"] == 40 and sh["iters"] == 20 and sh["frames_per_gpu"] == 40 // n and sh["frames_over_ranks"] == 40 and sh["ms_per_window"] is None
    else:
        assert out["sharded"] is None
    assert out["ms_per_step"] >= n * 1.0                                               # max over ranks: the slowest stub (rank N-1 sleeps N ms)


def test_bench_propagates_a_rank_failure():
    """--T 7 --gpus 2 is fine (replicas); a world / --gpus mismatch must fail loudly through the launcher."""
    r = _bench("--gpus", "2", "--dry-run", env=dict(RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_driver_command_shape_under_torchrun():
    """The driver's own command line: python -m torch.distributed.run ... bench.py --gpus N (RANK is set: no second launch)."""
    e = dict(os.environ, PPMS_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:                                      # a free port, as bench.launch_ranks takes one (a fixed one collides under pytest -n)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, cwd="/tmp", env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_extra_sharded_phase_failure_prints_the_replica_line_once_and_exits_nonzero(bad_rank):
    """bench.ExtraPhaseGuard: the extra frame-sharded window (N > 1, behind the replica measurement) fails on ONE rank while the others sit in a collective.
    Whichever rank it is, rank 0 prints its (complete) replica line exactly once with the failure under `sharded`, and the launcher comes back non-zero --
    quickly: the blocked ranks leave on the launcher's SIGTERM (read from the signal wake-up pipe by a helper thread), not after a collective timeout."""
    import time
    t0 = time.time()
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--inject-sharded-failure", str(bad_rank), "--sharded-timeout", "120")
    took = time.time() - t0
    assert r.returncode != 0, (r.stdout, r.stderr[-1500:])
    out = _line(r.stdout)
    assert out["n_gpus"] == 2 and out["ms_per_step"] > 0 and "error" in out["sharded"], out
    if bad_rank == 0:
        assert "injected failure on rank 0" in out["sharded"]["error"]
    else:
        assert "SIGTERM" in out["sharded"]["error"] or "injected" in out["sharded"]["error"], out["sharded"]
    assert took < 100, f"{took:.0f} s: the ranks waited for a timeout instead of leaving on the launcher's signal"
