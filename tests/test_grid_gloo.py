"""CPU: the N>1 sharding + gather path with world_size 2 over gloo.  The evaluator injected
here is the oracle (checker); on a GPU box the default evaluator is libgpmi."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_eval(X, y, alpha, rho, sigma, jitter):
    from oracle import oracle as orc
    out = np.empty((len(alpha), 3)); info = np.zeros(len(alpha), dtype=np.int32)
    for g in range(len(alpha)):
        lm, sld, q, inf = orc.logml(X, y, alpha[g], rho[g], sigma[g], jitter)
        out[g] = (lm, sld, q); info[g] = inf
    return out, info


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gp_amd.grid import logml_grid_sharded
        from oracle import oracle as orc
        X, y = orc.synth(40, 2)
        rho = np.repeat(np.geomspace(0.1, 1.0, 3), 3)[:7]; sig = np.tile(np.geomspace(0.05, 0.5, 3), 3)[:7]
        sig = sig.copy(); sig[4] = 1e-9; rho = rho.copy(); rho[4] = 50.0  # one non-PD point: NaN, grid continues
        res, info = logml_grid_sharded(X, y, np.ones(7), rho, sig, 0.0, evaluate=_oracle_eval)
        q.put((rank, res, info))
    finally:
        dist.destroy_process_group()


def test_grid_sharded_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import oracle as orc
    X, y = orc.synth(40, 2)
    rho = np.repeat(np.geomspace(0.1, 1.0, 3), 3)[:7].copy(); sig = np.tile(np.geomspace(0.05, 0.5, 3), 3)[:7].copy()
    sig[4] = 1e-9; rho[4] = 50.0
    want, winfo = _oracle_eval(X, y, np.ones(7), rho, sig, 0.0)
    for rank, res, info in got:
        np.testing.assert_array_equal(info, winfo)
        np.testing.assert_allclose(res, want, rtol=0, atol=0, equal_nan=True)
    assert winfo[4] > 0 and np.isnan(want[4, 0])


def test_bench_launcher_spawns_ranks_and_gathers():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts the two ranks itself (child processes made
    before anything touches a GPU), they rendezvous over gloo, shard the 64-point c4 grid g -> g mod 2 and
    all_gather it -- the bench's own N>1 path, with a closed-form stand-in for the evaluation (--dry-run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launched_by_bench"] and rec["results_ok"] and rec["grid_points"] == 64
    # a failing rank fails the launcher (no silent success)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "nope"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
