"""GPU: the product's multi-GPU entry point (gp_amd.grid.logml_grid_sharded_dev) -- what bench.py's c4
sub-record and `--workload c4` call -- on ONE GPU: in-process without a process group, and as a two-rank
rehearsal (`bench.py --gpus 2 --rehearse`: gloo, both ranks on cuda:0) started as a fresh child process.
Replaces the reference's fork-per-draw loop, pendulum_fit.R:261-268."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_dev_without_process_group_equals_grid_entry_point(ctx):
    import torch
    from gp_amd.grid import logml_grid_sharded_dev
    from gp_amd.synth import synth
    n, D = 1500, 3
    X, y = synth(n, D)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    dy = torch.from_numpy(y).to(dev)
    rho = np.geomspace(0.1, 1.0, 7); sig = np.geomspace(0.05, 0.5, 7)
    rho[3] = 50.0; sig[3] = 1e-9   # one non-PD point: NaN + info, the grid continues
    try:
        full = logml_grid_sharded_dev(ctx, dX.data_ptr(), n, n, D, dy.data_ptr(), np.ones(7), rho, sig, 0.0, device=dev)
        torch.cuda.synchronize(dev)
    finally:
        ctx.set_stream(None)
    got = full.cpu().numpy()
    want, info = ctx.logml_grid(X, y, np.ones(7), rho, sig)
    np.testing.assert_array_equal(got[:, :3], want)   # bit for bit (NaN == NaN here)
    np.testing.assert_array_equal(got[:, 3].astype(np.int32), info)
    assert info[3] > 0 and np.isnan(got[3, 0])


def test_bench_two_rank_rehearsal_runs_the_products_sharded_grid():
    """`python bench.py --gpus 2 --rehearse` as a fresh child process: the launcher starts two ranks, both on
    cuda:0, gloo rendezvous; the c3 line's c4 sub-record is produced by gp_amd.grid.logml_grid_sharded_dev
    and its sharded results equal rank 0's one-rank evaluation of all 64 points bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    print("rehearsal: n_gpus %d, %.2f evals/s, c4 %s" % (rec["n_gpus"], rec["value"], json.dumps(rec["c4"])))
    assert rec["n_gpus"] == 2 and rec["results_ok"]
    c4 = rec["c4"]
    assert c4["n_gpus"] == 2 and c4["grid_points"] == 64 and c4["results_ok"]
    assert c4["entry_point"] == "gp_amd.grid.logml_grid_sharded_dev"
    assert c4["sharded_results_bit_identical_to_one_rank"] is True
