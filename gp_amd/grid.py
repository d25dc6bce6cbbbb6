"""Hyper-parameter grid of marginal-likelihood evaluations sharded over the GPUs of a node.

Each grid point (alpha, rho, sigma) is one independent evaluation of
models/fit_hyperparameters.stan:18-32 on the same (X, y); the reference's only
parallelism for such work is fork-per-draw (parallel::mclapply, pendulum_fit.R:268).
Here: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI),
point g goes to rank g mod P, no data-path collective; a single all_gather of
4 doubles per point (logml, sum log L_ii, z'z, info) at the end.
"""
import numpy as np


def shard_indices(G, rank, world):
    """Grid points owned by `rank`: g = rank, rank + world, ...  (equal cost: N is fixed)."""
    return np.arange(rank, G, world)


def _gpu_evaluate(X, y, alpha, rho, sigma, jitter):
    from ._lib import default_context
    out, info = default_context().logml_grid(X, y, alpha, rho, sigma, jitter)
    return out, info


def logml_grid_sharded(X, y, alpha, rho, sigma, jitter=0.0, evaluate=None, group=None):
    """All ranks call this with the same arguments; every rank returns the full
    (G, 3) results and (G,) info.  Without an initialised process group it evaluates
    everything locally.  `evaluate(X, y, alpha, rho, sigma, jitter) -> (out (g,3), info (g,))`
    defaults to the libgpmi GPU path (tests inject a checker to exercise the sharding on CPU)."""
    import torch
    import torch.distributed as dist

    alpha, rho, sigma = np.broadcast_arrays(np.asarray(alpha, float), np.asarray(rho, float), np.asarray(sigma, float))
    alpha, rho, sigma = alpha.ravel(), rho.ravel(), sigma.ravel()
    G = alpha.size
    evaluate = evaluate or _gpu_evaluate
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate(X, y, alpha, rho, sigma, jitter)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(G, rank, world)
    per = (G + world - 1) // world
    local = np.full((per, 4), np.nan)
    if mine.size:
        out, info = evaluate(X, y, alpha[mine], rho[mine], sigma[mine], jitter)
        local[: mine.size, :3] = out
        local[: mine.size, 3] = info
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    send = torch.from_numpy(local).to(dev)
    parts = [torch.empty((per, 4), dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(parts, send, group=group)  # the path's only collective: 32 B per grid point
    recv = torch.stack(parts).cpu().numpy()
    res = np.full((G, 3), np.nan)
    info = np.zeros(G, dtype=np.int32)
    for r in range(world):
        idx = shard_indices(G, r, world)
        res[idx] = recv[r, : idx.size, :3]
        info[idx] = recv[r, : idx.size, 3].astype(np.int32)
    return res, info
