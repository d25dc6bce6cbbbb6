"""Hyper-parameter grid of marginal-likelihood evaluations sharded over the GPUs of a node.

Each grid point (alpha, rho, sigma) is one independent evaluation of
models/fit_hyperparameters.stan:18-32 on the same (X, y); the reference's only
parallelism for such work is fork-per-draw (parallel::mclapply, pendulum_fit.R:261-268).
Here: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI),
point g goes to rank g mod P, no data-path collective; a single all_gather of
4 doubles per point (logml, sum log L_ii, z'z, info) at the end.

ONE sharding implementation (`_sharded`) serves both entry points:
  logml_grid_sharded      host buffers in, numpy out (what an R wrapper would call)
  logml_grid_sharded_dev  device pointers in, device tensors out; nothing crosses PCIe and nothing
                          synchronises with the host: the evaluations, the packing of the results and
                          the all_gather are enqueued on one stream (what bench.py times)
"""
import numpy as np


def shard_indices(G, rank, world):
    """Grid points owned by `rank`: g = rank, rank + world, ...  (equal cost: N is fixed)."""
    return np.arange(rank, G, world)


def _bcast3(alpha, rho, sigma):
    alpha, rho, sigma = np.broadcast_arrays(np.asarray(alpha, float), np.asarray(rho, float), np.asarray(sigma, float))
    return alpha.ravel(), rho.ravel(), sigma.ravel()


def _sharded(G, eval_local, group, comm_device):
    """The sharding itself.  eval_local(mine, per) -> (per, 4) float64 torch tensor of this rank's rows
    (logml, sum log L_ii, z'z, info; rows past mine.size are padding).  One all_gather over `group`;
    returns the (G, 4) tensor in grid order on `comm_device` (identical on every rank).  The tensor
    is moved to `comm_device` for the collective only when it is not already there (gloo rehearsal of
    a GPU run: host tensors)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(G, rank, world)
    per = (G + world - 1) // world
    send = eval_local(mine, per)
    if send.device != comm_device:
        send = send.to(comm_device)
    recv = torch.empty((world, per, 4), dtype=torch.float64, device=comm_device)
    dist.all_gather([recv[r] for r in range(world)], send.contiguous(), group=group)  # the path's only collective: 32 B per grid point
    # point g = k * world + r sits at recv[r, k]
    return recv.permute(1, 0, 2).reshape(per * world, 4)[:G]


def _comm_device(group, device=None):
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


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

    alpha, rho, sigma = _bcast3(alpha, rho, sigma)
    G = alpha.size
    evaluate = evaluate or _gpu_evaluate
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate(X, y, alpha, rho, sigma, jitter)

    def eval_local(mine, per):
        local = np.full((per, 4), np.nan)
        if mine.size:
            out, info = evaluate(X, y, alpha[mine], rho[mine], sigma[mine], jitter)
            local[: mine.size, :3] = out
            local[: mine.size, 3] = info
        return torch.from_numpy(local)

    full = _sharded(G, eval_local, group, _comm_device(group)).cpu().numpy()
    return np.ascontiguousarray(full[:, :3]), full[:, 3].astype(np.int32)


def logml_grid_local_dev(ctx, dX_ptr, n, ldx, D, dy_ptr, alpha, rho, sigma, jitter=0.0, device=None, rows=None):
    """This process's GPU only: the given points through gpmi_logml_grid_dev (lanes of `ctx`) on torch's
    CURRENT stream of `device`; returns the (rows or G, 4) float64 device tensor [logml, sum log L_ii, z'z,
    info] (rows past G are NaN padding).  The building block of logml_grid_sharded_dev."""
    import torch

    alpha, rho, sigma = _bcast3(alpha, rho, sigma)
    G = alpha.size
    rows = G if rows is None else rows
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    out = torch.full((rows, 3), float("nan"), dtype=torch.float64, device=dev)
    info = torch.zeros(rows, dtype=torch.int32, device=dev)
    if G:
        ctx.logml_grid_dev(dX_ptr, n, ldx, D, dy_ptr, alpha, rho, sigma, jitter, out.data_ptr(), info.data_ptr())
    return torch.cat([out, info.to(torch.float64).unsqueeze(1)], dim=1)


def logml_grid_sharded_dev(ctx, dX_ptr, n, ldx, D, dy_ptr, alpha, rho, sigma, jitter=0.0, group=None, device=None,
                           comm_device=None):
    """Device-resident variant: X (n x D column-major, leading dimension ldx) and y live in HBM on
    `device` (default: the current cuda device); this rank's points run through gpmi_logml_grid_dev on the
    lanes of `ctx`, the results are packed on the device and gathered with ONE all_gather.  Returns the
    (G, 4) float64 tensor [logml, sum log L_ii, z'z, info] in grid order on `comm_device` (default: the
    device for nccl/RCCL, the host for gloo), identical on every rank.  Everything is enqueued on torch's
    CURRENT stream of `device` (the context is switched to it): no host synchronisation happens here
    unless the collective itself runs on host tensors.  Without a process group: all points locally."""
    import torch
    import torch.distributed as dist

    alpha, rho, sigma = _bcast3(alpha, rho, sigma)
    G = alpha.size
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

    def eval_local(mine, per):
        return logml_grid_local_dev(ctx, dX_ptr, n, ldx, D, dy_ptr, alpha[mine], rho[mine], sigma[mine], jitter, dev, per)

    if not (dist.is_available() and dist.is_initialized()):
        return eval_local(np.arange(G), G)
    cdev = comm_device if comm_device is not None else _comm_device(group, dev)
    return _sharded(G, eval_local, group, cdev)
