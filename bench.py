#!/usr/bin/env python
"""Hot-path benchmark: exact-GP log-marginal-likelihood evaluations per second.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full evaluation of models/fit_hyperparameters.stan:18-32 on synthetic
inputs already resident in HBM: SE covariance build (N x N, fused sigma^2 diagonal) ->
blocked fp64 Cholesky -> forward solve -> log-det -> scalar.  N = 1 runs BASELINE
config c3 (N=16384, D=3); with N > 1 GPUs every rank evaluates its own K hyper-parameter
points on the same data (weak scaling, no data-path collective) and the per-point results
are gathered once over RCCL at the end.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 (matrix == vector); not in the skills guides
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(n_sample, D, cores=1):
    """Oracle (CPU restatement of the reference path, 1 thread like rstan/Eigen) timed on a
    bounded sample: one full evaluation at N = n_sample, scaled to the N=16384 metric by
    the N^3/3 Cholesky flop count (the dominant term)."""
    from oracle import oracle as orc
    X, y = orc.synth(n_sample, D)
    t0 = time.perf_counter()
    lm = orc.logml(X, y, 1.0, 0.3, 0.1)
    dt = time.perf_counter() - t0
    return dt, lm[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--nb-outer", type=int, default=0)
    ap.add_argument("--cpu-sample-n", type=int, default=6144)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"],
                    help="c3 (default, the metric): N=16384 D=3 evaluations, weak scaling; c4: the 64-point "
                         "rho x sigma grid at N=8192 sharded over the ranks (strong scaling); c5: derivative "
                         "joint [y, y'] covariance, N=8192 (matrix order 16384)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import gp_amd
    ctx = gp_amd.Context(local_rank)
    n, D = args.n, args.d
    if args.workload == "c4":
        n = 8192
    elif args.workload == "c5":
        n, D = 8192, 1
    if args.nb_outer:
        ctx.set_option("nb_outer", args.nb_outer)
    ctx.reserve(n)

    # deterministic synthetic inputs, generated on the host once and
    # moved to HBM before the timed region
    from gp_amd.synth import synth
    X, y = synth(n, D)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)   # (D, n) row-major == n x D column-major
    dy = torch.from_numpy(y).to(dev)
    steps, warm = args.steps, args.warmup
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    cdev = torch.device("cpu") if args.rehearse else dev  # gloo rehearsal: collectives on host tensors

    if args.workload == "c3":
        # rank r, step k evaluates its own hyper-parameter point near (rho, sigma) = (0.3, 0.1)
        per_step = 1
        npts = steps + warm
        rho = 0.3 * (1.0 + 0.01 * ((np.arange(npts) * world + rank) % 16))
        sig = 0.1 * np.ones(npts)
    elif args.workload == "c4":
        # 8 x 8 (rho, sigma) grid, log-spaced; point g -> rank g mod world
        R, S = np.meshgrid(np.geomspace(0.1, 1.0, 8), np.geomspace(0.05, 0.5, 8), indexing="ij")
        mine = np.arange(rank, 64, world)
        per_step = mine.size
        npts = per_step * (steps + warm)
        rho = np.tile(R.ravel()[mine], steps + warm)
        sig = np.tile(S.ravel()[mine], steps + warm)
    else:
        per_step = 1
        npts = steps + warm
        t_host = np.linspace(0.0, 10.0, n)
        dX = torch.from_numpy(t_host).to(dev)
        dy = torch.from_numpy(np.concatenate([np.sin(t_host), np.cos(t_host)])).to(dev)
        rho = 0.5 * (1.0 + 0.01 * (np.arange(npts) % 8))
        sig = 0.1 * np.ones(npts)
    dout = torch.zeros((npts, 3), dtype=torch.float64, device=dev)
    dinfo = torch.zeros(npts, dtype=torch.int32, device=dev)

    def run(k):  # one step = per_step evaluations
        for p in range(k * per_step, (k + 1) * per_step):
            if args.workload == "c5":
                ctx.joint_logml_dev(dX.data_ptr(), n, dy.data_ptr(), 1.0, rho[p], sig[p], 1e-6,
                                    dout[p].data_ptr(), dinfo[p:].data_ptr())
            else:
                ctx.logml_dev(dX.data_ptr(), n, n, D, dy.data_ptr(), 1.0, [rho[p]], sig[p], 0.0,
                              dout[p].data_ptr(), dinfo[p:].data_ptr())

    for k in range(warm):
        run(k)
    torch.cuda.synchronize(dev)
    ctx.set_option("kernel_timing", 1)
    ctx.kernel_timing(reset=True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        run(k)
    if distributed:
        # the path's only collective: gather the per-point results (3 doubles per point)
        send = dout.to(cdev)
        gathered = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(gathered, send)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kt = ctx.kernel_timing(reset=True)
    ctx.set_option("kernel_timing", 0)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    res = dout.cpu().numpy()
    info = dinfo.cpu().numpy()
    ok = bool(np.all(info == 0) and np.all(np.isfinite(res[:, 0])))

    if rank == 0:
        evals = steps * (64 if args.workload == "c4" else world * per_step)
        value = evals / elapsed
        order = 2 * n if args.workload == "c5" else n
        syrk_n, syrk_ms, syrk_flops = kt["syrk"]
        build_n, build_ms, build_bytes = kt["build"]
        ach = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        chol_flops = order ** 3 / 3.0
        line = {
            "metric": {"c3": "gp_logml_evals_per_sec_N%d_D%d" % (n, D),
                       "c4": "gp_logml_grid64_evals_per_sec_N%d_D%d" % (n, D),
                       "c5": "gp_joint_deriv_logml_evals_per_sec_N%d" % n}[args.workload],
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warm,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "c4" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": {
                "c3": "c3: exact GP log marginal likelihood, N=%d, D=%d, SE kernel build + fp64 Cholesky + "
                      "solve + log-det, 1 hyper-parameter point per step per GPU" % (n, D),
                "c4": "c4: 64-point (rho x sigma) grid at N=%d, D=%d sharded over the ranks; 1 step = the "
                      "whole grid" % (n, D),
                "c5": "c5: derivative joint [y, y'] covariance, N=%d (matrix order %d), build + fp64 Cholesky "
                      "+ solve + log-det" % (n, 2 * n)}[args.workload],
                       "N": n, "D": D, "alpha": 1.0, "rho": 0.3, "sigma": 0.1, "nb_outer": args.nb_outer or "auto(512)",
                       "parallelism": "independent hyper-parameter points per GPU; one RCCL all_gather of results"},
            "results_ok": ok,
            "logml_first": float(res[warm * per_step, 0]),
            "cholesky_tflops_per_gpu_whole_eval": chol_flops * evals / world / elapsed / 1e12,
            "roofline": {
                "kernel": "k_gemm_nt<1> (trailing-update SYRK, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "achieved": ach,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / FP64_PEAK_TFLOPS,
                "traffic": None,
                "launches": int(syrk_n),
                "avg_launch_ms": syrk_ms / max(syrk_n, 1),
                "flops_per_launch_avg": syrk_flops / max(syrk_n, 1),
            },
            "roofline_build": {
                "kernel": "k_joint_cov (lower-triangular joint [y, y'] covariance build)" if args.workload == "c5"
                else "k_se_cov<3> (lower-triangular SE covariance build)",
                "bound": "hbm",
                "achieved": build_bytes / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (build_bytes / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if build_ms > 0 else 0.0,
                "traffic": None,
                "avg_launch_ms": build_ms / max(build_n, 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            ns = args.cpu_sample_n
            dt, lm_cpu = cpu_baseline(ns, D)
            # parity of the sample size on the GPU, for the record
            Xs, ys = synth(ns, D)
            ctx.set_stream(None)
            lm_gpu = ctx.logml(Xs, ys, 1.0, [0.3], 0.1)[0]
            scale = (float(n) / ns) ** 3
            line["cpu_baseline"] = {
                "value": 1.0 / (dt * scale),
                "unit": "evals/s",
                "cores": 1,
                "kind": "port",
                "sample": "one full evaluation at N=%d, D=%d by the single-threaded oracle (%.2f s), "
                          "scaled to N=%d by (N/Ns)^3" % (ns, D, dt, n),
                "sample_seconds": dt,
                "sample_rel_err_gpu_vs_cpu": abs(lm_gpu - lm_cpu) / abs(lm_cpu),
                "host_cores_available": os.cpu_count(),
            }
        print(json.dumps(line))
        sys.stdout.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
