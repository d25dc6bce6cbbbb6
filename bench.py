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


def cpu_lapack(n_sample, D, threads):
    """Best-effort CPU leg: the same evaluation with numpy + LAPACK
    dpotrf / dtrtrs (scipy, OpenBLAS) on `threads` host threads -- an upper bound for what a CPU
    build of the reference's path could reach, next to the single-threaded restatement."""
    import scipy.linalg as sla
    from threadpoolctl import threadpool_limits
    from oracle import oracle as orc
    X, y = orc.synth(n_sample, D)
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        G = X @ X.T
        sq = np.diag(G).copy()
        G *= -2.0
        G += sq[:, None]
        G += sq[None, :]
        np.maximum(G, 0.0, out=G)
        G *= -0.5 / (0.3 * 0.3)
        np.exp(G, out=G)
        G[np.diag_indices(n_sample)] += 0.1 * 0.1
        L = sla.cholesky(G, lower=True, overwrite_a=True, check_finite=False)
        z = sla.solve_triangular(L, y, lower=True, check_finite=False)
        sld = float(np.sum(np.log(np.diag(L))))
        lm = -0.5 * float(z @ z) - sld - 0.5 * n_sample * np.log(2.0 * np.pi)
        dt = time.perf_counter() - t0
    return dt, lm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--nb-outer", type=int, default=0)
    ap.add_argument("--cpu-sample-n", type=int, default=8192)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"],
                    help="c3 (default, the metric): N=16384 D=3 evaluations, weak scaling; c4: the 64-point "
                         "rho x sigma grid at N=8192 sharded over the ranks (strong scaling); c5: derivative "
                         "joint [y, y'] covariance, N=8192 (matrix order 16384)")
    ap.add_argument("--grid-lanes", type=int, default=0, help="concurrent evaluations per GPU in c4 (0 = auto)")
    ap.add_argument("--lookahead", type=int, default=-1, choices=[-1, 0, 1],
                    help="panel look-ahead inside one factorisation: 1 on; 0 / -1 (default) off")
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
    if args.grid_lanes:
        ctx.set_option("grid_lanes", args.grid_lanes)
    ctx.reserve(2 * n if args.workload == "c5" else n)  # workspaces of every grid lane, outside the timed region

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

    if args.grid_lanes:
        ctx.set_option("grid_lanes", args.grid_lanes)
    ctx.set_option("lookahead", args.lookahead)

    def run_points(lo, hi):
        """Evaluate points lo..hi-1 of this rank.  c3 / c4 go through the grid entry point, which
        overlaps independent points on internal lanes (own workspaces and streams)."""
        if args.workload == "c5":
            for p in range(lo, hi):
                ctx.joint_logml_dev(dX.data_ptr(), n, dy.data_ptr(), 1.0, rho[p], sig[p], 1e-6,
                                    dout[p].data_ptr(), dinfo[p:].data_ptr())
        else:
            ctx.logml_grid_dev(dX.data_ptr(), n, n, D, dy.data_ptr(), np.ones(hi - lo), rho[lo:hi], sig[lo:hi], 0.0,
                               dout[lo].data_ptr(), dinfo[lo:].data_ptr())

    if warm:
        run_points(0, warm * per_step)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_points(warm * per_step, (warm + steps) * per_step)  # exactly `steps` steps
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

    # Roofline pass (same run, rank 0 only, after the timed region): the same evaluations one at
    # a time with HIP-event pairs around every covariance-build and trailing-update launch on
    # the launch stream.  Done separately because concurrent lanes overlap launches, which makes
    # per-launch durations meaningless inside the throughput region.
    kt = {"syrk": (0, 0.0, 0.0), "build": (0, 0.0, 0.0)}
    seq_ms = None
    if rank == 0:
        ctx.set_option("grid_lanes", 1)
        ctx.set_option("lookahead", 0)   # nothing else on the chip while a bracketed launch runs
        ctx.set_option("kernel_timing", 1)
        ctx.kernel_timing(reset=True)
        nprof = min(steps * per_step, 4)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        lo = warm * per_step
        for p in range(lo, lo + nprof):
            run_points(p, p + 1)
        torch.cuda.synchronize(dev)
        seq_ms = 1e3 * (time.perf_counter() - t1) / nprof
        kt = ctx.kernel_timing(reset=True)
        ctx.set_option("kernel_timing", 0)
        ctx.set_option("lookahead", args.lookahead)
        ctx.set_option("grid_lanes", args.grid_lanes)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    res = dout.cpu().numpy()
    info = dinfo.cpu().numpy()
    ok = bool(np.all(info == 0) and np.all(np.isfinite(res[:, 0])))

    def pmc_traffic(kernel_substr):
        """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes over this same
        command (tools/pmc_bench.sh; FETCH_SIZE and WRITE_SIZE in separate passes, FETCH doubled as
        MI355X_MICROARCH.md prescribes for gfx950).  None when no PMC summary is committed."""
        path = os.path.join(ROOT, "profiles", "r01_pmc_bench_%s.json" % args.workload)
        try:
            with open(path) as f:
                for name, e in json.load(f).items():
                    if kernel_substr in name and "hbm_bytes_per_launch" in e:
                        return e["hbm_bytes_per_launch"], e.get("mfma_util"), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            pass
        return None, None, None

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
                       "N": n, "D": D, "alpha": 1.0, "rho": 0.3, "sigma": 0.1, "nb_outer": args.nb_outer or "auto(%d)" % (lambda nf: 1024 if nf >= 12288 else 512 if nf >= 6144 else 256)(2 * n if args.workload == "c5" else n),
                       "parallelism": "independent hyper-parameter points per GPU; one RCCL all_gather of results"},
            "results_ok": ok,
            "grid_lanes": args.grid_lanes or "auto(4)",
            "ms_per_eval_sequential": seq_ms,
            "logml_first": float(res[warm * per_step, 0]),
            "cholesky_tflops_per_gpu_whole_eval": chol_flops * evals / world / elapsed / 1e12,
            "roofline": {
                "kernel": "k_gemm_nt<1> (trailing-update SYRK, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "measured_in": "instrumented sequential pass of the same evaluations in this run "
                               "(lanes=1, look-ahead off: bracketed launches run alone on the chip)",
                "achieved": ach,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / FP64_PEAK_TFLOPS,
                "traffic": pmc_traffic("k_gemm_nt<1>")[0],
                "traffic_unit": "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes)",
                "traffic_source": pmc_traffic("k_gemm_nt<1>")[2],
                "mfma_util_pmc": pmc_traffic("k_gemm_nt<1>")[1],
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
                "traffic": pmc_traffic("k_joint_cov" if args.workload == "c5" else "k_se_cov<3>")[0],
                "algorithmic_bytes_per_launch": build_bytes / max(build_n, 1),
                "avg_launch_ms": build_ms / max(build_n, 1),
            },
        }
        if world == 1:
            # one large stand-alone launch of the same kernel (m = 14336, K = 1024: 12.4 rounds of tiles),
            # HIP events around 6 back-to-back launches: the per-launch rate without the small
            # trailing matrices that pull the whole-factorisation average down
            try:
                ctx.set_stream(None)
                ctx.probe_syrk(14336, 1024, 2)
                ms_l, tf_l = ctx.probe_syrk(14336, 1024, 6)
                line["roofline"]["standalone_launch"] = {"m": 14336, "k": 1024, "ms": ms_l, "achieved": tf_l,
                                                         "frac": tf_l / FP64_PEAK_TFLOPS}
            except Exception as e:  # diagnostic only
                line["roofline"]["standalone_launch"] = {"error": repr(e)}
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
            try:
                thr = min(16, os.cpu_count() or 1)   # the CPU share of a one-GPU box
                dt2, lm2 = cpu_lapack(ns, D, thr)
                line["cpu_baseline"]["lapack_multicore"] = {
                    "value": 1.0 / (dt2 * scale), "unit": "evals/s", "cores": thr,
                    "sample": "numpy + LAPACK dpotrf/dtrtrs (scipy, OpenBLAS) at N=%d (%.2f s), scaled by "
                              "(N/Ns)^3; upper bound for a CPU build, not the reference's path" % (ns, dt2),
                    "sample_rel_err_vs_oracle": abs(lm2 - lm_cpu) / abs(lm_cpu)}
            except Exception as e:  # optional leg: never fails the bench line
                line["cpu_baseline"]["lapack_multicore"] = {"error": repr(e)}
        print(json.dumps(line))
        sys.stdout.flush()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
