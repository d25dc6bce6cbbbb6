#!/usr/bin/env python
"""Hot-path benchmark: exact-GP log-marginal-likelihood evaluations per second.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full evaluation of models/fit_hyperparameters.stan:18-32 on synthetic
inputs already resident in HBM: SE covariance build (N x N, fused sigma^2 diagonal) ->
blocked fp64 Cholesky -> forward solve -> log-det -> scalar.  N = 1 runs BASELINE
config c3 (N=16384, D=3); with N > 1 GPUs every rank evaluates its own K hyper-parameter
points on the same data (weak scaling, no data-path collective) and the per-point results
are gathered once over RCCL at the end.  Prints ONE JSON line on rank 0.

Launching: under torchrun (WORLD_SIZE set) this process is one rank.  A plain
`python bench.py --gpus N` with N > 1 starts the N ranks itself -- child processes, one per
GPU, created BEFORE this process imports torch or touches HIP (a process that has
initialised the GPU is never forked or re-exec'ed) -- and exits with their status.  This
replaces the reference's fork-per-draw (parallel::mclapply, pendulum_fit.R:261-268).

The line also carries a `c4` sub-record: BASELINE config c4 (the 64-point rho x sigma grid at
N=8192) sharded over the same ranks, its wall time, and the speed-up over ONE rank evaluating
all 64 points in the same run (strong scaling), with the sharded results compared bit for bit.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 (matrix == vector); not in the skills guides
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args(argv=None):
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
    ap.add_argument("--grid-lanes", type=int, default=0, help="concurrent evaluations per GPU (0 = auto)")
    ap.add_argument("--syrk-order", type=int, default=-1, choices=[-1, 0, 2],
                    help="tile walk of the multi-round trailing updates: 0 row-major, 2 XCD-partitioned bands, -1 library default")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on cuda:0")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / sharding / gather only (gloo, no GPU): the CPU test of the N>1 path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the c4 strong-scaling sub-record")
    ap.add_argument("--no-c1", action="store_true", help="skip the c1 / small-N sub-record (PMC passes: keeps per-kernel averages clean)")
    return ap.parse_args(argv)


# ---- launcher (parent process: no torch, no HIP) ---------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(ngpus, argv):
    """Start `ngpus` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment) and wait for them.  The parent never initialises the GPU.  Returns the exit status:
    0 only if every rank succeeded; when one rank fails the others are stopped (by pid)."""
    assert "torch" not in sys.modules, "the launcher must run before torch is imported"
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, GPMI_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                time.sleep(2.0)  # let the peers notice the broken rendezvous by themselves first
                for q in live:
                    if q.poll() is None:
                        q.terminate()
    return rc


# ---- CPU legs -------------------------------------------------------------------------------------------
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


# ---- PMC summaries committed under profiles/ -----------------------------------------------------------
def pmc_lookup(workload, order, kernel_substr):
    """(hbm_bytes_per_launch, mfma_util, file) of a kernel from the newest committed rocprofv3 --pmc
    summary taken over THIS workload at THIS matrix order (profiles/rNN_pmc_bench_<workload>_n<order>.json;
    tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  (None, None, None) when no such summary exists -- a
    figure measured at another size is never attributed to this run."""
    pat = os.path.join(ROOT, "profiles", "r*_pmc_bench_%s_n%d.json" % (workload, order))
    for path in sorted(glob.glob(pat), reverse=True):
        try:
            with open(path) as f:
                for name, e in json.load(f).items():
                    if kernel_substr in name:
                        return e.get("hbm_bytes_per_launch"), e.get("mfma_util"), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None, None


C4_N, C4_D, C4_G = 8192, 3, 64


def c4_grid():
    """8 x 8 (rho, sigma) grid, log-spaced; point g = 8 i_rho + i_sigma."""
    R, S = np.meshgrid(np.geomspace(0.1, 1.0, 8), np.geomspace(0.05, 0.5, 8), indexing="ij")
    return R.ravel(), S.ravel()


def c1_record(ctx, dev, skip_cpu):
    """BASELINE config c1 (N = 256, D = 1: x = linspace(0, 10, 256), y = sin(x), alpha = 1, rho = 1, sigma = 0.1 --
    the reference's plumbing size, R/tests.R / SURVEY KAT K2) and the sizes its drivers really call the path at
    (N = 21: R/tests.R:5; N = 199: pendulum_fit.R:26-27): latency of ONE host-buffer call (what `.Call` binds),
    per-evaluation time of a 64-point grid (one launch of the one-workgroup kernels), and the single-threaded
    oracle timed directly at the same N beside them."""
    import torch
    rec = {"workload": "c1 and the reference's own sizes: one host-buffer gpmi_logml call; 64-point (rho x sigma) grid through "
                       "gpmi_logml_grid_dev (one workgroup per point, one launch)", "sizes": []}
    ctx.set_stream(None)
    for n in (21, 199, 256, 512, 1024):   # 512, 1024: the grid runs one workgroup per point with the parameters in device memory
        x = np.linspace(0.0, 10.0, n).reshape(-1, 1); y = np.sin(x[:, 0])
        ctx.logml(x, y, 1.0, [1.0], 0.1)
        t0 = time.perf_counter()
        calls = 200 if n <= 256 else 40
        for _ in range(calls):
            val = ctx.logml(x, y, 1.0, [1.0], 0.1)[0]
        host_us = 1e6 * (time.perf_counter() - t0) / calls
        dx = torch.from_numpy(x[:, 0].copy()).to(dev); dy = torch.from_numpy(y).to(dev)
        G = 64
        out = torch.zeros((G, 3), dtype=torch.float64, device=dev); info = torch.zeros(G, dtype=torch.int32, device=dev)
        R, S = c4_grid()
        best = float("inf")
        for r in range(6):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ctx.logml_grid_dev(dx.data_ptr(), n, n, 1, dy.data_ptr(), np.ones(G), R * 10.0, S, 0.0, out.data_ptr(), info.data_ptr())
            ctx.sync()
            if r:
                best = min(best, time.perf_counter() - t0)
        grad_us = grad4_us = None
        if n <= 256:   # value + gradient: what one leapfrog step of the reference's Stan fit asks for (one chain / rstan's four)
            ctx.logml_grad(x, y, 1.0, [1.0], 0.1)
            t0 = time.perf_counter()
            for _ in range(100):
                ctx.logml_grad(x, y, 1.0, [1.0], 0.1)
            grad_us = 1e6 * (time.perf_counter() - t0) / 100
            a4 = np.ones(4); r4 = np.array([1.0, 1.1, 0.9, 1.2]); s4 = np.full(4, 0.1)
            ctx.logml_grad_grid(x, y, a4, r4, s4)
            t0 = time.perf_counter()
            for _ in range(50):
                ctx.logml_grad_grid(x, y, a4, r4, s4)
            grad4_us = 1e6 * (time.perf_counter() - t0) / 50
        e = {"N": n, "logml": float(val), "host_buffer_call_us": host_us, "value_and_gradient_call_us": grad_us,
             "value_and_gradient_four_chains_call_us": grad4_us, "grid64_us_per_eval": 1e6 * best / G,
             "grid64_evals_per_s": G / best, "grid64_results_ok": bool(np.all(info.cpu().numpy() == 0))}
        if not skip_cpu:
            from oracle import oracle as orc
            orc.logml(x, y, 1.0, 1.0, 0.1)
            t0 = time.perf_counter()
            reps = 200 if n <= 100 else (20 if n <= 256 else 2)
            for _ in range(reps):
                want = orc.logml(x, y, 1.0, 1.0, 0.1)[0]
            e["cpu_oracle_1_thread_us"] = 1e6 * (time.perf_counter() - t0) / reps
            e["rel_err_vs_oracle"] = abs(val - want) / abs(want)
        rec["sizes"].append(e)
    return rec


def dry_run(args):
    """The N>1 path without a GPU: same launcher, same rendezvous, same sharding (point g -> rank
    g mod P) and the same all_gather as the c4 sub-record, with a closed-form stand-in for the
    evaluation.  Used by tests/test_grid_gloo.py."""
    import torch.distributed as dist
    from gp_amd.grid import logml_grid_sharded
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rho, sig = c4_grid()

    def fake(X, y, alpha, rho, sigma, jitter):
        out = np.stack([np.sin(7.0 * rho) - sigma, np.log(rho), sigma * sigma], axis=1)
        return out, np.zeros(len(rho), dtype=np.int32)

    res, info = logml_grid_sharded(None, None, np.ones(C4_G), rho, sig, 0.0, evaluate=fake)
    want, _ = fake(None, None, None, rho, sig, 0.0)
    ok = bool(np.array_equal(res, want) and np.all(info == 0))
    n_seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": n_seen, "gpus_arg": args.gpus, "grid_points": C4_G,
                          "results_ok": ok, "launched_by_bench": bool(os.environ.get("GPMI_BENCH_LAUNCHED"))}))
        sys.stdout.flush()
    return 0 if ok else 1


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        sys.exit(dry_run(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    coll_group, coll_backend = None, None   # group / backend of the data-path collectives (None: default group)
    if args.rehearse:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d needs cuda:%d but only %d device(s) are visible "
                         "(use --rehearse to run every rank on cuda:0)\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # Rendezvous and agreement channel: gloo.  Data-path collectives: RCCL ("nccl") over xGMI, as its own group.  If
        # RCCL cannot come up on this node (every rank reports through the gloo channel; a stuck communicator creation is
        # aborted by its timeout) the run is not thrown away: the one gather is 32 B per point, so it goes over gloo on
        # host tensors and the line says so (`collective_backend`); evaluations and timing are unaffected.
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if not args.rehearse:
            import datetime
            ok_here, note = 1, ""
            try:
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe, group=g)  # creates the communicator now, not inside the timed region
                torch.cuda.synchronize(dev)
                ok_here = int(float(probe.item()) == float(world))
            except Exception as e:               # noqa: BLE001 -- any RCCL failure
                ok_here, note = 0, repr(e)[:160]
            flag = torch.tensor([ok_here], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                coll_group, coll_backend = g, "nccl"
            else:
                sys.stderr.write("bench.py rank %d: nccl unavailable on some rank (%s); the gather goes over gloo\n" % (rank, note))
                args.rehearse_collectives = True
                coll_backend = "gloo (nccl unavailable%s)" % ((": " + note) if note else " on another rank")
        else:
            coll_backend = "gloo (--rehearse)"
        world = dist.get_world_size()   # what the collective layer actually sees

    import gp_amd
    ctx = gp_amd.Context(local_rank)
    n, D = args.n, args.d
    if args.workload == "c4":
        n = C4_N
    elif args.workload == "c5":
        n, D = 8192, 1
    if args.nb_outer:
        ctx.set_option("nb_outer", args.nb_outer)
    if args.grid_lanes:
        ctx.set_option("grid_lanes", args.grid_lanes)
    if args.syrk_order >= 0:
        ctx.set_option("syrk_order", args.syrk_order)
    ctx.reserve(2 * n if args.workload == "c5" else n)  # workspaces of every grid lane, outside the timed region

    # deterministic synthetic inputs, generated on the host once and
    # moved to HBM before the timed region
    from gp_amd.synth import synth
    X, y = synth(n, D)
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)   # (D, n) row-major == n x D column-major
    dy = torch.from_numpy(y).to(dev)
    steps, warm = args.steps, args.warmup
    # Everything below -- the library's launches, the copy of the results and the gather -- is enqueued
    # on ONE explicit torch stream, so the collective is ordered behind the evaluations it sends
    stream = torch.cuda.Stream(dev)
    ctx.set_stream(stream.cuda_stream)
    cdev = torch.device("cpu") if (args.rehearse or getattr(args, "rehearse_collectives", False)) else dev  # gloo: collectives on host tensors

    if args.workload == "c3":
        # rank r, step k evaluates its own hyper-parameter point near (rho, sigma) = (0.3, 0.1)
        per_step = 1
        npts = steps + warm
        rho = 0.3 * (1.0 + 0.01 * ((np.arange(npts) * world + rank) % 16))
        sig = 0.1 * np.ones(npts)
    elif args.workload == "c4":
        # point g -> rank g mod world
        R, S = c4_grid()
        mine = np.arange(rank, C4_G, world)
        per_step = mine.size
        npts = per_step * (steps + warm)
        rho = np.tile(R[mine], steps + warm)
        sig = np.tile(S[mine], steps + warm)
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

    def run_points(lo, hi):
        """Evaluate points lo..hi-1 of this rank.  c3 / c4 go through the grid entry point, which
        overlaps independent points on internal lanes (own workspaces and streams)."""
        if args.workload == "c5":
            ctx.joint_logml_grid_dev(dX.data_ptr(), n, dy.data_ptr(), np.ones(hi - lo), rho[lo:hi], sig[lo:hi], 1e-6,
                                     dout[lo].data_ptr(), dinfo[lo:].data_ptr())
        else:
            ctx.logml_grid_dev(dX.data_ptr(), n, n, D, dy.data_ptr(), np.ones(hi - lo), rho[lo:hi], sig[lo:hi], 0.0,
                               dout[lo].data_ptr(), dinfo[lo:].data_ptr())

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    from gp_amd.grid import logml_grid_local_dev, logml_grid_sharded_dev
    full4 = None

    def c4_step():
        """One step of the c4 workload = the whole 64-point grid through the PRODUCT's sharded entry point
        (gp_amd.grid.logml_grid_sharded_dev: point g -> rank g mod P on the lanes, results packed on the
        device, ONE all_gather), on the current stream."""
        R, S = c4_grid()
        return logml_grid_sharded_dev(ctx, dX.data_ptr(), n, n, D, dy.data_ptr(), np.ones(C4_G), R, S, 0.0,
                                      group=coll_group, device=dev, comm_device=cdev)

    with torch.cuda.stream(stream):
        if args.workload == "c4":
            for _ in range(max(warm, 1) if distributed else warm):  # also creates the gather's buffers / channels
                c4_step()
        elif warm:
            run_points(0, warm * per_step)
            if distributed:  # the gather's buffers / channels exist before the timed region, like the workspaces
                send = dout.to(cdev)
                gathered = [torch.empty_like(send) for _ in range(world)]
                dist.all_gather(gathered, send, group=coll_group)
        barrier()
        t0 = time.perf_counter()
        if args.workload == "c4":
            for _ in range(steps):  # exactly `steps` steps, each the whole grid + its gather
                full4 = c4_step()
        else:
            run_points(warm * per_step, (warm + steps) * per_step)  # exactly `steps` steps
            if distributed:
                # the path's only collective: gather the per-point results (3 doubles per point); same
                # stream as the evaluations, so it reads them only after the lanes have joined
                send = dout.to(cdev)
                gathered = [torch.empty_like(send) for _ in range(world)]
                dist.all_gather(gathered, send, group=coll_group)
        barrier()
        elapsed = time.perf_counter() - t0

    # Roofline pass (same run, rank 0 only, after the timed region): the same evaluations one at
    # a time with HIP-event pairs around every covariance-build, trailing-update and panel-phase
    # launch group on the launch stream.  Done separately because concurrent lanes overlap launches,
    # which makes per-launch durations meaningless inside the throughput region.
    kt = {"syrk": (0, 0.0, 0.0), "build": (0, 0.0, 0.0), "panel": (0, 0.0, 0.0), "syrk_multi_round": (0, 0.0, 0.0)}
    seq_ms = None
    if rank == 0:
        ctx.set_option("grid_lanes", 1)
        ctx.set_option("kernel_timing", 1)
        ctx.kernel_timing(reset=True)
        nprof = min(steps * per_step, 4)
        with torch.cuda.stream(stream):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            lo = warm * per_step
            for p in range(lo, lo + nprof):
                run_points(p, p + 1)
            torch.cuda.synchronize(dev)
            seq_inst_ms = 1e3 * (time.perf_counter() - t1) / nprof
        kt = ctx.kernel_timing(reset=True)
        ctx.set_option("kernel_timing", 0)
        # ... and the library's default one-at-a-time path (what a NUTS / optimiser loop sees: one
        # evaluation per step), without instrumentation
        with torch.cuda.stream(stream):
            run_points(lo, lo + 1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for p in range(lo, lo + nprof):
                run_points(p, p + 1)
            torch.cuda.synchronize(dev)
            seq_ms = 1e3 * (time.perf_counter() - t1) / nprof
        ctx.set_option("grid_lanes", args.grid_lanes)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=coll_group)
    elapsed = float(tmax.item())
    if args.workload == "c4":
        f4 = full4.cpu().numpy()
        res, info = f4[:, :3], f4[:, 3].astype(np.int32)
    else:
        res = dout.cpu().numpy()[warm * per_step:]
        info = dinfo.cpu().numpy()[warm * per_step:]
    ok = bool(np.all(info == 0) and np.all(np.isfinite(res[:, 0])))

    # ---- c4 sub-record: the 64-point grid at N=8192 over the same ranks (strong scaling) --------------
    c4 = None
    if args.workload == "c3" and not args.no_c4:
        X4, y4 = synth(C4_N, C4_D)
        dX4 = torch.from_numpy(np.ascontiguousarray(X4.T)).to(dev)
        dy4 = torch.from_numpy(y4).to(dev)
        R4, S4 = c4_grid()

        def sharded():   # the PRODUCT's multi-GPU function: gp_amd.grid.logml_grid_sharded_dev
            return logml_grid_sharded_dev(ctx, dX4.data_ptr(), C4_N, C4_N, C4_D, dy4.data_ptr(), np.ones(C4_G), R4, S4, 0.0,
                                          group=coll_group, device=dev, comm_device=cdev)

        def one_rank():  # all 64 points on this rank's GPU alone, same building block, no collective
            return logml_grid_local_dev(ctx, dX4.data_ptr(), C4_N, C4_N, C4_D, dy4.data_ptr(), np.ones(C4_G), R4, S4, 0.0, dev)

        with torch.cuda.stream(stream):
            sharded()  # warm-up (evaluation and the gather's buffers / channels)
            barrier()
            t0 = time.perf_counter()
            got4 = sharded()
            barrier()
            tN = time.perf_counter() - t0
            # the same run's ONE-rank figure: rank 0 evaluates all 64 points, the others wait
            t1r = tN
            ref4 = got4
            if distributed:
                if rank == 0:
                    one_rank()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    ref4 = one_rank()
                    torch.cuda.synchronize(dev)
                    t1r = time.perf_counter() - t0
                barrier()
        tt = torch.tensor([tN], dtype=torch.float64, device=cdev)
        if distributed:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=coll_group)
        tN = float(tt.item())
        if rank == 0:
            got = got4.cpu().numpy()[:, :3]
            same = bool(np.array_equal(got, ref4.cpu().numpy()[:, :3]))
            c4 = {"workload": "c4: 64-point (rho x sigma) grid at N=%d, D=%d, point g -> rank g mod %d, one all_gather"
                              % (C4_N, C4_D, world),
                  "grid_points": C4_G, "n_gpus": world, "wall_s": tN, "evals_per_s": C4_G / tN,
                  "wall_s_one_rank_same_run": t1r, "speedup_vs_one_rank": t1r / tN,
                  "sharded_results_bit_identical_to_one_rank": same,
                  "entry_point": "gp_amd.grid.logml_grid_sharded_dev",
                  "results_ok": bool(np.all(np.isfinite(got[:, 0]))),
                  "argmax_point": int(np.nanargmax(got[:, 0])), "logml_max": float(np.nanmax(got[:, 0]))}

    if rank == 0:
        evals = steps * (C4_G if args.workload == "c4" else world * per_step)
        value = evals / elapsed
        order = 2 * n if args.workload == "c5" else n
        syrk_n, syrk_ms, syrk_flops = kt["syrk"]
        build_n, build_ms, build_bytes = kt["build"]
        pan_n, pan_ms, pan_flops = kt["panel"]
        mr_n, mr_ms, mr_flops = kt["syrk_multi_round"]
        ach = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        pach = pan_flops / (pan_ms * 1e-3) / 1e12 if pan_ms > 0 else 0.0
        chol_flops = order ** 3 / 3.0
        nbo_auto = (lambda nf: 1024 if nf >= 8192 else 512 if nf >= 4608 else 256 if nf >= 3584 else 128)(order)
        build_kernel = "k_joint_cov" if args.workload == "c5" else "k_se_cov<%d>" % D
        syrk_traffic, syrk_util, syrk_src = pmc_lookup(args.workload, order, "k_gemm_nt<1>")
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
                      "+ solve + log-det, 1 (l, sigma) point per step per GPU" % (n, 2 * n)}[args.workload],
                       "N": n, "D": D, "alpha": 1.0,
                       "rho": {"c3": "0.300 ... 0.345: step k of rank r evaluates rho = 0.3 (1 + 0.01 ((k P + r) mod 16)) "
                                     "(distinct points, as a grid or a sampler would ask for)",
                               "c4": "8 log-spaced values in [0.1, 1.0] (x 8 sigma values)",
                               "c5": "l = 0.500 ... 0.535: step k evaluates l = 0.5 (1 + 0.01 (k mod 8))"}[args.workload],
                       "sigma": "8 log-spaced values in [0.05, 0.5]" if args.workload == "c4" else 0.1,
                       "nb_outer": args.nb_outer or "auto(adaptive: %d for the first block, narrower as the trailing matrix shrinks)" % nbo_auto,
                       "parallelism": "independent hyper-parameter points per GPU; one RCCL all_gather of results"},
            "results_ok": ok,
            "collective_backend": coll_backend,
            "launched_by": "bench.py launcher" if os.environ.get("GPMI_BENCH_LAUNCHED") else
                           ("torchrun / external" if distributed else "single process"),
            "grid_lanes": args.grid_lanes or "auto(4)",
            "ms_per_eval_sequential": seq_ms,
            "ms_per_eval_sequential_instrumented": seq_inst_ms,
            "logml_first": float(res[0, 0]),
            "cholesky_tflops_per_gpu_whole_eval": chol_flops * evals / world / elapsed / 1e12,
            "roofline": {
                "kernel": "k_gemm_nt<1> (trailing-update SYRK, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "measured_in": "instrumented sequential pass of the same evaluations in this run "
                               "(lanes=1: bracketed launches run alone on the chip)",
                "achieved": ach,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / FP64_PEAK_TFLOPS,
                "traffic": syrk_traffic,
                "traffic_unit": "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes)",
                "traffic_source": syrk_src,
                "mfma_util_pmc": syrk_util,
                "launches": int(syrk_n),
                "avg_launch_ms": syrk_ms / max(syrk_n, 1),
                "flops_per_launch_avg": syrk_flops / max(syrk_n, 1),
                # subset of the above: launches of more than one round of tiles (throughput-bound; a single-round
                # launch also factors the next diagonal block and is bounded by that ~27 us latency chain)
                "multi_round_launches": {
                    "launches": int(mr_n), "avg_launch_ms": mr_ms / max(mr_n, 1),
                    "achieved": mr_flops / (mr_ms * 1e-3) / 1e12 if mr_ms > 0 else 0.0,
                    "frac": (mr_flops / (mr_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if mr_ms > 0 else 0.0,
                    "share_of_syrk_flops": mr_flops / syrk_flops if syrk_flops > 0 else 0.0},
            },
            "roofline_panel": {
                "kernel": "panel phase of one outer block: k_potrf_diag4 + k_trsm_panel + k_gemm_nt<0> (in-block "
                          "products with the fused diagonal blocks); latency chain, 128 sequential pivot blocks",
                "bound": "mfma",
                "achieved": pach,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": pach / FP64_PEAK_TFLOPS,
                "outer_blocks": int(pan_n),
                "ms_per_eval": pan_ms / max(nprof, 1),
                "mfma_util_pmc_gemm_nt0": pmc_lookup(args.workload, order, "k_gemm_nt<0>")[1],
                "mfma_util_pmc_trsm_panel": pmc_lookup(args.workload, order, "k_trsm_panel")[1],
                "traffic_gemm_nt0": pmc_lookup(args.workload, order, "k_gemm_nt<0>")[0],
            },
            "roofline_build": {
                "kernel": "k_joint_cov (lower-triangular joint [y, y'] covariance build)" if args.workload == "c5"
                else "%s (lower-triangular SE covariance build)" % build_kernel,
                "bound": "hbm",
                "achieved": build_bytes / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (build_bytes / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if build_ms > 0 else 0.0,
                "traffic": pmc_lookup(args.workload, order, build_kernel)[0],
                "algorithmic_bytes_per_launch": build_bytes / max(build_n, 1),
                "avg_launch_ms": build_ms / max(build_n, 1),
            },
        }
        if c4 is not None:
            line["c4"] = c4
        if world == 1 and args.workload == "c3":
            # the host-buffer entry point `.Call` binds (gpmi_logml): X, y over PCIe, 3 doubles back,
            # blocking -- the PCIe-inclusive time of one evaluation, never `value`
            ctx.set_stream(None)
            ctx.logml(X, y, 1.0, [0.3], 0.1)
            t1 = time.perf_counter()
            ctx.logml(X, y, 1.0, [0.3], 0.1)
            line["ms_per_eval_host_buffer_abi"] = 1e3 * (time.perf_counter() - t1)
        if world == 1 and args.workload == "c3" and not args.no_c1:
            line["c1"] = c1_record(ctx, dev, args.no_cpu_baseline)
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
            # best-effort legs: LAPACK on the CPU share of a one-GPU box (16 threads) and on ALL host cores
            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            line["cpu_baseline"]["host_cores_usable"] = usable
            for key, thr in (("lapack_16_threads", min(16, usable)), ("lapack_all_cores", usable)):
                try:
                    cpu_lapack(1024, D, thr)  # thread-pool warm-up
                    dt2, lm2 = cpu_lapack(ns, D, thr)
                    line["cpu_baseline"][key] = {
                        "value": 1.0 / (dt2 * scale), "unit": "evals/s", "cores": thr,
                        "sample": "numpy + LAPACK dpotrf/dtrtrs (scipy, OpenBLAS) at N=%d (%.2f s), scaled by "
                                  "(N/Ns)^3; upper bound for a CPU build, not the reference's path" % (ns, dt2),
                        "sample_rel_err_vs_oracle": abs(lm2 - lm_cpu) / abs(lm_cpu)}
                except Exception as e:  # optional leg: never fails the bench line
                    line["cpu_baseline"][key] = {"error": repr(e)}
        print(json.dumps(line))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
