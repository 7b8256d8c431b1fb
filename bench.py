#!/usr/bin/env python3
"""Headline benchmark: GP posteriors/s and effective Cholesky GFLOP/s at N = 8192.

One "step" = one complete dense GP posterior of one (resolution, partition)
block with inputs resident in HBM: RBF Gram build (D1) -> blocked Cholesky (D2)
-> alpha = K^-1 r (D3) -> predictive mean and variance at N/4 test points
(D4/D5).  With --gpus N every rank runs the same step on its own independent
partition (weak scaling) and the per-step predictions are summed with ONE
all-reduce of the fused [mean | var] buffer (RCCL over xGMI).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...          # WORLD_SIZE unset: spawns its own N ranks (see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config 4 --gpus N [--n 8192]   # BASELINE configs[3]: the sharded 2-D hierarchy (run_config)
    python bench.py --nccl-world1                    # the N = 1 step with backend nccl (RCCL) initialised and the
                                                     # fused buffer reduced unconditionally (one-GPU RCCL evidence)
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (spec)
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA peak


import workloads

EVENT_EVERY = 4          # timed steps whose trailing-update launches are bracketed by events (main)
PMC_PROFILE = "r05_pmc_bench_n8192.json"


def _sources_sha():
    """sha256 over the kernel sources the committed PMC profile describes (tools/make_pmc_profile.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("common.hpp", "gemm_tile.hpp", "gemm_nt.hip", "potrf.hip", "gram.hip"):
        with open(os.path.join(ROOT, "cimrgp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def make_block(n, q, seed):
    """SURVEY.md 8d config 2 (workloads.make_block): x ~ sorted U(-sqrt3, sqrt3),
    y_k = sin(3x+k) + 0.5 sin(17x^2) + 0.1 N(0,1)."""
    return workloads.make_block(n, q, seed)


def _host_description():
    """CPU model, thread count and BLAS build of the box the baseline is timed on."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas, threads = "unknown", os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            threads = max(p.get("num_threads", 1) for p in pools)
            blas = "; ".join("%s %s (%s, %s)" % (p.get("internal_api"), p.get("version"), p.get("threading_layer", "?"),
                                                 p.get("architecture", "?")) for p in pools)
    except Exception:
        pass
    return model, int(threads), blas


def cpu_baseline(n, ns, q, ell, sf2, noise, seed, warmups=3, repeats=5):
    """The oracle (NumPy/SciPy port) timed on the host cores on complete posteriors of the same
    workload: ``warmups`` untimed runs, then the median of ``repeats``.  A reported
    baseline, never the product path.  Also returns the oracle's outputs: bench.py checks the GPU
    step against them at full size."""
    import oracle
    x, y = make_block(n, q, seed)
    xs = workloads.block_test_points(ns)
    times = []
    for it in range(warmups + repeats):
        t0 = time.perf_counter()
        fit = oracle.block_fit(x, y, ell, sf2, noise)
        omean, ovar = oracle.block_predict(x, fit, xs, ell, sf2, True)
        dt = time.perf_counter() - t0
        if it >= warmups:
            times.append(dt)
    med = float(np.median(times))
    model, threads, blas = _host_description()
    rec = dict(value=1.0 / med, unit="posteriors/s", cores=threads, kind="port",
               sample="%d warm-ups + median of %d complete posteriors (Gram + Cholesky + solves + mean/var at N/4 "
                      "points) of the N=%d block, NumPy/SciPy FP64: %.2f s each (min %.2f, max %.2f)"
                      % (warmups, repeats, n, med, min(times), max(times)),
               cpu_model=model, blas=blas, seconds_per_posterior=med)
    return rec, omean, ovar


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script as child
    processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one GPU each) BEFORE
    this process has made any GPU call (it never makes one), let them print, and exit with their
    status.  Children are started directly rather than through ``torch.distributed.run``: its
    argument parser tries to abbreviate-match this script's own flags (``--n``)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # RCCL shares device buffers between the ranks of a node through IPC handles; the host driver of this
        # pool only supports dmabuf IPC, and without this switch hipIpcGetMemHandle fails with "invalid argument"
        # at communicator creation (the image exports it already: kept for environments built by hand)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if failed:                       # a rank that failed must not leave the others waiting in a collective
                rc = abs(failed[0])
                break
        rc = max([rc] + [abs(p.returncode) for p in procs if p.poll() is not None])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def run_config(args, world, rank, rehearsal):
    """BASELINE configs[2] / configs[3]: the multiresolution hierarchy end to end through the host API, its
    (resolution, partition) blocks sharded over the ranks (dist.assign_blocks: longest processing time
    first), one all-reduce per layer of the layer's training-point prediction (Stats.py:126-157) and ONE of
    the fused [mean | var] buffer at prediction (MRGP.py:802-803).  Total work is fixed as ranks are added
    (strong scaling).  Rank 0 prints ONE JSON line; at n <= 16384 (or --check) the result is compared with
    the CPU oracle on the same arrays."""
    import torch
    import torch.distributed as td
    import cimrgp_amd as ca
    q = 2
    if args.config == 3:
        n = args.n or 65536
        res, d, power = 4, 1, 0
        x, y, xs = workloads.make_chain_1d(n, q)
        ells = workloads.chain_length_scales(res + 1, 1)
        name = "BASELINE configs[2]: 1-D, N=%d, IndexSetUniform(N, 4, 2): 5 resolutions, 31 partitions" % n
        policy = "single root region (the reference's index set)"
    else:
        # root-block policy: the hierarchy starts at a layer whose blocks fit one device
        # (first_divider_power=3 -> 8, 16, 32, 64, 128 regions); inputs in Hilbert order
        n = args.n or 262144
        res, d, power = 4, 2, 3
        x, y, xs = workloads.make_chain_2d(n, q, order=ca.space_filling_order)
        ells = workloads.chain_length_scales(res + 1, 2, ell0=0.7)
        name = "BASELINE configs[3]: 2-D, N=%d, 5 resolutions of 8/16/32/64/128 partitions" % n
        policy = "first_divider_power=3 (no single-region root: a 262144-point root would be a 512 GiB matrix); Hilbert-ordered inputs"
    ns = xs.shape[0]
    kernels = [ca.RBFKernel(l=l, sf=1.0, noise=0.01) for l in ells]
    idx = ca.IndexSetUniform(n, res, 2, first_divider_power=power)
    idx_t = ca.IndexSetUniform(ns, res, 2, first_divider_power=power)
    group = td.group.WORLD if td.is_initialized() else None

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    fits, preds, layer_ms = [], [], None
    mean = var = None
    owner = None
    for it in range(args.warmup + args.steps):
        barrier()
        t0 = time.perf_counter()
        model = ca.MultiResolutionGaussianProcess([x, y], index_set_obj=idx, spectral_density_obj=kernels, dtype=args.dtype,
                                                  process_group=group, keep_factors=True)
        model.fit()
        barrier()
        t1 = time.perf_counter()
        mean, var = model.get_predicted_mean_and_var(xs, idx_t)
        barrier()
        t2 = time.perf_counter()
        if it >= args.warmup:
            fits.append(t1 - t0)
            preds.append(t2 - t1)
            layer_ms = model.layer_fit_ms()
        owner = [np.asarray(o).tolist() for o in model.owner]
        first_local = int(model._first_local)
        n_regions, n_samps = model.n_regions, model.n_samps
        del model
    fit_s, pred_s = float(np.median(fits)), float(np.median(preds))
    if world > 1:                                   # the slowest rank's clock is the job's
        t = torch.tensor([fit_s, pred_s], dtype=torch.float64, device="cuda")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        fit_s, pred_s = float(t[0].item()), float(t[1].item())
    nblocks = int(sum(n_regions))
    if rank == 0:
        out = {
            "metric": "GP posteriors/sec, %s" % name,
            "value": nblocks / (fit_s + pred_s),
            "unit": "posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": (fit_s + pred_s) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": name + ", q=2 outputs, N/4=%d test points, noise 0.01" % ns, "root_policy": policy,
                       "regions_per_layer": n_regions, "block_sizes_per_layer": [sorted(set(v)) for v in n_samps],
                       "blocks_per_rank": [[int(sum(1 for o in layer if o == r)) for layer in owner] for r in range(world)],
                       "parallelism": "nested block ownership (cimrgp_amd.dist.plan_layers): layers below first_local_layer by LPT with "
                                      "1 all-reduce each, the others exchange nothing during the fit; 1 all-reduce after the sweep + 1 at prediction",
                       "first_local_layer": first_local,
                       "backend": (rehearsal or "nccl") if td.is_initialized() else "none"},
            "fit_s": fit_s, "predict_s": pred_s, "layer_fit_ms": layer_ms,
            "cholesky_flops": float(sum(sum(float(m) ** 3 / 3 for m in layer) for layer in n_samps)),
            "mean_finite": bool(np.isfinite(mean).all()), "var_finite": bool(np.isfinite(var).all()),
            "var_min": float(var.min()), "var_max": float(var.max()),
            "peak_mem_GiB_rank0": torch.cuda.max_memory_allocated() / 2 ** 30,
        }
        out["cholesky_tflops_fit"] = out["cholesky_flops"] / fit_s / 1e12
        if args.check or n <= 16384:
            import oracle
            xn, _, mu, sd = oracle.normalize_inputs(x)
            specs = [oracle.DenseLayerSpec(l, 1.0, 0.01) for l in ells]
            om, _ = oracle.mrgp_fit(xn, y, oracle.index_bounds_uniform(n, res, 2, power), specs)
            omean, ovar = oracle.mrgp_predict(xn, om, specs, (xs - mu) / sd, oracle.index_bounds_uniform(ns, res, 2, power))
            out.update(parity_fields(mean, var, omean, ovar))
        print(json.dumps(out))
        if out.get("parity_ok") is False:
            raise SystemExit("bench.py --config %d: the GPU result differs from the oracle by more than 1e-5" % args.config)
    if td.is_initialized():
        td.destroy_process_group()
    return 0


def parity_fields(mean, var, omean, ovar):
    """north_star: "1e-5 relative on predictive mean/variance".  Two readings are reported: relative to the
    ARRAY's largest magnitude (max |err| / max |ref|, the bar of tests/) and ELEMENT-WISE for the variance
    (max |err_i| / |ref_i|: the strict reading -- variances span orders of magnitude)."""
    rec = {
        "parity_rel_err_mean": float(np.max(np.abs(mean - omean)) / np.max(np.abs(omean))),
        "parity_rel_err_var": float(np.max(np.abs(var - ovar)) / np.max(np.abs(ovar))),
        "parity_rel_err_var_elementwise": float(np.max(np.abs(var - ovar) / np.abs(ovar))),
        "oracle_var_min": float(np.min(ovar)),
    }
    rec["parity_ok"] = bool(rec["parity_rel_err_mean"] <= 1e-5 and rec["parity_rel_err_var"] <= 1e-5
                            and rec["parity_rel_err_var_elementwise"] <= 1e-5)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="do not bracket the trailing-update launches of the timed steps with events (measures what the "
                         "roofline's live timing costs; the roofline then comes from the factorisations without carried rows only)")
    ap.add_argument("--cpu-warmups", type=int, default=3)
    ap.add_argument("--cpu-repeats", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4],
                    help="2: the headline N = 8192 block per GPU (default); 3 / 4: BASELINE configs[2] / [3], the "
                         "multiresolution hierarchy with its blocks sharded over the ranks (--n scales it down)")
    ap.add_argument("--check", action="store_true", help="--config 3/4: compare with the CPU oracle (done anyway for n <= 16384)")
    ap.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: consecutive (independent) blocks pipelined over three streams and three buffer sets "
                         "(cimrgp_block_posterior_staged); 0: one stream, one set; default: 1 unless ranks share one GPU")
    ap.add_argument("--front-queue", type=int, default=int(os.environ.get("CIMRGP_BENCH_FRONT", "1")), choices=[0, 1],
                    help="(pipelined steps) 1: the front end and the start of step i+1's factorisation on the context's chain queue "
                         "(cimrgp_front_queue), beside the last panels of step i; 0: behind step i on its stream (rounds 1-4)")
    ap.add_argument("--nccl-world1", action="store_true",
                    help="one rank, an RCCL communicator of one created, the step's reduce goes through RCCL unconditionally")
    ap.add_argument("--comm", default="cabi", choices=["cabi", "torch"],
                    help="the step's collective: cabi = cimrgp_allreduce_sum (the C ABI's RCCL communicator) enqueued on the step's "
                         "solve queue -- no stream of its own; torch = torch.distributed's all_reduce (backend nccl: ProcessGroupNCCL "
                         "brings a stream of its own, a fifth queue for a runtime that serves four: round-4 record)")
    ap.add_argument("--rows-queues", type=int, default=0, choices=[0, 1, 2],
                    help="queues for the carried rows (cimrgp_set_rows_queues); 0 = the default policy")
    args = ap.parse_args()
    if args.config in (3, 4) and "--n" not in " ".join(sys.argv):
        args.n = None

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as td
    import cimrgp_amd as ca
    from cimrgp_amd import device as dev, dist, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CIMRGP_BENCH_REHEARSAL=gloo: rehearse the N > 1 path on a box with ONE GPU (all ranks on
    # cuda:0, gloo carrying the device tensors); the driver's multi-GPU runs use nccl = RCCL
    rehearsal = os.environ.get("CIMRGP_BENCH_REHEARSAL", "")
    # The DATA path's collective is RCCL in every non-rehearsal run.  --comm cabi (default): the C ABI's communicator
    # (cimrgp_comm_* / cimrgp_allreduce_sum, RCCL loaded by the library) enqueued on the step's own solve queue; the
    # process group is then only the control plane (the id's broadcast, the barriers around the timed region, the MAX
    # over ranks of the time) and runs on gloo, so that no RCCL stream of torch's exists in the process.
    # --comm torch: backend nccl, the step's all_reduce through torch.distributed (the round-4 form).
    cabi = (args.comm == "cabi") and not rehearsal and args.config == 2
    comm = None
    fallback_group = None                  # torch.distributed nccl group for the data path when the C-ABI communicator failed
    if args.gpus > 1 or world > 1 or args.nccl_world1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0 if rehearsal else local_rank)
        if rehearsal:
            td.init_process_group(rehearsal, rank=rank, world_size=world)
        elif cabi:
            td.init_process_group("gloo", rank=rank, world_size=world)
            ident = [_lib.Comm.unique_id() if rank == 0 else None]
            td.broadcast_object_list(ident, src=0)
            # collective; binds the current device (set above).  Should the communicator fail on ANY rank (it has only
            # ever been created for a world of one: no multi-GPU node has been available), every rank falls back to
            # torch.distributed's nccl for the data path -- slower by the fifth queue, but a measured line
            try:
                if os.environ.get("CIMRGP_BENCH_FAIL_COMM"):      # test hook (tests/test_gpu_configs.py): exercise the fallback
                    raise RuntimeError("simulated failure")
                comm = _lib.Comm(world, rank, ident[0])
                ok = 1
            except Exception as exc:                              # noqa: BLE001 -- whatever it is, the ranks decide together
                sys.stderr.write("bench.py: rank %d: C-ABI communicator failed (%s): falling back to torch.distributed nccl\n" % (rank, exc))
                comm, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32)
            td.all_reduce(flag, op=td.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                    comm = None
                fallback_group = td.new_group(backend="nccl")
        else:
            # device_id: the communicator is bound to this rank's GPU up front (no guessing in barrier())
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    # Queues for the carried rows.  Ranks SHARING one GPU (the rehearsal) must each stay within the runtime's four
    # hardware queues -- caller, panel chain, carried rows, collective -- or the processes stall for hundreds of
    # milliseconds between steps (1581 against 22 ms per step with a fifth stream); with one
    # process per GPU the default (two rows queues) stays: --nccl-world1 measures that case with RCCL live.
    if args.rows_queues:
        _lib.set_rows_queues(args.rows_queues)
    elif rehearsal and world > 1:
        dist.share_one_gpu()
    if args.config in (3, 4):
        return run_config(args, world, rank, rehearsal)
    device = dev.require_gpu()
    tdt = dev.as_torch_dtype(args.dtype)
    lib = _lib.load()

    n, q = args.n, 2
    ns = n // 4
    ell, sf2, noise = 0.1, 1.0, 0.01
    x, y = make_block(n, q, 1234 + rank)
    xd = dev.to_device(x, tdt, device)
    yd = dev.to_device(y, tdt, device)
    xsd = dev.to_device(workloads.block_test_points(ns), tdt, device)

    # Buffers allocated once.  The timed step is ONE call of the boundary's fused entry point,
    # cimrgp_block_posterior[_staged] (include/cimrgp.h: Gram, cross-Gram and target rows, factorisation with the rows
    # carried, backward solve, mean and variance), on these buffers: no torch kernel and no allocation inside it.
    # Its outputs land directly in this rank's slices of the fused [mean (ns_total x q) | var (ns_total)] buffer of
    # the step's one collective.
    # Consecutive steps are INDEPENDENT blocks (the partitions of a layer), so they are pipelined: two buffer sets in
    # rotation, and the call's last stage -- the latency-bound backward solve and prediction (0.26 ms of 32 small
    # launches) -- on the look-ahead context's queue that is idle between two factorisations (cimrgp_solve_queue), beside
    # the Gram matrices and first panels of the next step.  (A stream of the bench's own for that stage is a fifth
    # queue for a runtime that serves four: 8.1 -> 10.2 ms per step; tools/lab/pipeline_probe.py.)
    # Every step's work is complete inside the timed region (device-wide synchronisation on both sides).
    pipeline = (args.pipeline == 1) or (args.pipeline == -1 and not (rehearsal and world > 1))
    nsets = 2 if pipeline else 1
    nst = ns * world

    def make_set():
        b = dict(kbuf=dev.alloc_matrix(n, n, tdt, device),
                 wbuf=dev.alloc_matrix(ns + q, n, tdt, device),      # carried rows: [K(X*, X); r^T]
                 ws=dev.potrf_workspace(n, tdt, device),
                 info=torch.zeros(1, dtype=torch.int32, device=device),
                 fused=torch.zeros(nst * (q + 1), dtype=tdt, device=device),
                 alpha=torch.empty((n, q), dtype=tdt, device=device),
                 zbuf=torch.empty((n, q), dtype=tdt, device=device),
                 scratch=torch.empty(2 * q * n, dtype=tdt, device=device))
        f = b["fused"]
        b["mean"] = f[:nst * q].view(nst, q)[rank * ns:(rank + 1) * ns]          # (ns x q), contiguous
        b["var"] = f[nst * q:][rank * ns:(rank + 1) * ns]                        # (ns,)
        b["others"] = [f[:nst * q].view(nst, q)[:rank * ns], f[:nst * q].view(nst, q)[(rank + 1) * ns:],
                       f[nst * q:][:rank * ns], f[nst * q:][(rank + 1) * ns:]] if world > 1 else []
        return b

    sets = [make_set() for _ in range(nsets)]
    kbuf, wbuf, ws, info, mean, var = (sets[0][k] for k in ("kbuf", "wbuf", "ws", "info", "mean", "var"))
    streams = None                                                                    # front, factor, solve
    if pipeline:
        cur = torch.cuda.current_stream()
        # front end of step i+1 on the context's chain queue, which falls idle in the last third of step i's factorisation
        # (cimrgp_front_queue; one box, three alternating pairs: 136.8 / 136.9 / 136.6 -> 137.4 / 137.2 / 137.3 posteriors/s)
        streams = (dev.front_queue(cur) if args.front_queue else cur, cur, dev.solve_queue(cur))
    pending = [None] * nsets                 # the set's collective in flight
    done = [None] * nsets                    # (pipeline) the set's last reader on the solve stream
    step_no = [0]
    last_set = [0]

    def reduce_begin(fused, stream):
        """Start the step's one collective behind the work already enqueued on ``stream``.  C ABI: enqueued ON that stream
        (ordered by it: no handle); torch.distributed: its own stream, a handle whose wait() makes the current stream wait."""
        if comm is not None:
            comm.allreduce_sum((_lib.F64 if args.dtype == "f64" else _lib.F32), fused.data_ptr(), fused.numel(), stream.cuda_stream)
            return None
        return dist.allreduce_sum_begin(fused, group=fallback_group, force=args.nccl_world1)

    def step(wait_inside=False):
        si = step_no[0] % nsets
        step_no[0] += 1
        last_set[0] = si
        b = sets[si]
        if not pipeline:
            # the previous step's collective owns `fused` until it is done: the STREAM waits for it here (the host does
            # not block), a whole step after it was started
            if pending[si] is not None:
                pending[si].wait()
                pending[si] = None
            for t in b["others"]:                                                   # the other ranks' slices (world > 1 only)
                t.zero_()
            dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["zbuf"],
                                b["mean"], b["var"], scratch=b["scratch"])
            pending[si] = reduce_begin(b["fused"], torch.cuda.current_stream())           # the one collective
            if wait_inside and pending[si] is not None:
                pending[si].wait()
                pending[si] = None
            return
        # a step that waits for its collective inside overlaps nothing with the next one: its front end stays on the
        # factorisation's stream (the front queue is for pipelining: on it the start of a factorisation runs one-queue style,
        # which only pays beside the previous factorisation's tail)
        s_front, s_factor, s_solve = (streams[1], streams[1], streams[2]) if wait_inside else streams
        # the set comes back after nsets steps: its collective (started nsets steps ago) and its solve are waited for by
        # the front stream, not by the host
        with torch.cuda.stream(s_front):
            if pending[si] is not None:
                pending[si].wait()
                pending[si] = None
            if done[si] is not None:
                s_front.wait_event(done[si])
            for t in b["others"]:
                t.zero_()
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["zbuf"],
                            b["mean"], b["var"], scratch=b["scratch"], streams=(s_front, s_factor, s_solve))
        with torch.cuda.stream(s_solve):
            pending[si] = reduce_begin(b["fused"], s_solve)                               # the one collective
            if wait_inside and pending[si] is not None:
                pending[si].wait()
                pending[si] = None
            done[si] = torch.cuda.Event()
            done[si].record(s_solve)
        if wait_inside and comm is not None:
            s_front.wait_event(done[si])         # C-ABI collective: stream-ordered, so "waiting inside" = nothing of the next step beside it

    # the same work as five separate calls (the boundary's fine-grained entry points), with an event between the
    # stages: run AFTER the timed region, for `stage_ms` only (tests/ hold the two forms to bit-equality)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]

    def step_in_stages():
        ev[0].record()
        dev.rbf_gram(xd, ell, sf2, noise, lower_only=True, out=kbuf)               # D1
        ev[1].record()
        dev.rbf_cross(xsd, xd, ell, sf2, out=wbuf)                                  # rows of D5 ...
        wbuf[ns:ns + q, :n] = yd.t()                                                # ... and of D3's forward half
        ev[2].record()
        dev.potrf_rows(kbuf, n, wbuf, ns + q, ws, info)                             # D2 with the rows carried
        ev[3].record()
        z = wbuf[ns:ns + q, :n].t().contiguous()
        dev.solve_lt(kbuf, n, ws, z.clone())                                        # D3 backward half
        dev.predict_from_w(wbuf, ns, n, z, sf2, 0.0, None, mean, var, accumulate=False)   # D4 + D5 tail
        ev[4].record()

    def drain():
        for si in range(nsets):
            if pending[si] is not None:
                pending[si].wait()
                pending[si] = None

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    assert all(int(b["info"].item()) == 0 for b in sets), "Cholesky failed in warm-up"

    # The roofline's live timing: HIP events around the trailing-update launches, on their own stream, in every
    # EVENT_EVERY-th timed step (two event records per launch on the update queue cost ~1 % of a step when every
    # step carries them: 120.6 against 121.3-122.3 posteriors/s without any).
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_launch_events and i % EVENT_EVERY == 0:
            _lib.check(lib.cimrgp_profile_begin(), "cimrgp_profile_begin")
        step()
        if i % EVENT_EVERY == 0:
            _lib.check(lib.cimrgp_profile_pause(), "cimrgp_profile_pause")
        # stage times are read after the loop from the last step's events only
    drain()                                       # the last step's collective completes inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    tr_ms, tr_fl, tr_by, tr_cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.cimrgp_profile_collect_bytes(ctypes.byref(tr_ms), ctypes.byref(tr_fl), ctypes.byref(tr_by), ctypes.byref(tr_cnt)),
               "cimrgp_profile_collect_bytes")
    last_mean = sets[last_set[0]]["mean"].double().cpu().numpy()     # the last timed step's outputs (this rank)
    last_var = sets[last_set[0]]["var"].double().cpu().numpy()
    # the collective of the timed steps is started at the end of a step and waited for (by the stream) at the start
    # of the next: its latency is hidden.  What it costs when it is NOT hidden: steps that wait for it inside.
    drained_ms = None
    if world > 1 or args.nccl_world1:
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            step(wait_inside=True)
        barrier()
        drained_ms = (time.perf_counter() - t1) / 5 * 1e3
    # the reduced buffer holds this rank's slice unchanged (the other ranks contribute zeros there): exact
    step()
    lb = sets[last_set[0]]
    with torch.cuda.stream(streams[2] if pipeline else torch.cuda.current_stream()):
        mine = torch.cat([lb["mean"].reshape(-1), lb["var"]]).clone()
    drain()
    torch.cuda.synchronize()
    reduce_diff = float((torch.cat([lb["mean"].reshape(-1), lb["var"]]) - mine).abs().max().item())
    # stage times: the five-call form, outside the timed region (median of 3 after a warm-up)
    stage_runs = []
    for rep in range(4):
        step_in_stages()
        torch.cuda.synchronize()
        if rep:
            stage_runs.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
    stage_ms = np.median(np.asarray(stage_runs), axis=0)
    # Cholesky alone (no carried rows) for the effective-GFLOP/s figure, timed separately
    torch.cuda.synchronize()
    chol_ms = []
    _lib.check(lib.cimrgp_profile_begin(), "cimrgp_profile_begin")     # the same kernel without carried rows beside it
    for rep in range(7):                          # 2 warm-ups (the host reads above let the clocks drop), median of 5
        dev.rbf_gram(xd, ell, sf2, noise, lower_only=True, out=kbuf)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        dev.potrf(kbuf, n, ws, info)
        c1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            chol_ms.append(c0.elapsed_time(c1))
    chol_ms = float(np.median(chol_ms))
    al_ms, al_fl, al_by, al_cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.cimrgp_profile_collect_bytes(ctypes.byref(al_ms), ctypes.byref(al_fl), ctypes.byref(al_by), ctypes.byref(al_cnt)),
               "cimrgp_profile_collect_bytes")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if cabi else device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    assert int(info.item()) == 0

    if rank == 0:
        peak = FP64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
        achieved = (tr_fl.value / (tr_ms.value * 1e-3)) / 1e12 if tr_ms.value > 0 else 0.0
        chol_gflops = (n ** 3 / 3.0) / (chol_ms * 1e-3) / 1e9
        out = {
            "metric": "GP posteriors/sec at N=%d (Gram + Cholesky + solve + predictive mean/var)" % n,
            "value": world * args.steps / dt,
            "unit": "posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 1-D, N=%d single-resolution single-partition RBF GP "
                                   "per GPU, q=2 outputs, N/4=%d test points, ell=0.1, sf2=1, noise=0.01" % (n, ns),
                       "partitions_per_gpu": 1,
                       "parallelism": "independent partitions, 1 all-reduce/step (started at the end of a step, waited for by the stream at the start of the next)",
                       "backend": ((rehearsal or ("rccl through the C ABI (cimrgp_allreduce_sum on the step's solve queue); control plane gloo"
                                                   if comm is not None else ("nccl (fallback: the C-ABI communicator could not be created)"
                                                                             if fallback_group is not None else "nccl"))) if td.is_initialized() else "none"),
                       "rows_queues": int(lib.cimrgp_get_rows_queues()),
                       "front_queue": bool(pipeline and args.front_queue)},
            "reduce_selfcheck_max_abs_diff": reduce_diff,
            "cholesky_gflops": chol_gflops,
            "cholesky_frac_of_peak": chol_gflops / 1e3 / peak,
            "step_is": ("ONE cimrgp_block_posterior_staged call per step on preallocated buffers (+ the collective's enqueue): consecutive steps "
                        "are independent blocks over two rotating buffer sets; the backward solve and prediction of step i run on the "
                        "look-ahead context's idle queue (cimrgp_solve_queue) beside the first panels of step i+1, whose Gram matrices run on the "
                        "context's chain queue (cimrgp_front_queue) beside the last panels of step i; "
                        "every step completes inside the timed region")
                       if pipeline else "ONE cimrgp_block_posterior call per step on preallocated buffers (+ the collective's enqueue)",
            "pipelined_steps": bool(pipeline),
            "reduce_overlapped": bool(world > 1 or args.nccl_world1),
            "drained_step_ms": drained_ms,
            "stage_ms": {"measured": "the same work as five separate calls on ONE stream, after the timed region (median of 3): a different "
                                     "schedule from the timed, pipelined step -- its stages do not add up to ms_per_step",
                         "gram": float(stage_ms[0]), "cross_gram_and_rhs_rows": float(stage_ms[1]),
                         "potrf_with_carried_rows": float(stage_ms[2]), "backward_solve_and_predict": float(stage_ms[3]),
                         "potrf_alone": chol_ms},
            "gram_gbps_lower": (n * (n + 1) / 2 * (8 if args.dtype == "f64" else 4)) / (stage_ms[0] * 1e-3) / 1e9,
            "roofline": {"bound": "mfma", "kernel": "lower trailing updates of cimrgp_potrf: k_gemm_nt_pers<%s, lower> (look-ahead phase: head + bulk in "
                                                    "one persistent launch) and k_gemm_nt_sub<%s, lower, *>"
                                                    % (("double", "double") if args.dtype == "f64" else ("float", "float")),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "launches": int(tr_cnt.value), "timed_steps_sampled": "every %d-th of %d" % (EVENT_EVERY, args.steps),
                         "avg_launch_ms": tr_ms.value / max(1, tr_cnt.value),
                         "avg_launch_gflop": tr_fl.value / max(1, tr_cnt.value) / 1e9,
                         "traffic": None,
                         # the same launches in the factorisation WITHOUT carried rows (the three potrf_alone runs
                         # after the timed steps): in the step the updates share the machine with the rows' updates
                         "without_carried_rows": {
                             "achieved": (al_fl.value / (al_ms.value * 1e-3)) / 1e12 if al_ms.value > 0 else 0.0,
                             "frac": ((al_fl.value / (al_ms.value * 1e-3)) / 1e12 / peak) if al_ms.value > 0 else 0.0,
                             "launches": int(al_cnt.value), "avg_launch_ms": al_ms.value / max(1, al_cnt.value)}},
        }
        # HBM traffic and matrix-core busy fraction of that kernel, and the Gram builder's written bytes, come from
        # committed rocprofv3 --pmc passes of this same command (counters cannot be collected from inside the process
        # being timed).  The profile names the source hash it was collected from: when the kernels have changed
        # since, the counters are reported as stale under `from_profile` and NOT put into the roofline.
        out["roofline"]["algorithmic_bytes_per_launch"] = tr_by.value / max(1, tr_cnt.value)
        # The whole step against the same peak: every matrix-core flop of a step (the factorisation n^3/3, the rows carried
        # through it (ns + q) n^2, the backward solve q n^2) over ms_per_step -- the updates above share the machine with
        # the carried rows' updates, so the per-launch fraction understates how busy the matrix cores are kept.
        step_flops = n ** 3 / 3.0 + (ns + q) * float(n) ** 2 + q * float(n) ** 2
        step_tflops = step_flops / (dt / args.steps) / 1e12          # per GPU (every rank runs one step per step)
        out["roofline"]["whole_step"] = {"flops_per_step": step_flops, "achieved": step_tflops, "frac": step_tflops / peak}
        pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if args.dtype == "f64" and n == 8192 and os.path.exists(pmc):
            with open(pmc) as fh:
                pj = json.load(fh)
            fresh = pj.get("sources_sha256") == _sources_sha()
            fp = {"file": "profiles/" + PMC_PROFILE, "commit": pj.get("commit"), "sources_sha256": pj.get("sources_sha256"),
                  "matches_this_code": fresh,
                  "traffic_bytes_per_launch": pj.get("trailing_update", {}).get("traffic_bytes_per_launch"),
                  "gram_hbm_write_gbps_rocprof": pj.get("gram", {}).get("hbm_write_GBps_rocprof"),
                  "gram_frac_of_hbm_peak": pj.get("gram", {}).get("frac_of_8TBps")}
            out["from_profile"] = fp
            if fresh:
                out["roofline"]["traffic"] = fp["traffic_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = ("bytes/launch (PMC upper bound: 2*FETCH_SIZE + WRITE_SIZE, separate passes; "
                                                   "%s, commit %s)" % (fp["file"], fp["commit"]))
        if world == 1 and not args.no_cpu_baseline:
            rec, omean, ovar = cpu_baseline(n, ns, q, ell, sf2, noise, 1234, args.cpu_warmups, args.cpu_repeats)
            out["cpu_baseline"] = rec
            # full-size parity of the headline config: the last timed step against the oracle's
            # posterior on the same arrays (north_star: 1e-5 relative on mean and variance, max
            # error over max magnitude as in tests/)
            out.update(parity_fields(last_mean, last_var, omean, ovar))
        print(json.dumps(out))
        if out.get("parity_ok") is False:
            raise SystemExit("bench.py: the GPU posterior differs from the oracle by more than 1e-5: mean %.3e var %.3e"
                             % (out["parity_rel_err_mean"], out["parity_rel_err_var"]))
    if comm is not None:
        comm.close()
    if td.is_initialized():
        td.destroy_process_group()


if __name__ == "__main__":
    main()
