#!/usr/bin/env python3
"""bench.py -- headline benchmark: process-tomography MLE reconstructions/sec (2-qubit, 100 iters).

Workload (BASELINE.json configs[1]): a batch of 1024 independent 2-qubit process tomographies
per GPU (Pauli in-basis, 540 settings, 1000 shots), 100 fixed outer iterations of projected
gradient descent with backtracking (fbx_pgdb_process_dev, FBX_MODE_FIXED), fp64.  Inputs are
resident in HBM before the timed region; a "step" is one pass of the kernel over the batch.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; the batch axis is embarrassingly parallel, so ranks shard items with no
data-path collective (scaling = weak: every rank owns `--batch` items).  torch.distributed is
used only for the barrier and the max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))

# algorithmic work of one 2-qubit, 100-iteration reconstruction in the
# reference's dense formulation (~7.7 MFLOP per outer iteration) and its HBM bytes
# (540 expectations + 540 counts in, 16x16 complex128 Choi out).
ALGO_FLOP_PER_RECON = 0.77e9
ALGO_BYTES_PER_RECON = 12736
FP64_PEAK_TFLOPS = 78.6          # MI355X fp64 vector == fp64 MFMA (v_mfma_f64_16x16x4) dense peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="reconstructions per GPU per step")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling (BASELINE configs[4]): this many reconstructions in total, "
                         "block-partitioned over the ranks (e.g. 65536); 0 = weak scaling with --batch per GPU")
    ap.add_argument("--distinct-shards", action="store_true",
                    help="weak scaling with a different block of synthetic experiments on every rank "
                         "(default: every rank runs the N = 1 workload)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--in-basis", default=None, choices=["pauli", "sic"],
                    help="input-state basis of the process design (default: pauli for pgdb, sic for pgdb3)")
    ap.add_argument("--cpu-sample", type=int, default=12,
                    help="items timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--workload", default="pgdb", choices=["pgdb", "sweep", "pgdb3"],
                    help="pgdb = the headline metric (BASELINE configs[1]); sweep = the secondary "
                         "HBM-bound conversion sweep of BASELINE configs[2] (1e6 Kraus sets); pgdb3 = BASELINE "
                         "configs[3], 256 three-qubit process tomographies")
    ap.add_argument("--sweep-items", type=int, default=1_000_000)
    return ap.parse_args()


def cpu_baseline(design, e, c, n_items, iters):
    """The oracle (numpy restatement of the reference) on a bounded sample of the same batch,
    one core, design matrix hoisted out of the loop (the fair variant of BASELINE.md section 3)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from fbx_oracle import design as od, estimators as oe
    d = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    t0 = time.perf_counter()
    for b in range(n_items):
        oe.pgdb_process_estimate(d, e[b], c[b], mode="fixed", max_iters=iters, A=A)
    dt = time.perf_counter() - t0
    return {"value": n_items / dt, "unit": "reconstructions/s", "cores": 1, "kind": "port",
            "sample": f"first {n_items} items of the bench batch, fixed {iters} iterations, "
                      f"numpy oracle with the design matrix hoisted, {dt:.1f} s"}


_POOL_WORKER = r"""
import os, sys, time
os.environ["OMP_NUM_THREADS"] = "1"; os.environ["OPENBLAS_NUM_THREADS"] = "1"; os.environ["MKL_NUM_THREADS"] = "1"
import numpy as np
sys.path.insert(0, sys.argv[1])
from fbx_oracle import design as od, estimators as oe
z = np.load(sys.argv[2]); lo, hi, iters = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
d = od.Design(int(z["n"]), "process", z["in_labels"], z["paulis"], z["coefs"])
A = oe.design_matrix_A(d)
t0 = time.perf_counter()
for b in range(lo, hi):
    oe.pgdb_process_estimate(d, z["e"][b], z["c"][b], mode="fixed", max_iters=iters, A=A)
print(time.perf_counter() - t0)
"""


def cpu_baseline_pool(design, e, c, iters, per_core=1, max_cores=64):
    """The fair multi-core variant of SURVEY.md 8d: one single-threaded oracle process per host core
    (separate interpreters -- nothing is forked from the process that owns the GPU), `per_core`
    items each, design matrix hoisted once per process and left out of the timed region."""
    import subprocess, tempfile
    cores = max(1, min(os.cpu_count() or 1, max_cores, e.shape[0] // per_core))
    n_items = cores * per_core
    with tempfile.TemporaryDirectory() as tmp:
        fn = os.path.join(tmp, "sample.npz")
        np.savez(fn, n=design.n_qubits, in_labels=design.in_labels, paulis=design.paulis, coefs=design.coefs,
                 e=e[:n_items], c=c[:n_items])
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", _POOL_WORKER, os.path.join(ROOT, "oracle"), fn,
                                   str(k * per_core), str((k + 1) * per_core), str(iters)],
                                  stdout=subprocess.PIPE, text=True) for k in range(cores)]
        inner = [float(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
        wall = time.perf_counter() - t0
    busy = max(inner)                      # slowest worker's reconstruction time, start-up excluded
    return {"value": n_items / busy, "unit": "reconstructions/s", "cores": cores, "kind": "port",
            "sample": f"{n_items} items of the bench batch, {per_core} per single-threaded oracle process, "
                      f"fixed {iters} iterations, slowest worker {busy:.1f} s (wall incl. start-up {wall:.1f} s)"}


def _profiled(key):
    """HBM bytes per launch measured with rocprofv3 PMC passes (profiles/pmc_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def sweep_cpu_baseline(ks, ref, n_items):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from fbx_oracle import superops as so, measures as om
    t0 = time.perf_counter()
    for b in range(n_items):
        choi = so.kraus2choi(list(ks[b]))
        ptm = so.choi2pauli_liouville(choi)
        so.choi2chi(choi)
        om.process_fidelity(ref, ptm)
    dt = time.perf_counter() - t0
    return {"value": n_items / dt, "unit": "items/s", "cores": 1, "kind": "port",
            "sample": f"first {n_items} Kraus sets, numpy oracle (reference-faithful: basis matrices "
                      f"rebuilt per call, choi2chi through eigh), {dt:.1f} s"}


def run_sweep(args, rank, world, dist, torch):
    """Secondary line: kraus2choi -> choi2pauli_liouville -> choi2chi + process_fidelity on
    `--sweep-items` random 2-qubit CPTP Kraus sets (K = 4) per GPU, inputs resident in HBM."""
    import ctypes
    from fbx import _lib, synthetic
    n, K, D = 2, 4, 16
    B = args.sweep_items
    base = synthetic.kraus_batch(n, K, 8192, seed=17 + rank)
    ks = np.ascontiguousarray(np.tile(base, (B // 8192 + 1, 1, 1, 1))[:B])
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.complex128)
    lib = _lib.lib()
    ref = np.empty((1, D, D), dtype=np.complex128)
    _lib.check(lib.fbx_convert(_lib.REP_KRAUS, _lib.REP_PAULI_LIOUVILLE, n, 1,
                               _lib.dptr(np.ascontiguousarray(cnot[None, None]).view(np.float64)), 1,
                               _lib.dptr(ref.view(np.float64))))
    d_k = _lib.DeviceBuffer.from_array(ks)
    d_r = _lib.DeviceBuffer.from_array(ref)
    d_c = _lib.DeviceBuffer(B * D * D * 16); d_p = _lib.DeviceBuffer(B * D * D * 16)
    d_x = _lib.DeviceBuffer(B * D * D * 16); d_f = _lib.DeviceBuffer(B * 8)

    def step():
        _lib.check(lib.fbx_kraus_sweep_dev(n, B, K, d_k.ptr, d_r.ptr, d_c.ptr, d_p.ptr, d_x.ptr, d_f.ptr))

    def barrier():
        if dist is not None:
            dist.barrier()
        _lib.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ms = ctypes.c_double(0.0)
    t0 = time.perf_counter()
    _lib.check(lib.fbx_timer_begin())
    for _ in range(args.steps):
        step()
    _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, ms.value], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kms = t.tolist()
    else:
        kms = ms.value
    if rank == 0:
        bytes_item = K * D * 16 + 3 * D * D * 16 + 8            # 13 320 B
        ksec = kms / 1e3 / args.steps
        gbs = B * bytes_item / ksec / 1e9
        line = {"metric": "conversion sweep items/sec (2-qubit Kraus -> Choi -> PTM -> chi + process_fidelity)",
                "value": world * B * args.steps / elapsed, "unit": "items/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"{B} random 2-qubit CPTP Kraus sets (K=4) per GPU, all three "
                                       f"representations + fidelity written, inputs resident in HBM",
                           "items_per_gpu": B, "parallelism": f"shard{world}"},
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": _profiled("sweep_kernel_hbm_bytes_per_launch"),
                             "kernel": "sweep2q_pair_kernel",
                             "kernel_ms": 1e3 * ksec,
                             "note": "achieved = 13 320 algorithmic bytes per item / HIP-event kernel time"}}
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = sweep_cpu_baseline(ks, ref[0], 2000)
        print(json.dumps(line), flush=True)


def run_pgdb3(args, rank, world, dist, torch):
    """Third line: BASELINE configs[3] -- 3-qubit (64 x 64 Choi) PGDB process tomography, batch 256 per
    GPU, SIC in-basis (4032 settings) unless --in-basis pauli (13 608), 100 fixed iterations."""
    import ctypes
    from fbx import _lib, synthetic
    B = 256
    basis = args.in_basis or "sic"
    design, _, e, c = synthetic.process_batch(3, basis, 32)
    e = np.tile(e, (B // 32, 1)); c = np.tile(c, (B // 32, 1))
    lib = _lib.lib()
    d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
    d_choi = _lib.DeviceBuffer(B * 64 * 64 * 16); d_it = _lib.DeviceBuffer(B * 4); d_dy = _lib.DeviceBuffer(B * 4)

    def step():
        _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED, args.iters,
                                            d_choi.ptr, d_it.ptr, d_dy.ptr, None, None))

    def barrier():
        if dist is not None:
            dist.barrier()
        _lib.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ms = ctypes.c_double(0.0)
    t0 = time.perf_counter()
    _lib.check(lib.fbx_timer_begin())
    for _ in range(args.steps):
        step()
    _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, ms.value], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kms = t.tolist()
    else:
        kms = ms.value
    if rank == 0:
        dyk = d_dy.to_array(np.int32, (B,))
        ksec = kms / 1e3 / args.steps
        # algorithmic flops in the reference's dense formulation (SURVEY 8d recipe at n = 3): per outer
        # iteration 3 R D^2 complex MACs (gradient 2, cost 1; R = 2 m rows, D^2 = 4096) + per Dykstra
        # iteration one 64 x 64 Hermitian eigendecomposition (~25 N^3 = 6.6 MFLOP) + V L V^H (2 N^3 cmac)
        m = design.m
        flop = args.iters * 3 * (2 * m) * 4096 * 8 + float(dyk.mean()) * (25 * 64 ** 3 + 2 * 64 ** 3 * 8)
        tflops = B * flop / ksec / 1e12
        line = {"metric": "process-tomography MLE reconstructions/sec (3-qubit, 64x64 Choi, 100 iters)",
                "value": world * B * args.steps / elapsed, "unit": "reconstructions/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"{B} independent 3-qubit process tomographies per GPU, {basis} in-basis "
                                       f"({m} settings, 1000 shots), {args.iters} fixed PGDB iterations, inputs "
                                       f"resident in HBM", "batch_per_gpu": B, "iters": args.iters,
                           "parallelism": f"shard{world}", "mean_dykstra_iters": float(dyk.mean())},
                "roofline": {"bound": "mfma", "pipe": "fp64 VALU + LDS", "achieved": tflops, "peak": FP64_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS, "traffic": None,
                             "kernel": "pgdb3_kernel", "kernel_ms": 1e3 * ksec,
                             "note": "achieved = algorithmic flops of the dense formulation (3 x 2m x 4096 complex "
                                     "MACs per outer iteration + ~25 N^3 per 64 x 64 eigendecomposition) / HIP-event "
                                     "kernel time; the kernel is co-limited by LDS bandwidth and fp64 issue in "
                                     "the eigensolver (DESIGN.md 2.2)"}}
        print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.workload == "pgdb" and args.in_basis is None:
        args.in_basis = "pauli"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # torch first: its bundled libamdhip64.so.7 and ours share one SONAME, so loading torch
    # before libfbx.so keeps a single HIP runtime in the process.
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from fbx import _lib, synthetic
    _lib.set_device(local_rank)                       # fails loudly without a GPU
    dev_name, cus = _lib.device_name()
    if args.workload == "pgdb3":
        run_pgdb3(args, rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "sweep":
        run_sweep(args, rank, world, dist, torch)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    scaling = "weak"
    if args.total_batch > 0:                          # contiguous block partition of the batch axis
        from fbx.parallel import shard_bounds
        lo, hi = shard_bounds(args.total_batch, rank, world)
        B, first = hi - lo, lo
        scaling = "strong"
    else:
        # weak scaling: every rank reconstructs the SAME --batch synthetic experiments (seeds 1000 ..
        # 1000 + batch - 1, the N = 1 workload), so that per-GPU work really is fixed as N grows; the
        # 1024-item blocks of consecutive seeds differ by +-15 % in kernel time (one slow item decides,
        # scripts/block_spread.py), which would otherwise read as scaling loss.  --distinct-shards
        # gives every rank its own block of seeds instead.
        B, first = args.batch, (rank * args.batch if args.distinct_shards else 0)
    # distinct synthetic items are generated for up to 4096 per rank and tiled beyond that
    n_distinct = min(B, 4096)
    design, _, e, c = synthetic.process_batch(2, args.in_basis, n_distinct, first_item=first)
    if n_distinct < B:
        reps = -(-B // n_distinct)
        e = np.tile(e, (reps, 1))[:B]; c = np.tile(c, (reps, 1))[:B]
    d_e = _lib.DeviceBuffer.from_array(e)
    d_c = _lib.DeviceBuffer.from_array(c)
    D = 16
    d_choi = _lib.DeviceBuffer(B * D * D * 16)
    d_it = _lib.DeviceBuffer(B * 4)
    d_dy = _lib.DeviceBuffer(B * 4)
    d_bt = _lib.DeviceBuffer(B * 4)
    d_cost = _lib.DeviceBuffer(B * 8)
    lib = _lib.lib()

    def step():
        _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED,
                                            args.iters, d_choi.ptr, d_it.ptr, d_dy.ptr, d_bt.ptr,
                                            d_cost.ptr))

    def barrier():
        if dist is not None:
            dist.barrier()
        _lib.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    import ctypes
    ms = ctypes.c_double(0.0)
    t0 = time.perf_counter()
    _lib.check(lib.fbx_timer_begin())
    for _ in range(args.steps):
        step()
    _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))     # HIP events on the launch stream
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed, ms.value], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total = t.tolist()
    else:
        kernel_ms_total = ms.value

    iters = d_it.to_array(np.int32, (B,))
    dyk = d_dy.to_array(np.int32, (B,))
    bt = d_bt.to_array(np.int32, (B,))

    if rank == 0:
        total_recons = (args.total_batch if args.total_batch > 0 else world * B) * args.steps
        value = total_recons / elapsed
        kernel_s = kernel_ms_total / 1e3 / args.steps           # average launch duration
        achieved_tflops = B * ALGO_FLOP_PER_RECON * (args.iters / 100.0) / kernel_s / 1e12
        traffic = _profiled("pgdb_kernel_hbm_bytes_per_launch")
        line = {
            "metric": "process-tomography MLE reconstructions/sec (2-qubit, 100 iters)",
            "value": value, "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{B} independent 2-qubit process tomographies per GPU, "
                                   f"{args.in_basis} in-basis ({design.m} settings, 1000 shots), "
                                   f"{args.iters} fixed PGDB iterations, inputs resident in HBM",
                       "batch_per_gpu": B, "iters": args.iters, "parallelism": f"shard{world}",
                       "shards": ("distinct seeds per rank" if (args.distinct_shards or args.total_batch > 0)
                                  else "same experiments on every rank"),
                       "device": dev_name.strip(), "compute_units": cus,
                       "mean_dykstra_iters": float(dyk.mean()),
                       "mean_backtracks": float(bt.mean()),
                       "mean_outer_iters": float(iters.mean())},
            "roofline": {"bound": "mfma", "pipe": "fp64 VALU + MFMA", "achieved": achieved_tflops, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / FP64_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": "pgdb_kernel<2,9>", "kernel_ms": 1e3 * kernel_s,
                         "note": "PGDB is fp64-compute bound (SURVEY.md 8d): achieved = "
                                 "0.77 GFLOP algorithmic (dense-A formulation) x batch / HIP-event "
                                 "kernel time; peak = dense fp64 MFMA peak of MI355X, which equals its fp64 "
                                 "vector peak -- the Jacobi rotations are VALU FMAs, the warm-start "
                                 "basis change of every eigendecomposition runs on the fp64 MFMA pipe",
                         "hbm": {"achieved": B * ALGO_BYTES_PER_RECON / kernel_s / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": B * ALGO_BYTES_PER_RECON / kernel_s / 1e9 / HBM_PEAK_GBS}},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(design, e, c, min(args.cpu_sample, B), args.iters)
            line["cpu_baseline_multicore"] = cpu_baseline_pool(design, e, c, args.iters)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
