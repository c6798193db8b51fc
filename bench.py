#!/usr/bin/env python3
"""bench.py -- process-tomography MLE reconstructions/sec (2-qubit, 100 iters) on MI355X.

    python bench.py                                   # N = 1: headline + the secondary workloads
    python bench.py --gpus N --steps K --warmup W     # spawns its N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N ...         # same ranks, launched by torchrun

A "step" is one pass of the hot path (fbx_pgdb_process_dev, FBX_MODE_FIXED, 100 outer iterations
of projected gradient descent with backtracking, fp64) over one batch that is already resident in HBM.

OUTPUT: one short JSON line per secondary workload / parity leg first, and -- LAST -- the compact headline object the driver
parses (< 4 KB: the contract's keys, `config`, `roofline`, `cpu_baseline`); the full record goes to gpurun_out/bench_detail.json
(--detail-out).  See DESIGN.md section 5 and tests/test_bench_line.py.

N = 1 (BASELINE.json configs[1], the configuration the metric is quoted on): 1024 independent 2-qubit
process tomographies (Pauli in-basis, 540 settings, 1000 shots).  The same run also times, outside
the headline's timed region and reported as secondary lines: the conversion sweeps of configs[2]
(10^6 two-qubit / 65 536 three-qubit Kraus sets), the 3-qubit PGDB of configs[3] (batch 256, both in-bases), the
single-qubit PGDB (2^20 experiments to convergence), the state-estimator half of the path (iterative MLE, 2 and 3 qubits,
maxiter 100), the caller-side shot reduction (shots -> moments), configs[4]'s whole batch on one GPU, the converge-mode
(reference semantics) throughput, the host-pointer (H2D + D2H inclusive) rate, the latency of one experiment
through the reference-signature call, the parity of the timed items against the committed reference fixtures and the oracle,
and the CPU baselines (oracle = numpy restatement of the reference, timed on this box's host cores).
`--workload pgdb | sweep | sweep3 | pgdb3 | pgdb1 | mle_state | mle_state3 | shots` runs one of them as the primary line.

N > 1 (BASELINE.json configs[4]): 65 536 two-qubit tomographies with distinct seeds, block-partitioned
over the ranks (strong scaling: the total is fixed for N = 2, 4, 8), no data-path collective.  The
headline's config also carries the weak-scaling figure with configs[1]'s 1024 items on every rank
("per_gpu_1024"), which is the number comparable with the N = 1 headline.

No torch: ranks meet through fbx.parallel (RCCL communicator inside libfbx.so; rank 0's unique id
travels through a rendezvous directory); barrier and max-over-ranks of the elapsed time are RCCL
all-reduces.  When RCCL cannot be initialised (e.g. test ranks sharing one GPU with
--oversubscribe) the barrier falls back to the rendezvous files and the line says so.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))

# algorithmic work of one 2-qubit, 100-iteration reconstruction in the reference's
# dense formulation (~7.7 MFLOP per outer iteration) and its HBM bytes (540 expectations + 540
# counts in, 16 x 16 complex128 Choi out).
ALGO_FLOP_PER_RECON = 0.77e9
FP64_PEAK_TFLOPS = 78.6          # MI355X fp64 vector == fp64 MFMA (v_mfma_f64_16x16x4) dense peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="reconstructions per GPU per step (weak form)")
    ap.add_argument("--total-batch", type=int, default=-1,
                    help="reconstructions in total, block-partitioned over the ranks; default: 65536 "
                         "(BASELINE configs[4]) when --gpus > 1, 0 = weak scaling with --batch per GPU")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--in-basis", default=None, choices=["pauli", "sic"],
                    help="input-state basis of the process design (default: pauli for pgdb, sic for pgdb3)")
    ap.add_argument("--cpu-sample", type=int, default=16,
                    help="items run through the oracle for cpu_baseline, its reference-faithful and multi-core variants and "
                         "the parity self-check (0 = skip every CPU leg; ~25 s of one core at the default)")
    ap.add_argument("--workload", default="all", choices=["all", "pgdb", "sweep", "sweep3", "pgdb3", "pgdb1", "mle_state", "mle_state3", "shots"],
                    help="all (N = 1 default) = headline pgdb + every secondary leg; pgdb = headline only; "
                         "sweep / pgdb3 / pgdb1 = that workload as the primary line")
    ap.add_argument("--sweep-items", type=int, default=1_000_000)
    ap.add_argument("--anchor-items", type=int, default=65536,
                    help="N = 1, workload all: also time this many items (BASELINE configs[4]'s whole batch) on the one "
                         "GPU -- the same-workload anchor of the 1/2/4/8-GPU strong-scaling curve (0 = skip)")
    ap.add_argument("--spawn-timeout", type=float, default=1800.0,
                    help="seconds the self-spawned ranks (--gpus > 1 without a launcher) may take before they are killed")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the FULL record goes (every leg with its notes); stdout carries one short line per leg and, last, "
                         "the compact headline the driver parses")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing: allow more ranks than visible GPUs (ranks share devices; host-file barrier)")
    return ap.parse_args()


# ================================================================================== rank spawning
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) with the
    environment torchrun would give them and relay rank 0's line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdzv = tempfile.mkdtemp(prefix="fbx_rdzv_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBX_RDZV_DIR=rdzv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank stuck in a collective (RCCL bootstrap waiting for a peer that died) must not hang the job
    deadline = time.monotonic() + args.spawn_timeout
    try:
        out, _ = procs[0].communicate(timeout=args.spawn_timeout)
        codes = [procs[0].returncode] + [p.wait(timeout=max(deadline - time.monotonic(), 1.0)) for p in procs[1:]]
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()                                            # exactly the processes started above
        sys.exit(f"bench.py: ranks did not finish within {args.spawn_timeout:.0f} s "
                 f"(exit codes so far {[p.poll() for p in procs]})")
    sys.stdout.write(out)
    sys.stdout.flush()
    try:
        os.rmdir(rdzv)
    except OSError:
        pass
    if any(codes):
        sys.exit(f"bench.py: rank exit codes {codes}")


# ================================================================================== timing
def timed_steps(step, steps, warmup, comm, _lib):
    """W untimed steps, then exactly K steps bracketed by barrier + stream synchronisation on both
    sides; returns (wall seconds, HIP-event milliseconds on the launch stream), max over ranks."""
    for _ in range(warmup):
        step()
    comm.barrier()
    ms = ctypes.c_double(0.0)
    t0 = time.perf_counter()
    _lib.check(_lib.lib().fbx_timer_begin())
    for _ in range(steps):
        step()
    _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))      # HIP events on the stream the kernels run on
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, kms = comm.allreduce([elapsed, ms.value], "max")
    return float(elapsed), float(kms)


def _profiled(key):
    """HBM bytes per launch measured with rocprofv3 PMC passes (profiles/pmc_traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def _measured_mfma_flop(kernel, items):
    """fp64 MFMA operations of one launch counted by the hardware (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512), per item, or None."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_flops.json")))[f"{kernel}@{items}"]
        return float(rec["mfma_flop"]) / float(rec["items_per_launch"])
    except Exception:
        return None


def _measured_flop(kernel, items):
    """fp64 operations of one launch of `kernel` COUNTED by the hardware (profiles/pmc_flops.json, written by
    scripts/profile_flops.py from a `rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64
    SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64` pass over this very command): wave-instructions x 64 lanes (fma = 2;
    lanes masked off by EXEC are counted as if active -- an upper bound on useful work) + MFMA operations.  Returns flop per
    item of a launch of `items` items, or None when no such pass is on file."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_flops.json")))[f"{kernel}@{items}"]
        return float(rec["flop_per_launch"]) / float(rec["items_per_launch"])
    except Exception:
        return None


# ================================================================================== flop accounting
def pgdb2q_executed_flop(m, S, iters, dyk, backtracks, work, two_workers=False):
    """Floating-point operations the 2-qubit kernel EXECUTES for one reconstruction (mean over the
    batch), counted from the source (csrc/fbx_pgdb.hip, fbx_choi.hpp, fbx_eigh.hpp; fma = 2, DESIGN.md
    2.1) and the per-item work counters the kernel returns.  Kronecker formulation: the dense design
    matrix A of the reference (the 0.77 GFLOP numerator) is never applied."""
    D, lanes = 16, 64
    sweeps, terms, cost_evals, sum_passes = (float(np.mean(work[:, k])) for k in range(4))
    it, dy, bt = float(np.mean(iters)), float(np.mean(dyk)), float(np.mean(backtracks))
    f = {}
    # rotation 42 + 2x2 block update 80 + eigenvector update 40 per lane and round; the one-wave kernels (B <= 1024) update the
    # Hermitian work matrix with two workers per upper block since round 5 (csrc/fbx_eigh.hpp): half a block update per lane
    f["jacobi_sweeps"] = sweeps * 15 * lanes * (122 if two_workers else 162)
    f["jacobi_offnorm_tests"] = (sweeps + dy) * lanes * 20
    f["basis_change_mfma"] = max(dy - np.ceil(it / 16.0), 0.0) * 32 * 2 * 16 * 16 * 4   # 32 v_mfma_f64_16x16x4 per warm decomposition
    f["reconstruct"] = terms * lanes * 28                          # V diag(lam+) V^H, one rank-1 term per kept eigenvalue
    f["dykstra_rest"] = dy * lanes * 220                           # Hermitise, differences, TP projection, stop functional
    f["pauli_transforms"] = it * 3 * 3300                          # estimate, update direction, gradient (4 butterfly stages each)
    f["prediction_tables"] = it * 2 * 2 * S * D * D                # T = R C for estimate and update direction
    f["gradient"] = it * (30 * m + 2 * S * D * D)                  # eta = n / p, LDS-atomic weights, W C^T
    f["probabilities"] = it * 3 * 6 * m
    f["cost_evaluations"] = cost_evals * 2 * m * 45                # clip + log (~40) + accumulate per outcome
    f["power_sums"] = sum_passes * 2 * m * 40
    f["series_steps"] = bt * lanes * 34
    return sum(f.values()), f


def pgdb3q_executed_flop(m, S, iters, dyk, work):
    sweeps, terms, cost_evals = (float(np.mean(work[:, k])) for k in range(3))
    it, dy = float(np.mean(iters)), float(np.mean(dyk))
    f = {}
    # a round of the role-split solver with published rotations (csrc/fbx_eigh64.hpp, round 5): 496 matrix threads x (2x2 block
    # update 80) + 32 of them x (next round's rotation 42) + 512 eigenvector threads x (two block updates 2 x 40).  Round 4, every
    # thread evaluating the rotations it applies: 496 x 164 + 512 x 122; rounds 1-3: 1024 threads x 162
    f["jacobi_sweeps"] = sweeps * 63 * (496 * 80 + 32 * 42 + 512 * 80)
    f["basis_change_mfma"] = dy * (1 + 10 / 16) * 64 ** 3 * 8      # V^H H V on the fp64 matrix cores: H V in full, of V^H (H V) the ten upper tiles of sixteen
    f["reconstruct"] = terms * 1024 * 28
    f["dykstra_rest"] = dy * 1024 * 240
    f["tables_mfma"] = it * 3 * 2 * S * 64 * 64                    # T = C^T R^T (x2) and R^G = -W C^T / d^2 on the fp64 matrix cores
    f["gradient_probabilities"] = it * 48 * m
    f["cost_evaluations"] = cost_evals * 2 * m * 45
    f["pauli_transforms"] = it * 3 * 6 * 4096 * 4
    return sum(f.values()), f


# ================================================================================== CPU legs (oracle)
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from fbx_oracle import design as od, estimators as oe, superops as so, measures as om
    return od, oe, so, om


def cpu_baseline_and_parity(design, us, e, c, n_items, iters, gpu_fixed, gpu_conv):
    """The oracle on the first `n_items` items of the bench batch, one core.  Returns the cpu_baseline
    object (fixed mode, design matrix hoisted -- the fair variant of BASELINE.md section 3), the
    converge-mode and reference-faithful (design rebuilt per call, tomography.py:494-539) variants,
    and the parity self-check of the GPU results of the timed launch against the oracle."""
    od, oe, so, om = _oracle()
    d = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)

    def fid(choi, b):
        return om.process_fidelity(so.kraus2pauli_liouville([us[b]]), so.choi2pauli_liouville(choi))

    def compare(mode, gpu, label):
        choi_g, st_g = gpu
        t0 = time.perf_counter()
        res = [oe.pgdb_process_estimate(d, e[b], c[b], mode=mode, max_iters=iters if mode == "fixed" else 0,
                                        A=A, return_stats=True) for b in range(n_items)]
        dt = time.perf_counter() - t0
        dchoi = max(float(np.abs(choi_g[b] - res[b][0]).max()) for b in range(n_items))
        dfid = max(abs(fid(choi_g[b], b) - fid(res[b][0], b)) for b in range(n_items))
        mism = {k: int(sum(int(st_g[k][b]) != res[b][1][k] for b in range(n_items)))
                for k in ("iterations", "dykstra", "backtracks")}
        return dt, {"mode": label, "items": n_items, "max_abs_choi_diff": dchoi,
                    "max_process_fidelity_diff": dfid, "count_mismatches": mism,
                    "mean_oracle_iterations": float(np.mean([r[1]["iterations"] for r in res]))}

    dt_fixed, par_fixed = compare("fixed", gpu_fixed, f"fixed {iters} iterations (the timed mode)")
    dt_conv, par_conv = compare("converge", gpu_conv, "converge (reference semantics, tomography.py:589)")
    # "what a forest-benchmarking user gets today" (variant 1): one experiment at a time, the design matrix
    # rebuilt inside every call as tomography.py:494-539 does, to the reference's own stopping rule (it has no iteration cap)
    n_faith = n_items
    t0 = time.perf_counter()
    for b in range(n_faith):
        oe.pgdb_process_estimate(d, e[b], c[b], mode="converge")          # A rebuilt inside, as the reference does
    dt_faith = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in range(min(4, n_items)):
        oe.design_matrix_A(d)
    dt_A = (time.perf_counter() - t0) / min(4, n_items)
    base = {"value": n_items / dt_fixed, "unit": "reconstructions/s", "cores": 1, "kind": "port",
            "sample": f"first {n_items} items of the bench batch, fixed {iters} iterations, numpy oracle with "
                      f"the design matrix hoisted, {dt_fixed:.1f} s",
            "converge_mode": {"value": n_items / dt_conv, "unit": "reconstructions/s",
                              "sample": f"same items to convergence, {dt_conv:.1f} s"},
            "reference_faithful": {"value": n_faith / dt_faith, "unit": "reconstructions/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_faith} items to convergence (the reference's stopping rule, "
                                             f"tomography.py:589) with the design matrix rebuilt per call as "
                                             f"tomography.py:494-539 does ({dt_A:.2f} s of each call), {dt_faith:.1f} s",
                                   "fixed_iters_equivalent": n_items / (dt_fixed + n_items * dt_A)}}
    return base, [par_fixed, par_conv]


def fixture_parity(batch, gpu_fixed, gpu_conv, iters):
    """The timed launch against the COMMITTED REFERENCE FIXTURES: tests/golden/process_2q_pauli_fixed100.npz holds, for the
    first 64 items of this very batch, what the reference itself (tests/golden/make_goldens.py --fixed2q, run in the build
    container) produced -- the estimate after exactly 100 iterations, the estimate and iteration count at its own stopping point,
    per-iteration Dykstra counts.  Returns the deviation histogram of both modes (share of items <= 1e-9 / <= 1e-8, the
    maximum, the largest process-fidelity difference), or None when the fixture does not describe this batch."""
    fn = os.path.join(ROOT, "tests", "golden", f"process_2q_{batch.design.in_basis if hasattr(batch.design, 'in_basis') else 'pauli'}_fixed100.npz")
    if iters != 100 or not os.path.exists(fn):
        return None
    g = np.load(fn)
    n = int(g["expectations"].shape[0])
    if n > batch.B or not (np.array_equal(g["expectations"], batch.e[:n]) and np.array_equal(g["counts"], batch.c[:n])):
        return None
    _, _, so, om = _oracle()

    def fid(choi, b):
        return om.process_fidelity(so.kraus2pauli_liouville([g["unitaries"][b]]), so.choi2pauli_liouville(choi))

    out = []
    for label, (choi_g, st_g), want, last in (
            (f"fixed {iters} iterations (the timed launch)", gpu_fixed, g["pgdb_fixed"], np.full(n, iters)),
            ("converge (reference semantics, tomography.py:589)", gpu_conv, g["pgdb_conv"], g["conv_iter"])):
        dev = np.abs(choi_g[:n] - want).reshape(n, -1).max(axis=1)
        fdev = np.array([abs(fid(choi_g[b], b) - fid(want[b], b)) for b in range(n)])
        dyk_ref = np.array([int(g["dykstra"][b][:int(last[b])].sum()) for b in range(n)])
        out.append({"mode": label, "items": n, "against": "reference-generated fixtures " + os.path.relpath(fn, ROOT),
                    "share_le_1e-9": float((dev <= 1e-9).mean()), "share_le_1e-8": float((dev <= 1e-8).mean()),
                    "max_abs_choi_diff": float(dev.max()), "median_abs_choi_diff": float(np.median(dev)),
                    "max_process_fidelity_diff": float(fdev.max()),
                    "outer_iteration_mismatches": int((np.asarray(st_g["iterations"][:n]) != last).sum()),
                    "dykstra_total_mismatches": int((np.asarray(st_g["dykstra"][:n]) != dyk_ref).sum())})
    return out


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


def sweep_cpu_baseline(ks, ref, n_items):
    _, _, so, om = _oracle()
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


def pgdb3_cpu_baseline(design, e, c, iters, n_iters=10):
    """One 3-qubit item, `n_iters` outer iterations of the oracle (dense 8064 x 4096 design matrix, hoisted; ~2 s per
    iteration on one core), scaled to `iters` iterations: every iteration is one gradient (two dense mat-vecs), one
    Dykstra projection and a line search, so the per-iteration cost of the first ten is that of the run (the stalled
    iterations of a converged run are MORE expensive in the reference: ~50 cost evaluations each)."""
    od, oe, _, _ = _oracle()
    d = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    t0 = time.perf_counter()
    oe.pgdb_process_estimate(d, e[0], c[0], mode="fixed", max_iters=n_iters, A=A)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * iters / float(n_iters)), "unit": "reconstructions/s", "cores": 1, "kind": "port",
            "sample": f"item 0, {n_iters} outer iterations of the numpy oracle ({dt:.1f} s, design matrix hoisted), "
                      f"extrapolated linearly to {iters} iterations"}


# ================================================================================== workloads
def run_sweep(args, comm, _lib, synthetic, with_cpu, n=2):
    """BASELINE configs[2]: kraus2choi -> choi2pauli_liouville -> choi2chi + process_fidelity on
    `--sweep-items` random 2-qubit CPTP Kraus sets (K = 4) per GPU, generated ON THE DEVICE from the
    counter-based Philox stream keyed by the item id, all three representations written.
    n = 3: the d = 8 leg of the same pipeline (64 x 64 matrices, 65 536 items, fused sweep3_regs_kernel)."""
    K, D = 4, 4 ** n
    B = args.sweep_items if n == 2 else min(args.sweep_items, 65536)
    lib = _lib.lib()
    d_k = _lib.DeviceBuffer(B * K * D * 16)
    _lib.check(lib.fbx_random_kraus_dev(n, B, K, 17, comm.rank * B, d_k.ptr))
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.complex128)
    if n == 3:
        cnot = np.kron(cnot, np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2))
    ref = np.empty((1, D, D), dtype=np.complex128)
    _lib.check(lib.fbx_convert(_lib.REP_KRAUS, _lib.REP_PAULI_LIOUVILLE, n, 1,
                               _lib.dptr(np.ascontiguousarray(cnot[None, None]).view(np.float64)), 1,
                               _lib.dptr(ref.view(np.float64))))
    d_r = _lib.DeviceBuffer.from_array(ref)
    d_c, d_p, d_x = (_lib.DeviceBuffer(B * D * D * 16) for _ in range(3))
    d_f = _lib.DeviceBuffer(B * 8)

    def step():
        _lib.check(lib.fbx_kraus_sweep_dev(n, B, K, d_k.ptr, d_r.ptr, d_c.ptr, d_p.ptr, d_x.ptr, d_f.ptr))

    elapsed, kms = timed_steps(step, args.steps, args.warmup, comm, _lib)
    bytes_item = K * D * 16 + 3 * D * D * 16 + 8            # 13 320 B
    ksec = kms / 1e3 / args.steps
    gbs = B * bytes_item / ksec / 1e9
    fid = d_f.to_array(np.float64, (min(B, 4096),))
    line = {"tag": f"sweep_{n}q", "metric": f"conversion sweep items/sec ({n}-qubit Kraus -> Choi -> PTM -> chi + process_fidelity)",
            "value": comm.world * B * args.steps / elapsed, "unit": "items/s", "n_gpus": comm.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{B} random {n}-qubit CPTP Kraus sets (K=4) per GPU generated on the device "
                                   f"(Philox4x32-10 keyed by item id), all three representations + fidelity "
                                   f"written, inputs resident in HBM",
                       "items_per_gpu": B, "parallelism": f"shard{comm.world}",
                       "mean_fidelity_to_cnot": float(fid.mean())},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS,
                         "traffic": _profiled("sweep_kernel_hbm_bytes_per_launch" if n == 2 else "sweep3_kernel_hbm_bytes_per_launch"),
                         "kernel": "sweep2q_pair_kernel" if n == 2 else "sweep3_regs_kernel", "kernel_ms": 1e3 * ksec,
                         "note": f"achieved = {bytes_item} algorithmic bytes per item (K x {D * 16} in, 3 x {D * D * 16} + 8 out) / HIP-event kernel time",
                         "box": {"device": _lib.device_name()[0].strip(), "pci_bus_id": _lib.device_id()[1],
                                 "note": "an HBM-write-bound kernel: the boxes of the pool differ on it (2-qubit sweep 2.16-2.75 ms = "
                                         "4.8-6.2 TB/s over the boxes seen in rounds 3-4, a plain streaming fill 4.9-6.7 TB/s); "
                                         "kernel_ms / achieved of THIS line are this box's"}}}
    if with_cpu and comm.rank == 0:
        n_cpu = 2000 if n == 2 else 60
        ks = d_k.to_array(np.complex128, (n_cpu, K, 2 ** n, 2 ** n))
        line["cpu_baseline"] = sweep_cpu_baseline(ks, ref[0], n_cpu)
    for buf in (d_k, d_r, d_c, d_p, d_x, d_f):
        buf.free()
    return line


def run_pgdb3(args, comm, _lib, synthetic, with_cpu):
    """BASELINE configs[3]: 3-qubit (64 x 64 Choi) PGDB process tomography, batch 256 per GPU, SIC
    in-basis (4032 settings) unless --in-basis pauli (13 608), 100 fixed iterations."""
    B = 256
    basis = args.in_basis or "sic"
    design, _, e, c = synthetic.process_batch(3, basis, B, first_item=B * comm.rank)       # 256 DISTINCT experiments
    lib = _lib.lib()
    d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
    d_choi = _lib.DeviceBuffer(B * 64 * 64 * 16)
    d_it, d_dy, d_w = _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 16)

    def step():
        _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED, args.iters,
                                            d_choi.ptr, d_it.ptr, d_dy.ptr, None, None, d_w.ptr))

    steps = max(1, min(args.steps, 5))
    elapsed, kms = timed_steps(step, steps, min(args.warmup, 1), comm, _lib)
    dyk = d_dy.to_array(np.int32, (B,)); its = d_it.to_array(np.int32, (B,)); work = d_w.to_array(np.int32, (B, 4))
    ksec = kms / 1e3 / steps
    m = design.m
    # algorithmic flops in the reference's dense formulation (SURVEY 8d recipe at n = 3): per outer
    # iteration 3 R D^2 complex MACs (gradient 2, cost 1; R = 2 m rows, D^2 = 4096) + per Dykstra
    # iteration one 64 x 64 Hermitian eigendecomposition (~25 N^3) + V L V^H (2 N^3 complex MACs)
    flop = args.iters * 3 * (2 * m) * 4096 * 8 + float(dyk.mean()) * (25 * 64 ** 3 + 2 * 64 ** 3 * 8)
    tflops = B * flop / ksec / 1e12
    ex, parts = pgdb3q_executed_flop(m, design.n_states, its, dyk, work)
    meas = _measured_flop("pgdb3_kernel<4>" if basis == "sic" else "pgdb3_kernel<14>", B)
    mfma_meas = _measured_mfma_flop("pgdb3_kernel<4>" if basis == "sic" else "pgdb3_kernel<14>", B)
    line = {"tag": f"pgdb_3q_{basis}", "metric": "process-tomography MLE reconstructions/sec (3-qubit, 64x64 Choi, 100 iters)",
            "value": comm.world * B * steps / elapsed, "unit": "reconstructions/s", "n_gpus": comm.world,
            "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{B} independent 3-qubit process tomographies per GPU, {basis} in-basis "
                                   f"({m} settings, 1000 shots), {args.iters} fixed PGDB iterations, inputs "
                                   f"resident in HBM ({B} distinct experiments: items {B * comm.rank}..{B * comm.rank + B - 1} of the "
                                   f"SURVEY 8d recipe)", "batch_per_gpu": B,
                       "iters": args.iters, "parallelism": f"shard{comm.world}",
                       "mean_dykstra_iters": float(dyk.mean()), "mean_jacobi_sweeps": float(work[:, 0].mean()),
                       "max_over_mean_jacobi_sweeps": float(work[:, 0].max() / work[:, 0].mean())},
            "roofline": {"bound": "mfma", "pipe": "fp64 VALU + fp64 MFMA (basis changes, table products) + LDS",
                         "achieved": B * ex / ksec / 1e12, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": B * ex / ksec / 1e12 / FP64_PEAK_TFLOPS,
                         "executed_flop_measured": meas,
                         "measured_frac": (B * meas / ksec / 1e12 / FP64_PEAK_TFLOPS) if meas else None,
                         "mfma_frac": B * (mfma_meas if mfma_meas else parts["basis_change_mfma"] + parts["tables_mfma"]) / ksec / 1e12 / FP64_PEAK_TFLOPS,
                         "mfma_frac_source": "hardware count (profiles/pmc_flops.json)" if mfma_meas else "work-counter model",
                         "traffic": _profiled("pgdb3_kernel_hbm_bytes_per_launch" if basis == "sic" else "pgdb3pauli_kernel_hbm_bytes_per_launch"),
                         "kernel": "pgdb3_kernel", "kernel_ms": 1e3 * ksec,
                         "executed_flop": ex, "executed_breakdown": {k: round(v) for k, v in parts.items()},
                         "dense_accounting_tflops": tflops, "dense_accounting_frac": tflops / FP64_PEAK_TFLOPS,
                         "note": "achieved / frac = flops the Kronecker-form kernel really performs per reconstruction, from its "
                                 "work counters (DESIGN.md 4.4), x batch / HIP-event kernel time; executed_flop_measured = the "
                                 "hardware's count of the same launch (profiles/pmc_flops.json).  dense_accounting_tflops = the "
                                 "same launch priced in the reference's dense formulation (3 x 2m x 4096 complex MACs per outer "
                                 "iteration + ~25 N^3 per 64 x 64 eigendecomposition): an accounting figure that exceeds the fp64 "
                                 "peak for the Pauli in-basis, because the dense A products it counts are never executed"}}
    if with_cpu and comm.rank == 0 and basis == "sic":
        line["cpu_baseline"] = pgdb3_cpu_baseline(design, e, c, args.iters)
    for buf in (d_e, d_c, d_choi, d_it, d_dy, d_w):
        buf.free()
    return line


def run_pgdb1(args, comm, _lib, synthetic, with_cpu):
    """Single-qubit process tomography to convergence (what the reference's own tests and notebook run,
    tests/test_process_tomography.py:72-112), 2^20 experiments per GPU, one reconstruction per lane (csrc/fbx_pgdb1.hip; at
    this size one launch per outer iteration, the reconstructions re-binned by Dykstra count in between): 16 384 distinct
    experiments of the SURVEY 8d recipe, each 64 times."""
    distinct, reps = 16384, 64
    B = distinct * reps
    design, _, e0, c0 = synthetic.process_batch(1, "pauli", distinct, first_item=distinct * comm.rank)
    lib = _lib.lib()
    d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (reps, 1))), _lib.DeviceBuffer.from_array(np.tile(c0, (reps, 1)))
    d_choi = _lib.DeviceBuffer(B * 16 * 16)
    d_it, d_dy, d_w = _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 16)

    def step():
        _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_CONVERGE, 0,
                                            d_choi.ptr, d_it.ptr, d_dy.ptr, None, None, d_w.ptr))

    steps = max(1, min(args.steps, 5))
    elapsed, kms = timed_steps(step, steps, min(args.warmup, 1), comm, _lib)
    its = d_it.to_array(np.int32, (B,))[:distinct]; dyk = d_dy.to_array(np.int32, (B,))[:distinct]
    work = d_w.to_array(np.int32, (B, 4))[:distinct]
    ksec = kms / 1e3 / steps
    m, S = design.m, design.n_states
    # flops a lane executes (fma = 2), from the work counters: a 4 x 4 Jacobi sweep = 6 rotations x (36 + 2 x 24 + 4 x 24) ~ 1080,
    # a warm basis change 2 x 64 + 40 complex MACs ~ 1350, V L V^H + Dykstra bookkeeping ~ 700 per Dykstra iteration; a full
    # cost evaluation 2 m x 45 + 32 S; a gradient 2 m x 30 + 64 S; a power-sum pass 2 m x 40; three Pauli transforms of 130
    ex = float(np.mean(work[:, 0] * 1080.0 + dyk * (1350.0 + 700.0) + work[:, 2] * (90.0 * m + 32.0 * S)
                       + its * (60.0 * m + 64.0 * S + 390.0) + work[:, 3] * 80.0 * m))
    line = {"tag": "pgdb_1q", "metric": "process-tomography MLE reconstructions/sec (1-qubit, to convergence)",
            "value": comm.world * B * steps / elapsed, "unit": "reconstructions/s", "n_gpus": comm.world,
            "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} single-qubit process tomographies per GPU ({distinct} distinct experiments x {reps}), pauli "
                                   f"in-basis ({m} settings, 1000 shots), PGDB to convergence (the reference's stopping rule), "
                                   "inputs resident in HBM", "batch_per_gpu": B, "parallelism": f"shard{comm.world}",
                       "mean_outer_iters": float(its.mean()), "max_outer_iters": int(its.max()),
                       "mean_dykstra_iters": float(dyk.mean()), "mean_jacobi_sweeps": float(work[:, 0].mean())},
            "roofline": {"bound": "mfma", "pipe": "fp64 VALU (one reconstruction per lane)", "achieved": B * ex / ksec / 1e12,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": B * ex / ksec / 1e12 / FP64_PEAK_TFLOPS,
                         "traffic": _profiled("pgdb1_kernel_hbm_bytes_per_launch") if B == 1 << 20 else None,
                         "kernel": "pgdb1_step_kernel (all launches of one call: one per outer iteration)",
                         "kernel_ms": 1e3 * ksec, "executed_flop": ex,
                         "executed_flop_measured": _measured_flop("pgdb1_step_kernel", B),
                         "note": "flops executed per reconstruction (work counters x per-unit counts from the source) x batch / "
                                 "HIP-event time of the call; lanes idle through divergence (a wavefront runs the longest Dykstra / "
                                 "line-search trip count of its 64 lanes) are not counted as work; executed_flop_measured = the "
                                 "hardware's count summed over the call's launches (idle lanes included)"}}
    if with_cpu and comm.rank == 0:
        od, oe, _, _ = _oracle()
        d = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
        A = oe.design_matrix_A(d)
        n, t0 = 0, time.perf_counter()
        while n < 256 and time.perf_counter() - t0 < max(2.0, args.cpu_sample / 4):
            oe.pgdb_process_estimate(d, e0[n], c0[n], A=A)
            n += 1
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": n / dt, "unit": "reconstructions/s", "cores": 1, "kind": "port",
                                "sample": f"first {n} experiments to convergence, numpy oracle with the design matrix hoisted, {dt:.1f} s"}
    for buf in (d_e, d_c, d_choi, d_it, d_dy, d_w):
        buf.free()
    return line


def mle_state_executed_flop(n, m, updates):
    """fp64 operations one state reconstruction executes per the source (csrc/fbx_state.hip, fma = 2, a division ~ 10): per
    update Pauli expectations D x d x 2, per setting two guarded ratios + the weights ~ 40, synthesis D x d, two d x d complex
    products 2 x 8 d^3, the complex division by the trace ~ 30 per entry, the step norm 6 per entry."""
    d = 2 ** n
    D = d * d
    return updates * (2 * D * d + 40 * m + D * d + 16 * d ** 3 + 36 * D)


def run_mle_state(args, comm, _lib, synthetic, with_cpu, n=2):
    """The state-estimator half of north_star: iterative_mle_state_estimate (tomography.py:168-270, _R :273-338) with the
    reference's defaults (epsilon 0.1, tol 1e-9) and maxiter = 100 -- 99 updates, the configuration of BASELINE.md section 2's
    CPU probes (80 ms / 0.45 s per 2- / 3-qubit state) -- on a large resident batch: 4096 distinct experiments of the SURVEY 8d
    recipe (Haar pure state mixed 5 % with I/d, 1000 shots per setting), tiled."""
    distinct = 4096
    B = {1: 1 << 21, 2: 1 << 20, 3: 1 << 18}[n]
    design, _, e0, c0 = synthetic.state_batch(n, distinct, first_item=distinct * comm.rank, mixed=0.05)
    reps = B // distinct
    lib = _lib.lib()
    d = 2 ** n
    d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (reps, 1))), _lib.DeviceBuffer.from_array(np.tile(c0, (reps, 1)))
    d_rho, d_it, d_hit = _lib.DeviceBuffer(B * d * d * 16), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4)
    maxiter = args.iters

    def step():
        _lib.check(lib.fbx_mle_state_dev(design.handle, B, d_e.ptr, d_c.ptr, 0.1, 0.0, 0.0, 1e-9, maxiter,
                                         d_rho.ptr, d_it.ptr, d_hit.ptr))

    steps = max(1, min(args.steps, 10))
    elapsed, kms = timed_steps(step, steps, min(args.warmup, 2), comm, _lib)
    its = d_it.to_array(np.int32, (B,))[:distinct]
    rho = d_rho.to_array(np.complex128, (distinct, d, d))
    ksec = kms / 1e3 / steps
    m = design.m
    # the iteration counter starts at 1 and the cap check precedes the update (tomography.py:241-246): min(its, maxiter - 1) updates
    ex = mle_state_executed_flop(n, m, float(np.mean(np.minimum(its, maxiter - 1))))
    kernel = {1: "mle_state_packed_kernel<1>", 2: "mle_state_packed_kernel<2>", 3: "mle_state_plain3_kernel"}[n]
    meas = _measured_flop(kernel, B)
    algo_bytes = 2 * m * 8 + d * d * 16 + 8
    line = {"tag": f"mle_state_{n}q",
            "metric": f"state-tomography iterative-MLE reconstructions/sec ({n}-qubit, maxiter {maxiter})",
            "value": comm.world * B * steps / elapsed, "unit": "reconstructions/s", "n_gpus": comm.world,
            "steps": steps, "warmup": min(args.warmup, 2), "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} {n}-qubit state tomographies per GPU ({distinct} distinct experiments x {reps}: Haar pure "
                                   f"state mixed 5 % with I/d, {m} Pauli settings, 1000 shots), iterative_mle_state_estimate defaults "
                                   f"(epsilon 0.1, tol 1e-9) with maxiter {maxiter} = {maxiter - 1} updates, inputs resident in HBM",
                       "batch_per_gpu": B, "iters": maxiter, "parallelism": f"shard{comm.world}",
                       "mean_outer_iters": float(its.mean()), "max_outer_iters": int(its.max()),
                       "max_trace_error": float(np.abs(np.trace(rho, axis1=1, axis2=2) - 1).max())},
            "roofline": {"bound": "mfma", "pipe": "fp64 VALU through LDS round trips (no MFMA: d <= 8)",
                         "achieved": B * ex / ksec / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": B * ex / ksec / 1e12 / FP64_PEAK_TFLOPS,
                         "executed_flop": ex, "executed_flop_measured": meas,
                         "measured_frac": (B * meas / ksec / 1e12 / FP64_PEAK_TFLOPS) if meas else None, "mfma_frac": 0.0,
                         "traffic": _profiled(f"mle_state{n}_kernel_hbm_bytes_per_launch"),
                         "kernel": kernel, "kernel_ms": 1e3 * ksec,
                         "hbm": {"achieved": B * algo_bytes / ksec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": B * algo_bytes / ksec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": B * algo_bytes},
                         "note": "executed flops per reconstruction from the source (fma = 2, division ~ 10) x batch / HIP-event kernel "
                                 "time; a dependent chain of ~8 LDS round trips per update bounds it, not the fp64 pipe (DESIGN.md 4.7)"}}
    if with_cpu and comm.rank == 0:
        od, oe, _, _ = _oracle()
        dd = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
        import warnings
        k, t0, dev = 0, time.perf_counter(), 0.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            while k < 64 and time.perf_counter() - t0 < max(3.0, args.cpu_sample / 3):
                want = oe.iterative_mle_state_estimate(dd, e0[k], c0[k], maxiter=maxiter)
                dev = max(dev, float(np.abs(want - rho[k]).max()))
                k += 1
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": k / dt, "unit": "reconstructions/s", "cores": 1, "kind": "port",
                                "sample": f"first {k} experiments, numpy oracle, maxiter {maxiter}, {dt:.1f} s"}
        line["config"]["max_abs_diff_vs_oracle"] = dev
    for buf in (d_e, d_c, d_rho, d_it, d_hit):
        buf.free()
    return line


def run_shots(args, comm, _lib, synthetic, with_cpu):
    """SURVEY 8 row f2, the caller side of the path: shots_to_obs_moments (observable_estimation.py:804-853) for the shots of one
    configs[1]-sized acquisition -- 1024 two-qubit process tomographies x 540 settings x 1000 shots, [n_shots][n_qubits] 0/1 bytes per
    setting as qc.run returns them -- reduced on the device to the expectations and variances the estimators take.  Bit patterns come
    from a fixed 64 Ki-setting block (generated once, tiled): the kernel's work does not depend on the values."""
    n, shots = 2, 1000
    S = 1024 * 540
    block = 1 << 16
    rs = np.random.RandomState(11 + comm.rank)
    bits0 = rs.randint(0, 2, size=(block, shots, n)).astype(np.uint8)
    mask0 = rs.randint(0, 2, size=(block, n)).astype(np.uint8); mask0[:, 0] |= (mask0.sum(1) == 0)
    reps = -(-S // block)
    bits = np.tile(bits0, (reps, 1, 1))[:S]; mask = np.tile(mask0, (reps, 1))[:S]
    lib = _lib.lib()
    d_bits, d_mask = _lib.DeviceBuffer.from_array(bits), _lib.DeviceBuffer.from_array(mask)
    d_mean, d_var = _lib.DeviceBuffer(S * 8), _lib.DeviceBuffer(S * 8)

    def step():
        _lib.check(lib.fbx_shots_to_moments_dev(n, S, shots, d_bits.ptr, d_mask.ptr, None, 0, d_mean.ptr, d_var.ptr))

    elapsed, kms = timed_steps(step, args.steps, args.warmup, comm, _lib)
    ksec = kms / 1e3 / args.steps
    mean = d_mean.to_array(np.float64, (S,)); var = d_var.to_array(np.float64, (S,))
    bytes_setting = shots * n + n + 16
    gbs = S * bytes_setting / ksec / 1e9
    line = {"tag": "shots_2q", "metric": "shots -> observable moments, settings/sec (2-qubit, 1000 shots per setting)",
            "value": comm.world * S * args.steps / elapsed, "unit": "settings/s", "n_gpus": comm.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{S} settings (1024 two-qubit process tomographies x 540) x {shots} shots x {n} qubits of 0/1 bytes per GPU, "
                                   "resident in HBM: +-1 products under the observable's mask, mean and variance of the mean per setting",
                       "items_per_gpu": S, "parallelism": f"shard{comm.world}"},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": _profiled("shots_kernel_hbm_bytes_per_launch"), "kernel": "shots_pipe_kernel<2,2>",
                         "kernel_ms": 1e3 * ksec,
                         "note": f"achieved = {bytes_setting} algorithmic bytes per setting ({shots} x {n} shot bytes + mask in, mean + variance out) / HIP-event kernel time"}}
    if with_cpu and comm.rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from fbx_oracle import acquisition as oa
        k, t0, exact = 0, time.perf_counter(), True
        while k < 4096 and time.perf_counter() - t0 < 3.0:
            m_, v_ = oa.shots_to_obs_moments(bits[k], mask[k])
            exact = exact and m_ == mean[k] and abs(v_ - var[k]) <= 1e-18 + 1e-15 * abs(v_)
            k += 1
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": k / dt, "unit": "settings/s", "cores": 1, "kind": "port",
                                "sample": f"first {k} settings, numpy oracle (observable_estimation.py:804-853), {dt:.1f} s"}
        line["config"]["matches_oracle_on_sample"] = bool(exact)
    for buf in (d_bits, d_mask, d_mean, d_var):
        buf.free()
    return line


class PgdbBatch:
    """A resident batch of 2-qubit process tomographies and its output buffers."""

    def __init__(self, _lib, synthetic, in_basis, B, first_item, max_distinct=1 << 20):
        # every item distinct (its own Haar unitary and shot noise, seeds first_item + b): generated on the host in
        # blocks of 8192 (~3 s each) so that the generator's temporaries stay small
        n_distinct = min(B, max_distinct)
        parts, self.us = [], None
        for b0 in range(0, n_distinct, 8192):
            self.design, us, e, c = synthetic.process_batch(2, in_basis, min(8192, n_distinct - b0), first_item=first_item + b0)
            parts.append((e, c))
            if self.us is None:
                self.us = us
        e = np.concatenate([p[0] for p in parts]); c = np.concatenate([p[1] for p in parts])
        if n_distinct < B:
            reps = -(-B // n_distinct)
            e = np.tile(e, (reps, 1))[:B]; c = np.tile(c, (reps, 1))[:B]
        self.e, self.c, self.B, self.n_distinct, self._lib = e, c, B, n_distinct, _lib
        DB = _lib.DeviceBuffer
        self.d_e, self.d_c = DB.from_array(e), DB.from_array(c)
        self.d_choi, self.d_it, self.d_dy = DB(B * 256 * 16), DB(B * 4), DB(B * 4)
        self.d_bt, self.d_cost, self.d_work = DB(B * 4), DB(B * 8), DB(B * 16)

    def launch(self, mode, iters):
        L = self._lib
        L.check(L.lib().fbx_pgdb_process_dev(self.design.handle, self.B, self.d_e.ptr, self.d_c.ptr, 1, mode, iters,
                                             self.d_choi.ptr, self.d_it.ptr, self.d_dy.ptr, self.d_bt.ptr,
                                             self.d_cost.ptr, self.d_work.ptr))

    def stats(self):
        B = self.B
        return {"iterations": self.d_it.to_array(np.int32, (B,)), "dykstra": self.d_dy.to_array(np.int32, (B,)),
                "backtracks": self.d_bt.to_array(np.int32, (B,)), "work": self.d_work.to_array(np.int32, (B, 4))}

    def choi(self, n):
        return self.d_choi.to_array(np.complex128, (n, 16, 16))

    def free(self):
        for b in (self.d_e, self.d_c, self.d_choi, self.d_it, self.d_dy, self.d_bt, self.d_cost, self.d_work):
            b.free()


def pgdb_roofline(batch, st, kernel_s, iters):
    """`achieved` / `frac`: the flops the kernel EXECUTES (per-item work counters x per-unit counts from the source; next to
    it the hardware's own count of the same launch when a PMC pass is on file) / HIP-event kernel time, against the fp64 peak.
    The reference's dense-A pricing of SURVEY.md 8d is kept as `dense_accounting_*`: it counts A-products the Kronecker-form
    kernel never performs and exceeds the peak at large batches, so it is not a utilisation figure."""
    B, m, S = batch.B, batch.design.m, batch.design.n_states
    dense = B * ALGO_FLOP_PER_RECON * (iters / 100.0) / kernel_s / 1e12
    # which kernel ran (csrc/fbx_pgdb.hip launch_pgdb): the two-waves-per-SIMD body from 1025 items on when the design's Bloch table fits
    # beside the Pauli coefficients (at most 50 input states), else the one-wave body -- whose eigensolver has two workers per block
    lean = B > 1024 and S <= 50
    ex, parts = pgdb2q_executed_flop(m, S, st["iterations"], st["dykstra"], st["backtracks"], st["work"], two_workers=not lean)
    algo_bytes = 2 * m * 8 + 4096
    kernel = ("pgdb_lean_pieces_kernel" if lean else "pgdb_pieces_kernel" if B > 1024 else "pgdb_kernel") + \
             ("<2,4>" if m <= 256 else "<2,9>" if m <= 576 else "<2,16>")
    measured = _measured_flop(kernel, B)
    mfma_measured = _measured_mfma_flop(kernel, B)
    out = {"bound": "mfma", "pipe": "fp64 VALU + MFMA", "achieved": B * ex / kernel_s / 1e12, "peak": FP64_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": B * ex / kernel_s / 1e12 / FP64_PEAK_TFLOPS,
           "traffic": _profiled({1024: "pgdb_kernel_hbm_bytes_per_launch", 8192: "pgdb_lean8192_hbm_bytes_per_launch",
                                 65536: "pgdb_lean65536_hbm_bytes_per_launch"}.get(B, "-")),
           "kernel": kernel, "kernel_ms": 1e3 * kernel_s,
           "executed_flop": ex, "executed_breakdown": {k: round(v) for k, v in parts.items()},
           "executed_flop_measured": measured,
           "measured_frac": (B * measured / kernel_s / 1e12 / FP64_PEAK_TFLOPS) if measured else None,
           # MFMA utilisation AS a utilisation: matrix-core flops / the dense fp64 MFMA peak / kernel time (<= 1 by construction)
           "mfma_frac": B * (mfma_measured if mfma_measured else parts["basis_change_mfma"]) / kernel_s / 1e12 / FP64_PEAK_TFLOPS,
           "mfma_frac_source": "hardware count (profiles/pmc_flops.json)" if mfma_measured else "work-counter model",
           "note_short": "executed fp64 flops (work counters x per-unit source counts; measured_frac = hardware count) / kernel time / 78.6 TFLOP/s; mfma_frac = matrix-core share",
           "dense_accounting_tflops": dense, "dense_accounting_frac": dense / FP64_PEAK_TFLOPS,
           "note": "PGDB is fp64-compute / latency bound (SURVEY.md 8d).  achieved / frac = flops the kernel really executes "
                   "per reconstruction (work counters: Jacobi sweeps, eigenvalue terms, cost evaluations; x per-unit counts "
                   "from the source, fma = 2) x batch / HIP-event kernel time / 78.6 TFLOP/s.  executed_flop_measured = the "
                   "same launch counted by SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes + SQ_INSTS_VALU_MFMA_MOPS_F64 "
                   "(profiles/pmc_flops.json; EXEC-masked lanes included, so an upper bound).  dense_accounting_* = SURVEY "
                   "8d's agreed 0.77 GFLOP per reconstruction in the reference's dense-A formulation: an accounting "
                   "figure (86 % of it is A-GEMV work the Kronecker-form kernel never performs; it exceeds 1 at 65 536 items)",
           "hbm": {"achieved": B * algo_bytes / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": B * algo_bytes / kernel_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": B * algo_bytes}}
    return out


def run_pgdb(args, comm, _lib, synthetic, rank_info):
    """The headline.  N = 1: configs[1] (1024 items).  N > 1: configs[4] (total batch block-partitioned)
    plus the 1024-per-GPU weak figure."""
    from fbx.parallel import shard_bounds
    world, rank = comm.world, comm.rank
    total = args.total_batch
    if total < 0:
        total = 65536 if world > 1 else 0
    extras = {}
    if total > 0:
        lo, hi = shard_bounds(total, rank, world)
        batch = PgdbBatch(_lib, synthetic, args.in_basis, hi - lo, lo)
        scaling, shards = "strong", f"contiguous blocks of a {total}-item batch, distinct seeds (items {lo}..{hi - 1} on this rank, all distinct)"
    else:
        batch = PgdbBatch(_lib, synthetic, args.in_basis, args.batch, rank * args.batch)
        scaling, shards = "weak", "distinct seeds per rank"
    elapsed, kms = timed_steps(lambda: batch.launch(_lib.MODE_FIXED, args.iters), args.steps, args.warmup, comm, _lib)
    st = batch.stats()
    total_recons = (total if total > 0 else world * batch.B) * args.steps
    kernel_s = kms / 1e3 / args.steps
    # whole-job summary over the ranks (the one collective of the path besides the optional all-gather)
    sums = comm.allreduce([float(st["iterations"].sum()), float(st["dykstra"].sum()), float(st["backtracks"].sum()),
                           float(st["work"][:, 0].sum()), float(batch.B)], "sum")
    line = {
        "metric": "process-tomography MLE reconstructions/sec (2-qubit, 100 iters)",
        "value": total_recons / elapsed, "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{total} independent 2-qubit process tomographies sharded over {world} GPU(s) "
                                f"(BASELINE configs[4])" if total > 0 else
                                f"{batch.B} independent 2-qubit process tomographies per GPU (BASELINE configs[1])")
                               + f", {args.in_basis} in-basis ({batch.design.m} settings, 1000 shots), {args.iters} "
                                 f"fixed PGDB iterations, inputs resident in HBM",
                   "batch_per_gpu": batch.B, "total_batch": total if total > 0 else world * batch.B,
                   "iters": args.iters, "parallelism": f"shard{world}", "shards": shards,
                   "device": rank_info["device"], "compute_units": rank_info["cus"],
                   "collectives": rank_info["transport"],
                   "mean_outer_iters": sums[0] / sums[4], "mean_dykstra_iters": sums[1] / sums[4],
                   "mean_backtracks": sums[2] / sums[4], "mean_jacobi_sweeps": sums[3] / sums[4]},
        "roofline": pgdb_roofline(batch, st, kernel_s, args.iters),
    }
    if world > 1 and total > 0:
        # the N = 1-comparable figure: configs[1]'s 1024 items on every rank (weak scaling)
        batch.free()
        wb = PgdbBatch(_lib, synthetic, args.in_basis, args.batch, rank * args.batch)
        el, km = timed_steps(lambda: wb.launch(_lib.MODE_FIXED, args.iters), args.steps, args.warmup, comm, _lib)
        line["per_gpu_1024"] = {"value": world * wb.B * args.steps / el, "unit": "reconstructions/s", "scaling": "weak",
                                "ms_per_step": 1e3 * el / args.steps, "kernel_ms": km / args.steps,
                                "workload": f"{wb.B} items on every rank (BASELINE configs[1] per GPU), distinct seeds per rank"}
        wb.free()
        return line, None
    return line, batch


def _pipeline_stages(n, chunk):
    """Launches of the pipelined host-pointer call (csrc/fbx_pgdb.hip): a small first stage, the bulk in pieces of <= 65 536
    items on a high-priority stream, a small last stage."""
    if n <= chunk:
        return 1
    first = min(chunk, n // 2)
    last = chunk if n >= 4 * chunk else 0
    return 1 + -(-(n - first - last) // 65536) + (1 if last else 0)


def strong_anchor(args, comm, _lib, synthetic):
    """N = 1 only: BASELINE configs[4]'s whole batch (65 536 distinct items, the very items the ranks of an N-GPU run
    own between them) on ONE GPU, so that the driver's 1 -> 2 -> 4 -> 8 curve has a same-workload N = 1 point next to
    the 1024-item headline (which runs the one-wave kernel; this batch, like every rank's share of it, runs the
    two-waves-per-SIMD kernel in launches of 8192)."""
    total = args.anchor_items
    t0 = time.perf_counter()
    batch = PgdbBatch(_lib, synthetic, args.in_basis, total, 0)
    t_gen = time.perf_counter() - t0
    steps = max(1, min(args.steps, 3))
    elapsed, kms = timed_steps(lambda: batch.launch(_lib.MODE_FIXED, args.iters), steps, 1, comm, _lib)
    st = batch.stats()
    out = {"value": total * steps / elapsed, "unit": "reconstructions/s", "n_gpus": 1, "scaling": "strong",
           "steps": steps, "warmup": 1, "ms_per_step": 1e3 * elapsed / steps, "kernel_ms": kms / steps,
           "workload": f"{total} independent 2-qubit process tomographies (BASELINE configs[4], items 0..{total - 1}, all "
                       f"distinct) on one GPU, {args.in_basis} in-basis, {args.iters} fixed PGDB iterations, inputs resident in HBM",
           "kernel": "pgdb_lean_pieces_kernel<2,9>" if batch.design.m > 256 else "pgdb_lean_pieces_kernel<2,4>",
           "mean_outer_iters": float(st["iterations"].mean()), "mean_dykstra_iters": float(st["dykstra"].mean()),
           "mean_jacobi_sweeps": float(st["work"][:, 0].mean()), "host_input_generation_s": t_gen,
           "roofline": pgdb_roofline(batch, st, kms / 1e3 / steps, args.iters),
           "note": "compare bench.py --gpus N (value = the same 65 536 items block-partitioned over N ranks) with THIS "
                   "figure, not with the 1024-item headline"}
    # ---- the same with the transfers inside (SURVEY 8d's metric): page-locked host buffers, pipelined in stages
    from fbx import tomography
    incl = {}
    pe, pc_ = _lib.pinned_copy(batch.e), _lib.pinned_copy(batch.c)
    pout = _lib.pinned_empty((total, 16, 16), np.complex128)
    for nb, calls in ((8192, 10), (total, 4)):
        if nb > total:
            continue
        # resident reference for this batch size: the same items, inputs in HBM
        L = _lib
        def resident():
            L.check(L.lib().fbx_pgdb_process_dev(batch.design.handle, nb, batch.d_e.ptr, batch.d_c.ptr, 1, L.MODE_FIXED, args.iters,
                                                 batch.d_choi.ptr, batch.d_it.ptr, batch.d_dy.ptr, batch.d_bt.ptr,
                                                 batch.d_cost.ptr, batch.d_work.ptr))
        el, _ = timed_steps(resident, 3, 1, comm, _lib)
        res_ms = 1e3 * el / 3
        ts = []
        for k in range(calls + 1):
            t0 = time.perf_counter()
            tomography.pgdb_process_estimate_batch(batch.design, pe[:nb], pc_[:nb], mode="fixed", max_iters=args.iters, out=pout[:nb])
            ts.append(time.perf_counter() - t0)
        th = float(np.median(ts[1:]))
        incl[str(nb)] = {"value": nb / th, "unit": "reconstructions/s", "ms_per_call": 1e3 * th, "resident_ms": res_ms,
                         "fraction_of_resident": res_ms / (1e3 * th), "calls": calls,
                         "stages": _pipeline_stages(nb, int(_lib.get_option("pgdb_host_chunk")))}
    out["pcie_inclusive"] = incl
    out["pcie_inclusive_note"] = ("fbx_pgdb_process on page-locked host buffers: a small first stage (fbx_set_option('pgdb_host_chunk') "
                                  "items) whose kernel covers the upload of the rest, the bulk in one launch on a high-priority stream, a "
                                  "small last stage that is still computing while the bulk's results go down; median of the calls after "
                                  "one warm-up, against the HBM-resident launch of the same items")
    del pe, pc_, pout
    batch.free()
    return out


def single_gpu_extras(args, comm, _lib, synthetic, batch, line):
    """N = 1 only, outside the headline's timed region: converge mode, PCIe-inclusive rate, latency of one
    experiment through the reference-signature call, parity self-check + CPU baselines."""
    from fbx import tomography
    B, iters = batch.B, args.iters
    n_cpu = min(args.cpu_sample, B)
    n_fix = min(64, B)
    gpu_fixed = (batch.choi(n_fix), batch.stats())
    # ---- converge mode: the reference's own semantics (stop at delta cost < 1e-10)
    el, km = timed_steps(lambda: batch.launch(_lib.MODE_CONVERGE, 0), args.steps, 1, comm, _lib)
    stc = batch.stats()
    gpu_conv = (batch.choi(n_fix), stc)
    line["converge_mode"] = {"value": B * args.steps / el, "unit": "reconstructions/s",
                             "ms_per_step": 1e3 * el / args.steps, "kernel_ms": km / args.steps,
                             "mean_outer_iters": float(stc["iterations"].mean()),
                             "max_outer_iters": int(stc["iterations"].max()),
                             "mean_dykstra_iters": float(stc["dykstra"].mean()),
                             "note": "FBX_MODE_CONVERGE: the reference loop (tomography.py:570-592); the mode the 1e-9 / "
                                     "1e-8 parity claim is made in"}
    # ---- host-pointer entry point: H2D of expectations + counts, kernel, D2H of the Choi matrices (SURVEY.md 8d's
    # definition of the metric).  Page-locked caller buffers (fbx_host_alloc): 8.8 MB in + 4.2 MB out at the PCIe rate.
    # At B = 1024 there is exactly one reconstruction per SIMD, so the batch is ONE pipeline stage and nothing overlaps
    # (a kernel reads its inputs in its first microseconds and writes its result in its last); the larger batches of
    # the strong-scaling leg below are pipelined.  The pageable-numpy figure is kept next to it.
    def host_calls(e_h, c_h, out_h, n):
        ts = []
        for k in range(n + 1):
            t0 = time.perf_counter()
            tomography.pgdb_process_estimate_batch(batch.design, e_h, c_h, mode="fixed", max_iters=iters, out=out_h)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[1:]))
    pe, pc_, pout = _lib.pinned_copy(batch.e), _lib.pinned_copy(batch.c), _lib.pinned_empty((B, 16, 16), np.complex128)
    th = host_calls(pe, pc_, pout, 10)
    tp = host_calls(batch.e, batch.c, None, 4)
    resident_ms = line["ms_per_step"]
    line["pcie_inclusive"] = {"value": B / th, "unit": "reconstructions/s", "ms_per_call": 1e3 * th,
                              "fraction_of_resident": resident_ms / (1e3 * th),
                              "pageable": {"value": B / tp, "ms_per_call": 1e3 * tp},
                              "note": "fbx_pgdb_process on page-locked host buffers (fbx_host_alloc): H2D of 8.8 MB, kernel, D2H of "
                                      "4.2 MB, host call overhead; median of 10 calls after one warm-up.  `value` of this line "
                                      "is the HBM-resident rate (the bench contract); this is SURVEY 8d's transfer-inclusive "
                                      "rate.  One launch at B = 1024 (nothing to overlap); see strong_65536.pcie_inclusive"}
    # SURVEY 8d defines the metric as this host-pointer call with the transfers inside; the bench contract asks for the
    # HBM-resident rate as `value`.  The driver's record keeps `config`: the transfer-inclusive figure goes there as well.
    line["config"]["transfer_inclusive"] = {"value": B / th, "unit": "reconstructions/s", "ms_per_call": 1e3 * th, "calls": 10,
                                            "fraction_of_resident": resident_ms / (1e3 * th),
                                            "what": "fbx_pgdb_process on page-locked host buffers: H2D 8.8 MB + kernel + D2H 4.2 MB, "
                                                    "median of 10 calls after one warm-up (SURVEY.md 8d's definition of the metric)"}
    del pe, pc_, pout
    # ---- one experiment at a time through the reference signature (List[ExperimentResult], qubits)
    from fbx.observable_estimation import ExperimentResult
    settings = tomography.generate_process_tomography_settings([0, 1], args.in_basis)
    res = [ExperimentResult(setting=s, expectation=float(x), std_err=0.0, total_counts=int(n))
           for s, x, n in zip(settings, batch.e[0], batch.c[0])]
    lat = []
    for k in range(6):
        t0 = time.perf_counter()
        tomography.pgdb_process_estimate(res, [0, 1])
        lat.append(time.perf_counter() - t0)
    line["single_experiment_latency_ms"] = {"value": 1e3 * float(np.median(lat[1:])),
                                            "note": "pgdb_process_estimate(results, qubits) -- flattening 540 result objects, "
                                                    "cached design, one-item launch to convergence, D2H; median of 5 after one warm-up"}
    if n_cpu:
        base, parity = cpu_baseline_and_parity(batch.design, batch.us, batch.e, batch.c, n_cpu, iters, gpu_fixed, gpu_conv)
        line["cpu_baseline"] = base
        line["parity_self_check"] = parity
        cores = os.cpu_count() or 1
        pool = cpu_baseline_pool(batch.design, batch.e, batch.c, iters, per_core=max(1, -(-n_cpu // min(cores, 64))))
        # the two other CPU figures of SURVEY 8d, as siblings of cpu_baseline and -- because the driver's record keeps `config`
        # but not unknown top-level keys -- once more, compactly, inside config
        line["cpu_baseline_reference_faithful"] = base["reference_faithful"]
        line["cpu_baseline_multicore"] = pool
        line["config"]["cpu_reference_faithful_1core"] = {k: base["reference_faithful"][k] for k in ("value", "unit", "sample")}
        line["config"]["cpu_multicore"] = {k: pool[k] for k in ("value", "unit", "cores", "sample")}
    fx = fixture_parity(batch, gpu_fixed, gpu_conv, iters)
    if fx is not None:
        line["parity_vs_reference_fixtures"] = fx
        line["config"]["parity_vs_reference_fixtures"] = [
            {k: f[k] for k in ("mode", "items", "share_le_1e-9", "share_le_1e-8", "max_abs_choi_diff", "max_process_fidelity_diff")} for f in fx]


# ================================================================================== output
# What the driver keeps of a run is the tail of stdout, and it parses the LAST line.  Round 5's single 20 KB line outgrew
# that; since round 6 the full record goes to gpurun_out/bench_detail.json, every secondary workload is printed as its own
# short line FIRST, and the last stdout line is the compact headline object (< 4 KB; tests/test_bench_line.py).
HEADLINE_MAX_BYTES = 4096
_ROOFLINE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "measured_frac", "mfma_frac", "traffic", "kernel", "kernel_ms")


def _sig(x, digits=6):
    """floats to `digits` significant digits, recursively (a bench line is read by people and size-capped)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if np.isfinite(x) else None
    if isinstance(x, (np.floating,)):
        return _sig(float(x), digits)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3].rstrip() + "..."


def compact_roofline(r):
    out = _pick(r, _ROOFLINE_KEEP)
    if r and "traffic" in r and "traffic" not in out:
        out["traffic"] = None
    return out


def compact_cpu(c, sample_chars=120):
    out = _pick(c, ("value", "unit", "cores", "kind"))
    if c and "sample" in c:
        out["sample"] = _short(c["sample"], sample_chars)
    return out


def compact_secondary(line):
    """one short stdout line per secondary workload: what it is, its rate, its roofline, its CPU baseline"""
    cfg = line.get("config", {})
    out = {"detail": line.get("tag", "secondary"), "metric": line["metric"], "value": line["value"], "unit": line["unit"],
           "n_gpus": line.get("n_gpus", 1), "steps": line.get("steps"), "warmup": line.get("warmup"),
           "ms_per_step": line.get("ms_per_step"), "dtype": line.get("dtype", "f64"),
           "config": {"workload": _short(cfg.get("workload", ""), 200),
                      **_pick(cfg, ("batch_per_gpu", "items_per_gpu", "iters", "mean_outer_iters", "mean_dykstra_iters",
                                    "mean_jacobi_sweeps"))},
           "roofline": compact_roofline(line.get("roofline"))}
    if "cpu_baseline" in line:
        out["cpu_baseline"] = compact_cpu(line["cpu_baseline"])
    return _sig(out)


def compact_headline(line):
    """The LAST stdout line: the bench contract's keys + `roofline` + `cpu_baseline`, with everything the driver's record should
    keep folded into `config` as bare numbers.  Always < HEADLINE_MAX_BYTES."""
    cfg = line.get("config", {})
    c = {"workload": _short(cfg.get("workload", ""), 260),
         **_pick(cfg, ("batch_per_gpu", "items_per_gpu", "total_batch", "iters", "parallelism", "device", "compute_units",
                       "mean_outer_iters", "mean_dykstra_iters", "mean_backtracks", "mean_jacobi_sweeps",
                       "mean_fidelity_to_cnot", "max_outer_iters"))}
    if isinstance(cfg.get("collectives"), dict):
        c["collectives"] = _pick(cfg["collectives"], ("backend", "ranks", "rccl_version"))
    if "transfer_inclusive" in cfg:
        c["transfer_inclusive"] = _pick(cfg["transfer_inclusive"], ("value", "ms_per_call", "fraction_of_resident"))
    if "converge_mode" in line:
        c["converge_mode"] = _pick(line["converge_mode"], ("value", "ms_per_step", "mean_outer_iters"))
    if "cpu_reference_faithful_1core" in cfg:
        c["cpu_reference_faithful_1core"] = _pick(cfg["cpu_reference_faithful_1core"], ("value",))
    if "cpu_multicore" in cfg:
        c["cpu_multicore"] = _pick(cfg["cpu_multicore"], ("value", "cores"))
    if "parity_vs_reference_fixtures" in cfg:
        c["parity_vs_reference_fixtures"] = [
            {"mode": "fixed100" if f["mode"].startswith("fixed") else "converge", "items": f["items"],
             "le_1e-9": f["share_le_1e-9"], "le_1e-8": f["share_le_1e-8"], "max_choi": f["max_abs_choi_diff"],
             "max_fidelity": f["max_process_fidelity_diff"]} for f in cfg["parity_vs_reference_fixtures"]]
    for key, val in line.items():
        if key.startswith("strong_") and isinstance(val, dict):
            c[key] = {**_pick(val, ("value", "ms_per_step", "kernel_ms")),
                      **{k: v for k, v in compact_roofline(val.get("roofline")).items() if k in ("frac", "measured_frac", "mfma_frac")}}
            c["multi_gpu_expectation"] = ("extrapolation: each of N ranks runs its 65536/N share at the one-GPU rate of that share "
                                          "(no data-path collective); no multi-GPU lease has run it")
    if "per_gpu_1024" in line:
        c["per_gpu_1024"] = _pick(line["per_gpu_1024"], ("value", "ms_per_step", "kernel_ms"))
    if "secondary" in line:
        c["secondary"] = {s.get("tag", _short(s["metric"], 60)): {"value": s["value"], **_pick(compact_roofline(s.get("roofline")), ("frac",))}
                          for s in line["secondary"]}
    c["detail"] = "earlier stdout lines + gpurun_out/bench_detail.json"
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in line}
    out["config"] = c
    out["roofline"] = compact_roofline(line.get("roofline"))
    if line.get("roofline", {}).get("note"):
        out["roofline"]["note"] = _short(line["roofline"]["note_short"] if "note_short" in line["roofline"] else line["roofline"]["note"], 160)
    if "cpu_baseline" in line:
        out["cpu_baseline"] = compact_cpu(line["cpu_baseline"], 160)
    out = _sig(out)
    # belt and braces: drop the optional parts, largest first, until the line fits
    for victim in ("secondary", "multi_gpu_expectation", "collectives", "cpu_multicore", "converge_mode"):
        if len(json.dumps(out)) < HEADLINE_MAX_BYTES:
            break
        out["config"].pop(victim, None)
    return out


def emit(line, stream=None, detail_out=None):
    """Full record -> gpurun_out/bench_detail.json; per-workload detail lines; the compact headline LAST."""
    stream = stream or sys.stdout
    detail_out = detail_out or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(detail_out), exist_ok=True)
        with open(detail_out, "w") as fh:
            json.dump(line, fh)
    except OSError:
        pass
    for sec in line.get("secondary", []):
        print(json.dumps(compact_secondary(sec)), file=stream)
    for key in ("parity_vs_reference_fixtures", "parity_self_check", "pcie_inclusive", "single_experiment_latency_ms"):
        if key in line:
            print(json.dumps(_sig({"detail": key, "data": line[key]})), file=stream)
    for key, val in line.items():
        if key.startswith("strong_") and isinstance(val, dict):
            d = {"detail": key, **_pick(val, ("value", "unit", "n_gpus", "scaling", "steps", "warmup", "ms_per_step", "kernel_ms",
                                              "mean_dykstra_iters", "mean_jacobi_sweeps")),
                 "workload": _short(val.get("workload", ""), 200), "roofline": compact_roofline(val.get("roofline")),
                 "pcie_inclusive": val.get("pcie_inclusive")}
            print(json.dumps(_sig(d)), file=stream)
    head = compact_headline(line)
    text = json.dumps(head)
    assert len(text) < HEADLINE_MAX_BYTES, len(text)
    print(text, file=stream, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.workload in ("pgdb", "all") and args.in_basis is None:
        args.in_basis = "pauli"

    from fbx import _lib, synthetic, parallel
    # Selects GPU LOCAL_RANK and fails loudly without one.  One GPU per rank: the ranks form an RCCL communicator or
    # the run exits non-zero -- the host-files barrier is a test convenience for ranks that SHARE a device and is only
    # reachable with --oversubscribe.
    comm, rdzv = parallel.init_from_env(allow_host_fallback=args.oversubscribe, allow_oversubscribe=args.oversubscribe)
    dev_name, cus = _lib.device_name()
    ordinal, pci = _lib.device_id()
    transport = {"backend": comm.backend}
    if comm.backend == "rccl":
        q = comm.query()                       # what the COMMUNICATOR reports (ncclCommCount / UserRank / CuDevice), not the environment
        transport.update(ranks=q["world"], rccl_version=comm.rccl_version)
        mine = np.zeros(40, dtype=np.uint8)
        mine[:4] = np.frombuffer(np.array([q["rank"], q["device"]], dtype=np.int16).tobytes(), dtype=np.uint8)
        mine[4:4 + len(pci)] = np.frombuffer(pci.encode(), dtype=np.uint8)
        rows = comm.allgather(mine)            # one all-gather over RCCL: every rank's own view of itself
        transport["rank_devices"] = [{"rank": int(r[:4].view(np.int16)[0]), "device": int(r[:4].view(np.int16)[1]),
                                      "pci_bus_id": bytes(r[4:]).rstrip(b"\0").decode()} for r in rows]
        if len({d["pci_bus_id"] for d in transport["rank_devices"]}) != comm.world:
            sys.exit(f"bench.py: ranks share a physical device: {transport['rank_devices']}")
    else:
        transport.update(ranks=comm.world, rank_devices=[{"rank": comm.rank, "device": ordinal, "pci_bus_id": pci}]
                         if comm.world == 1 else "ranks share devices (--oversubscribe)")
    if getattr(comm, "failure", None):
        transport["rccl_failure"] = comm.failure
    rank_info = {"device": dev_name.strip(), "cus": cus, "transport": transport}
    with_cpu = comm.world == 1 and args.cpu_sample > 0

    if args.workload == "sweep":
        line = run_sweep(args, comm, _lib, synthetic, with_cpu)
    elif args.workload == "sweep3":
        line = run_sweep(args, comm, _lib, synthetic, with_cpu, n=3)
    elif args.workload == "pgdb3":
        line = run_pgdb3(args, comm, _lib, synthetic, with_cpu)
    elif args.workload == "pgdb1":
        line = run_pgdb1(args, comm, _lib, synthetic, with_cpu)
    elif args.workload == "shots":
        line = run_shots(args, comm, _lib, synthetic, with_cpu)
    elif args.workload in ("mle_state", "mle_state3"):
        line = run_mle_state(args, comm, _lib, synthetic, with_cpu, n=2 if args.workload == "mle_state" else 3)
    else:
        secondary = []
        if args.workload == "all" and comm.world == 1:
            basis = args.in_basis
            secondary.append(run_sweep(args, comm, _lib, synthetic, with_cpu))
            secondary.append(run_sweep(args, comm, _lib, synthetic, with_cpu, n=3))
            args.in_basis = None
            secondary.append(run_pgdb3(args, comm, _lib, synthetic, with_cpu))
            args.in_basis = "pauli"                          # the stretch form of configs[3]: 13 608 settings
            secondary.append(run_pgdb3(args, comm, _lib, synthetic, False))
            secondary.append(run_pgdb1(args, comm, _lib, synthetic, with_cpu))
            secondary.append(run_mle_state(args, comm, _lib, synthetic, with_cpu, n=2))
            secondary.append(run_mle_state(args, comm, _lib, synthetic, with_cpu, n=3))
            secondary.append(run_shots(args, comm, _lib, synthetic, with_cpu))
            args.in_basis = basis
            _lib.release_workspace()
        line, batch = run_pgdb(args, comm, _lib, synthetic, rank_info)
        if batch is not None:
            if comm.world == 1 and args.workload == "all":
                single_gpu_extras(args, comm, _lib, synthetic, batch, line)
            batch.free()
            if comm.world == 1 and args.workload == "all" and args.anchor_items > 0:
                _lib.release_workspace()
                line["strong_65536" if args.anchor_items == 65536 else f"strong_{args.anchor_items}"] = \
                    strong_anchor(args, comm, _lib, synthetic)
        if secondary:
            line["secondary"] = secondary
    comm.barrier()
    comm.close()
    if rdzv is not None:
        rdzv.close()
    if comm.rank == 0:
        emit(line, detail_out=args.detail_out)


if __name__ == "__main__":
    main()
