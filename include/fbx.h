/*
 * fbx.h -- C ABI of libfbx.so, the MI355X (gfx950) tomography-reconstruction library.
 *
 * This is the drop-in boundary for the hot path of rigetti/forest-benchmarking
 * (forest/benchmarking/tomography.py:130-633, operator_tools/, distance_measures.py).
 * The reference has no FFI of its own (it is 100 % Python); each entry point below names
 * the reference function it replaces (file:line relative to forest/benchmarking/).
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - every entry point returns int: FBX_OK or an FBX_ERR_* category; the message is
 *    available per thread through fbx_last_error().  No exceptions / abort() cross the ABI.
 *  - threading: the library is re-entrant.  Every host thread that calls in owns its own HIP
 *    stream, timer events, staging-buffer pool and cached device workspaces (the 2-qubit PGDB
 *    kernel keeps 128 KiB of Dykstra bases per reconstruction of a launch -- 8 GiB for a 65 536-item
 *    batch, less when the device cannot give that much -- the 3-qubit one 768 MiB);
 *    fbx_release_workspace() gives the calling thread's cached device memory back.  The only
 *    process-wide state is the selected device (one process per GPU: fbx_set_device once,
 *    before other threads use the library) and the RCCL communicator (fbx_comm_*, one thread
 *    at a time).  "The library stream" below is the calling thread's stream.
 *  - all buffers are caller-owned, C-contiguous; complex128 is interleaved (re, im) doubles
 *    (binary compatible with `double _Complex` and numpy complex128); matrices are
 *    row-major [B][row][col].  Plain entry points take HOST pointers and do H2D/D2H
 *    themselves; *_dev entry points take DEVICE pointers (HBM-resident data) and are
 *    asynchronous on the library stream until fbx_synchronize().
 *  - column-stacking vec; un-normalised Choi on H_in (x) H_out; n-qubit Pauli order
 *    itertools.product('IXYZ', repeat=n) with qubits[0] the left-most tensor factor.
 *  - sizes: the estimators, projections, state measures and channel application take 1..3 qubits
 *    (fbx_kraus_sweep is one fused kernel for 1..2 and a composition of the pairwise conversions for 3);
 *    fbx_convert and fbx_process_fidelity 1..5 qubits; fbx_eigh / fbx_matmul any N <= 1024;
 *    fbx_convert_general and fbx_partial_trace any dimension (each entry point states its own range).
 *  - label codes: one-qubit input states 0:X+ 1:X- 2:Y+ 3:Y- 4:Z+ 5:Z- 6:SIC0 7:SIC1
 *    8:SIC2 9:SIC3; one-qubit Paulis 0:I 1:X 2:Y 3:Z.
 */
#ifndef FBX_H
#define FBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBX_OK               0
#define FBX_ERR_BAD_ARG      1   /* -> ValueError in the Python shim */
#define FBX_ERR_HIP          2   /* HIP runtime / kernel failure */
#define FBX_ERR_NO_DEVICE    3   /* no gfx950 device visible: the product fails loudly */
#define FBX_ERR_UNSUPPORTED  4   /* valid request outside what this build implements */
#define FBX_ERR_NOMEM        5
#define FBX_ERR_RCCL         6   /* RCCL (multi-GPU) failure, or librccl.so could not be opened */

#define FBX_KIND_STATE    0
#define FBX_KIND_PROCESS  1

/* pgdb modes */
#define FBX_MODE_CONVERGE 0   /* reference loop: stop when old_cost - new_cost < 1e-10
                                 (tomography.py:589); max_iters > 0 adds a cap */
#define FBX_MODE_FIXED    1   /* exactly max_iters outer iterations (benchmark mode) */
/* Flag, OR-ed into either mode of fbx_pgdb_process* (per call): the backtracking line search of tomography.py:575-585 taken
 * LITERALLY -- every halving evaluates the full cost sum and the acceptance test is the reference's rounded comparison
 * `new_cost > old_cost + change` -- instead of the default, which knows the cost DIFFERENCE of a small step exactly (a power
 * series in alpha) and tests that.  The two agree wherever the comparison is not decided by the rounding of the cost sums,
 * i.e. in every iteration up to the reference's own stopping point (tests hold the halving counts equal there); past it
 * (FBX_MODE_FIXED beyond convergence) the reference performs a rounding-driven walk that no other summation order
 * reproduces, and this flag yields ANOTHER such walk, not the reference's.  Slower: ~50 full cost
 * evaluations per stalled iteration.  Ignored for 3 qubits, whose kernel always evaluates this way. */
#define FBX_MODE_LS_REFERENCE 0x100

/* superoperator representations for fbx_convert */
#define FBX_REP_KRAUS   0
#define FBX_REP_CHOI    1
#define FBX_REP_SUPEROP 2
#define FBX_REP_PAULI_LIOUVILLE 3
#define FBX_REP_CHI     4

/* Choi projections for fbx_proj_choi */
#define FBX_PROJ_CP        0  /* project_superoperators.py:19  proj_choi_to_completely_positive */
#define FBX_PROJ_TP        1  /* project_superoperators.py:62  proj_choi_to_trace_preserving    */
#define FBX_PROJ_TNI       2  /* project_superoperators.py:37  proj_choi_to_trace_non_increasing */
#define FBX_PROJ_PHYSICAL_TP  3  /* project_superoperators.py:87 proj_choi_to_physical(.., True)  */
#define FBX_PROJ_PHYSICAL_TNI 4  /* project_superoperators.py:87 proj_choi_to_physical(.., False) */

typedef struct fbx_design fbx_design;   /* opaque: owns the device copy of a design */

/* ---------------------------------------------------------------- library / device */
int         fbx_version(void);
const char* fbx_last_error(void);
int         fbx_device_count(int* count);
int         fbx_set_device(int device_id);          /* one process per GPU: call once */
/* One call, several GPUs (SURVEY.md 8b "fbx_set_devices(ids, count)", 8e "host thread per device does H2D of its slab,
 * launches, D2H"; the unit that is split is the reference's independent experiment, e.g. one entry of
 * get_results_by_qubit_groups, observable_estimation.py:1145-1173).  ids[0] becomes the process's device (fbx_set_device);
 * with count > 1 the HOST-POINTER batch entry points fbx_pgdb_process[_ex] and fbx_kraus_sweep split a batch of at least
 * 2 x count items into contiguous blocks and run block g on entry g of the list, on a long-lived worker thread of the
 * library that owns that device's stream, staging pool, workspaces and a replica of the design -- no exchange between
 * devices, results bit-identical to the single-device call.  A device may be listed more than once (two workers share
 * it).  count <= 1 restores the single-device behaviour.  The *_dev entry points and everything else stay on the calling
 * thread's device; multi-PROCESS runs (one rank per GPU, fbx_comm_*) do not need this call. */
int         fbx_set_devices(const int* device_ids, int count);
int         fbx_device_name(char* buf, size_t len, int* compute_units);
int         fbx_device_id(int* ordinal, char* pci_bus_id, size_t len);   /* the selected device; "0000:05:00.0"-style id (len >= 16) */
int         fbx_synchronize(void);                  /* the calling thread's stream */
/* Frees the calling thread's cached device workspaces / staging pool.  With a device list of more than one entry
 * (fbx_set_devices) it ALSO asks every device worker to free its own and waits for them: it takes the device-list lock, so it
 * blocks while a multi-device call of any thread is running, and returns the first worker's error code if one fails. */
int         fbx_release_workspace(void);
/* Process-wide DEFAULTS, read when a kernel is launched.  A thread that needs its own value passes it per call
 * (fbx_pgdb_process_ex): changing an option changes the arithmetic of every thread's later launches.
 *   "pgdb_eig_rel_tol"  (default 1e-8), "pgdb3_eig_rel_tol" (default 1e-7; 3 qubits): while the projected-gradient
 *   iteration of fbx_pgdb_process is far from its fixed point, the eigensolver of its CP projections stops at an
 *   off-diagonal norm of <value> x the previous outer step (relative to ||H||_F) instead of always at 1e-13 --
 *   an inexact projection whose error is that fraction of the distance the estimate still moves per iteration.
 *   0 reproduces the reference's eigh-to-machine-precision trajectory iteration by iteration (tests use it);
 *   the defaults leave the converged estimates within 1e-9 of the reference's.  Range [0, 1e-3].
 *   "pgdb_host_chunk" (default 4096): items of the first and of the last stage of the pipelined host-pointer form of
 *   fbx_pgdb_process (see fbx_host_alloc); the bulk in between goes in one launch per 65 536 items.
 *   "eigh_cooperative" (default 1): fbx_eigh of a few matrices with N >= 128 spreads each matrix over the whole chip with a
 *   cooperative launch; 0 keeps one workgroup per matrix.
 *   "pgdb_packed_1q" (default 1): single-qubit fbx_pgdb_process* with at most 64 settings and at least 8192 experiments (16 384 for designs of more than 12 settings) runs
 *   64 reconstructions per wavefront, one per lane (csrc/fbx_pgdb1.hip; same line-search rule as the other kernels, the
 *   eigensolver always at full tolerance; a call that passes an explicit eig_rel_tol >= 0 to fbx_pgdb_process_ex[_dev]
 *   therefore stays on the wavefront-per-reconstruction kernel whatever its batch size, so that an experiment's iterates do
 *   not depend on how many neighbours it is batched with); 2 = the lane-per-item kernel for every batch size and every
 *   tolerance argument (diagnostics / tests), 0 = never (the wavefront-per-reconstruction kernel, which smaller batches and
 *   larger designs use).
 *   "pgdb_pieces" (default 8): two-qubit fbx_pgdb_process* of more than 1024 experiments (and single-qubit ones of 1025 .. 8191
 *   with a fixed iteration count) runs every reconstruction as that many pieces of outer iterations, drawn from a ticket counter by persistent workgroups, so that a launch does not wait for
 *   whole reconstructions at its end (csrc/fbx_pgdb_lean.hip); 1 = whole reconstructions.  Results do not depend on it.
 *   "pgdb1_binned" (default 1): the lane-per-reconstruction single-qubit kernel runs one launch per outer iteration with the
 *   reconstructions re-binned by Dykstra count in between from 2^20 experiments to convergence (2^19 for at most 12 settings;
 *   2^17 for a fixed iteration count); 2 = always, 0 = never (one persistent launch).  Results do not depend on it; the binned
 *   form holds 3.5 KB + 16 m bytes of device workspace per reconstruction and synchronises the calling thread's stream. */
int         fbx_set_option(const char* name, double value);
int         fbx_get_option(const char* name, double* value);

/* ---------------------------------------------------------------- multi-GPU
 * One process per GPU; RCCL over xGMI.  The reconstruction path shards on the batch axis (the
 * independent units of observable_estimation.py:1145-1173 get_results_by_qubit_groups,
 * tomography.py:440-451 bootstrap resamples) with NO collective on the data path; these entry
 * points cover what ranks exchange around it: broadcast of design-sized constants, all-gather
 * of result slabs, all-reduce of a summary vector.  Rank 0 obtains an id with
 * fbx_comm_unique_id, hands the FBX_COMM_ID_BYTES bytes to the other ranks by any host channel,
 * every rank calls fbx_comm_init (collective).  Collectives are enqueued on the calling thread's
 * stream (the *_dev forms are asynchronous like every other *_dev entry point).  librccl.so is
 * opened on first use. */
#define FBX_COMM_ID_BYTES 128
#define FBX_COMM_SUM 0
#define FBX_COMM_MAX 1
#define FBX_COMM_MIN 2
int fbx_comm_unique_id(uint8_t* id_out /* [FBX_COMM_ID_BYTES] */);
int fbx_comm_init(const uint8_t* id, int rank, int world);   /* = fbx_comm_init_timeout with 180 s (env FBX_RCCL_INIT_TIMEOUT) */
/* ncclCommInitRank is a collective: it blocks for as long as a peer is missing.  It runs in a helper thread of the
 * library (no library lock is held meanwhile); when it has not returned after timeout_seconds the call fails with
 * FBX_ERR_RCCL and the process can form no further communicator (every later fbx_comm_* call says so at once). */
int fbx_comm_init_timeout(const uint8_t* id, int rank, int world, double timeout_seconds);
int fbx_comm_info(int* rank, int* world, int* rccl_version);   /* world = 0 without a communicator */
/* rank / size / device as the COMMUNICATOR reports them (ncclCommUserRank, ncclCommCount, ncclCommCuDevice) */
int fbx_comm_query(int* rank, int* world, int* device);
int fbx_comm_destroy(void);
int fbx_comm_allgather_dev(const void* d_send, void* d_recv /* [world][bytes_per_rank] */,
                           size_t bytes_per_rank);
int fbx_comm_broadcast_dev(void* d_buf, size_t bytes, int root);
int fbx_comm_allreduce_f64_dev(const double* d_send, double* d_recv, size_t n, int op);
int fbx_comm_allreduce_f64(double* host_inout, size_t n, int op);   /* host vector of any length (meant for summary vectors); synchronises */
int fbx_comm_barrier(void);   /* all ranks' work enqueued by the calling threads is complete */

/* device memory helpers so callers can keep batches resident in HBM */
int fbx_malloc(void** dev_ptr, size_t bytes);
int fbx_free(void* dev_ptr);
int fbx_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes);
int fbx_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);

/* page-locked host memory: caller buffers allocated here cross PCIe at the full rate and asynchronously; when
 * expect, counts and choi_out of fbx_pgdb_process[_ex] are all page-locked and the batch exceeds one stage
 * (fbx_set_option "pgdb_host_chunk", default 4096 items), the call overlaps H2D, kernels and D2H: a small first stage covers the
 * upload of the rest, the bulk runs on a high-priority stream, a small last stage covers the download of the bulk's results */
int fbx_host_alloc(void** host_ptr, size_t bytes);
int fbx_host_free(void* host_ptr);

/* HIP-event timing of everything enqueued on the library stream between begin and end */
int fbx_timer_begin(void);
int fbx_timer_end(double* elapsed_ms);

/* ---------------------------------------------------------------- designs
 * A design is the data-independent half of an experiment: m settings on n qubits, shared by
 * every item of a batch.  Replaces the per-call rebuilding of measurement operators in
 * tomography.py:159-160 (linear inversion), :326-327 (_R), :482-486, :494-539
 * (_extract_from_results).  in_labels / paulis are [m][n_qubits] codes (in_labels may be
 * NULL for FBX_KIND_STATE); coefs[m] are the observables' real coefficients (NULL = 1).
 * n_qubits: 1..3 for process designs (4^n x 4^n Choi matrices up to 64 x 64), 1..5 for state designs
 * (density matrices up to 32 x 32, 1023 settings). */
int fbx_design_create(int n_qubits, int kind, int m, const uint8_t* in_labels,
                      const uint8_t* paulis, const double* coefs, fbx_design** out);
int fbx_design_destroy(fbx_design* design);
int fbx_design_info(const fbx_design* design, int* n_qubits, int* kind, int* m,
                    int* n_input_states);

/* ---------------------------------------------------------------- process estimators
 * fbx_pgdb_process replaces pgdb_process_estimate (tomography.py:542-594) together with
 * _extract_from_results (:494-539), _cost (:597-614), _grad_cost (:617-633) and
 * proj_choi_to_physical (operator_tools/project_superoperators.py:87-144), for a batch of B
 * independent experiments that share `design`.
 *   expect[B][m], counts[B][m]  -- ExperimentResult.expectation / .total_counts
 *   choi_out[B][D][D] complex128, D = 4^n
 *   iters_out / dykstra_out / backtracks_out [B] (may be NULL): outer iterations, total
 *   Dykstra (= eigendecomposition) iterations, total step halvings; cost_out[B] (may be
 *   NULL): final negative log-likelihood; work_out[B][4] (may be NULL): work accounting --
 *   Jacobi sweeps of the eigensolver, eigenvalue terms rebuilt by the CP projections, cost
 *   evaluations over all outcomes, power-sum reductions of the small-step line search
 *   (bench.py derives the executed flops from these counters).
 * Design sizes: 1 and 2 qubits any number of settings (above 256 / 1024 the outcome slots are streamed from HBM, slower);
 * 3 qubits up to 65 536 settings (FBX_ERR_UNSUPPORTED beyond); process designs above 3 qubits are refused by
 * fbx_design_create. */
int fbx_pgdb_process(const fbx_design* design, int64_t B, const double* expect,
                     const double* counts, int trace_preserving, int mode, int max_iters,
                     double* choi_out, int32_t* iters_out, int32_t* dykstra_out,
                     int32_t* backtracks_out, double* cost_out, int32_t* work_out);
int fbx_pgdb_process_dev(const fbx_design* design, int64_t B, const double* d_expect,
                         const double* d_counts, int trace_preserving, int mode,
                         int max_iters, double* d_choi_out, int32_t* d_iters_out,
                         int32_t* d_dykstra_out, int32_t* d_backtracks_out,
                         double* d_cost_out, int32_t* d_work_out);

/* The same estimator with per-call arguments (what concurrent callers use instead of the process-wide option):
 *   eig_rel_tol   tolerance factor of the CP projections' eigensolver for THIS call (see fbx_set_option
 *                 "pgdb_eig_rel_tol"); negative = the process default; 0 = the reference's eigh-to-machine-precision
 *                 trajectory; range [0, 1e-3]
 *   trace_out     [B][trace_iters][2] int32 (may be NULL): for every outer iteration k < trace_iters of item b the
 *                 Dykstra iterations (proj_choi_to_physical, project_superoperators.py:112-144) and the step halvings
 *                 (tomography.py:578-585) of that iteration; rows beyond the item's last iteration are zero. */
int fbx_pgdb_process_ex(const fbx_design* design, int64_t B, const double* expect,
                        const double* counts, int trace_preserving, int mode, int max_iters,
                        double eig_rel_tol, double* choi_out, int32_t* iters_out,
                        int32_t* dykstra_out, int32_t* backtracks_out, double* cost_out,
                        int32_t* work_out, int32_t* trace_out, int trace_iters);
int fbx_pgdb_process_ex_dev(const fbx_design* design, int64_t B, const double* d_expect,
                            const double* d_counts, int trace_preserving, int mode, int max_iters,
                            double eig_rel_tol, double* d_choi_out, int32_t* d_iters_out,
                            int32_t* d_dykstra_out, int32_t* d_backtracks_out, double* d_cost_out,
                            int32_t* d_work_out, int32_t* d_trace_out, int trace_iters);

/* _cost and _grad_cost (tomography.py:597-614, :617-633) as functions of their own: ONE evaluation of the negative log-likelihood
 * -n^T log(clip(A vec(E), eps)) and of its gradient -unvec(A^H (n / clip(A vec(E), eps))) at the Choi matrices choi_in[B][D][D]
 * (Hermitian, as every iterate of the estimator is), computed with the device functions of the reconstruction kernels (Pauli
 * transform, prediction table T = R C, per-state weights, R^G = -(W C^T) / d^2, inverse transform): a diagnostic that pins the
 * gradient -- which fbx_pgdb_process never outputs -- directly.  `nvec` is the reference's n of _extract_from_results
 * (tomography.py:528-538): [B][2 m], row 2 k / 2 k + 1 = the +1 / -1 counts of result k over the grand total.  `design` stands
 * for A (never materialised).  cost_out[B] or grad_out[B][D][D] may be NULL (not both).  1..3 qubits, any number of settings. */
int fbx_pgdb_cost_grad(const fbx_design* design, int64_t B, const double* nvec, const double* choi_in, double eps,
                       double* cost_out, double* grad_out);
int fbx_pgdb_cost_grad_dev(const fbx_design* design, int64_t B, const double* d_nvec, const double* d_choi_in, double eps,
                           double* d_cost_out, double* d_grad_out);

/* linear_inv_process_estimate (tomography.py:459-491): choi_out[B][D][D]. */
int fbx_linv_process(const fbx_design* design, int64_t B, const double* expect,
                     double* choi_out);
int fbx_linv_process_dev(const fbx_design* design, int64_t B, const double* d_expect,
                         double* d_choi_out);

/* ---------------------------------------------------------------- state estimators (1..5 qubits)
 * linear_inv_state_estimate (tomography.py:130-165): rho_out[B][d][d], d = 2^n. */
int fbx_linv_state(const fbx_design* design, int64_t B, const double* expect, double* rho_out);
int fbx_linv_state_dev(const fbx_design* design, int64_t B, const double* d_expect,
                       double* d_rho_out);

/* iterative_mle_state_estimate (tomography.py:168-270) incl. _R (:273-338): diluted
 * iterative MLE with optional max-entropy (entropy_penalty > 0) or hedging (beta > 0).
 * Performs at most maxiter-1 updates like the reference; hit_max_out[b] = 1 where the cap was
 * reached (the shim then emits the reference's warning). */
int fbx_mle_state(const fbx_design* design, int64_t B, const double* expect,
                  const double* counts, double epsilon, double entropy_penalty, double beta,
                  double tol, int maxiter, double* rho_out, int32_t* iters_out,
                  int32_t* hit_max_out);
int fbx_mle_state_dev(const fbx_design* design, int64_t B, const double* d_expect,
                      const double* d_counts, double epsilon, double entropy_penalty, double beta,
                      double tol, int maxiter, double* d_rho_out, int32_t* d_iters_out,
                      int32_t* d_hit_max_out);

/* _R (tomography.py:273-338): r_out[B][d][d] for given states rho[B][d][d]. */
int fbx_r_operator(const fbx_design* design, int64_t B, const double* rho, const double* expect,
                   double* r_out);
int fbx_r_operator_dev(const fbx_design* design, int64_t B, const double* d_rho,
                       const double* d_expect, double* d_r_out);

/* state_log_likelihood (tomography.py:341-375): ll_out[B] (log10). */
int fbx_state_log_likelihood(const fbx_design* design, int64_t B, const double* rho,
                             const double* expect, const double* counts, double* ll_out);
int fbx_state_log_likelihood_dev(const fbx_design* design, int64_t B, const double* d_rho,
                                 const double* d_expect, const double* d_counts,
                                 double* d_ll_out);

/* ---------------------------------------------------------------- operator tools
 * fbx_convert: the pairwise conversions of operator_tools/superoperator_transformations.py
 * :82-371.  `in` is [B][K][d][d] for FBX_REP_KRAUS (K operators per item), else [B][D][D];
 * `out` is [B][D][D].  Conversions *to* Kraus are fbx_choi2kraus below (eigenvector-valued outputs
 * are only defined up to phase, superoperator_transformations.py:325-336).  n_qubits 1..5; for
 * 4 and 5 qubits (256^2 / 1024^2 matrices, work matrices in HBM) the conversions INTO chi from a Choi /
 * superoperator / Pauli-Liouville matrix -- which the reference routes through a D x D eigendecomposition
 * (choi2kraus: chi of |C|, eigenvalues within 1e-9 dropped) -- are composed from fbx_eigh_dev, fbx_matmul_dev and the
 * linear basis change (25 ms per 256 x 256 item, ~1 s per 1024 x 1024 item: complete, not fast).  Kraus sets with more
 * operators than the fused kernels stage in LDS (K x D x 16 B against 160 KiB) go through the basis-free kernel
 * (fbx_convert_general: any K) to the Choi matrix and on from there. */
int fbx_convert(int from_rep, int to_rep, int n_qubits, int64_t B, const double* in, int K,
                double* out);
/* same with device pointers (buffers from fbx_malloc): the batch stays resident in HBM */
int fbx_convert_dev(int from_rep, int to_rep, int n_qubits, int64_t B, const double* d_in, int K,
                    double* d_out);
/* The conversions that involve no operator basis, for ANY Hilbert-space dimension `dim` (qutrits, ...;
 * 1..256): kraus -> superop (superoperator_transformations.py:100-145), kraus -> choi (:159-182),
 * superop <-> choi (:267-277, :351-361).  Shapes as above with d = dim, D = dim^2. */
int fbx_convert_general(int from_rep, int to_rep, int dim, int64_t B, const double* in, int K, double* out);
int fbx_convert_general_dev(int from_rep, int to_rep, int dim, int64_t B, const double* d_in, int K,
                            double* d_out);

/* Fused conversion sweep of BASELINE config 3: kraus2choi -> choi2pauli_liouville ->
 * choi2chi -> process_fidelity(ptm_ref, ptm) (superoperator_transformations.py:159,364,339;
 * distance_measures.py:315).  Any of choi_out / ptm_out / chi_out / fid_out may be NULL. */
int fbx_kraus_sweep(int n_qubits, int64_t B, int K, const double* kraus, const double* ptm_ref,
                    double* choi_out, double* ptm_out, double* chi_out, double* fid_out);
int fbx_kraus_sweep_dev(int n_qubits, int64_t B, int K, const double* d_kraus,
                        const double* d_ptm_ref, double* d_choi_out, double* d_ptm_out,
                        double* d_chi_out, double* d_fid_out);

/* Choi projections (operator_tools/project_superoperators.py:19-144): out[B][D][D];
 * iters_out[B] (may be NULL) = Dykstra iterations for the PHYSICAL kinds. */
int fbx_proj_choi(int proj_kind, int n_qubits, int64_t B, const double* choi, double* out,
                  int32_t* iters_out);

int fbx_proj_choi_dev(int proj_kind, int n_qubits, int64_t B, const double* d_choi, double* d_out,
                      int32_t* d_iters_out);

/* project_state_matrix_to_physical (operator_tools/project_state_matrix.py:6-52). */
int fbx_proj_state_physical(int n_qubits, int64_t B, const double* rho, double* out);
int fbx_proj_state_physical_dev(int n_qubits, int64_t B, const double* d_rho, double* d_out);

/* apply_choi_matrix_2_state (operator_tools/apply_superoperator.py:60-90): out[B][d][d]. */
int fbx_apply_choi(int n_qubits, int64_t B, const double* choi, const double* rho, double* out);
int fbx_apply_choi_dev(int n_qubits, int64_t B, const double* d_choi, const double* d_rho,
                       double* d_out);

/* tensor_channel_kraus / compose_channel_kraus (operator_tools/compose_superoperators.py:7-44) for
 * batches: k2[B][K2][rows2][cols2], k1[B][K1][rows1][cols1]; out[B][K1*K2][.][.] with operator
 * p = j*K2 + l (the reference's list order) = kron(k2[l], k1[j]) when tensor != 0, else k2[l] . k1[j]
 * (needs cols2 == rows1). */
int fbx_kraus_pairs(int tensor, int64_t B, int K2, int rows2, int cols2, int K1, int rows1, int cols1,
                    const double* k2, const double* k1, double* out);
int fbx_kraus_pairs_dev(int tensor, int64_t B, int K2, int rows2, int cols2, int K1, int rows1,
                        int cols1, const double* d_k2, const double* d_k1, double* d_out);

/* pauli_twirl_chi_matrix (operator_tools/channel_approximation.py:31-49): out[B][D][D] keeps the
 * diagonal of chi[B][D][D]. */
int fbx_pauli_twirl_chi(int64_t B, int D, const double* chi, double* out);
int fbx_pauli_twirl_chi_dev(int64_t B, int D, const double* d_chi, double* d_out);

/* entanglement_fidelity / process_fidelity (distance_measures.py:271-359) on
 * Pauli-Liouville matrices [B][D][D] (real parts of tr(A^H B) / d^2): fe_out, fp_out may be
 * NULL.  n_qubits 1..5. */
int fbx_process_fidelity(int n_qubits, int64_t B, const double* ptm0, const double* ptm1,
                         double* fe_out, double* fp_out);
int fbx_process_fidelity_dev(int n_qubits, int64_t B, const double* d_ptm0, const double* d_ptm1,
                             double* d_fe_out, double* d_fp_out);

/* State measures (distance_measures.py:14-114, :198): purity tr(rho^2), fidelity
 * (tr sqrt(sqrt(rho) sigma sqrt(rho)))^2, trace_distance = 0.5 * induced 1-norm,
 * hilbert_schmidt_ip Re tr(rho^H sigma); rho, sigma are [B][d][d]; out arrays [B], NULL to skip. */
int fbx_state_measures(int n_qubits, int64_t B, const double* rho, const double* sigma,
                       double* purity_out, double* fidelity_out, double* trace_dist_out,
                       double* hs_ip_out);
int fbx_state_measures_dev(int n_qubits, int64_t B, const double* d_rho, const double* d_sigma,
                           double* d_purity_out, double* d_fidelity_out,
                           double* d_trace_dist_out, double* d_hs_ip_out);

/* ---------------------------------------------------------------- plot inputs
 * Pauli-Liouville vector of a state, the input of plotting/state_process.py:10-87:
 * out[b][k] = (computational2pauli_basis_matrix(2 n) vec(rho_b))[k] = tr[P_k rho_b] / d (real part), P_k in the
 * order of n_qubit_pauli_basis(n).labels (utils.py:398-409; itertools.product('IXYZ', repeat=n)).
 * rho is [B][d][d] complex, out [B][d^2] real; n_qubits 1..5.  (The Pauli transfer matrix that
 * plot_pauli_transfer_matrix (:90) draws is fbx_convert(FBX_REP_CHOI -> FBX_REP_PAULI_LIOUVILLE).) */
int fbx_pauli_vector(int n_qubits, int64_t B, const double* rho, double* out);
int fbx_pauli_vector_dev(int n_qubits, int64_t B, const double* d_rho, double* d_out);

/* ---------------------------------------------------------------- shots -> moments
 * shots_to_obs_moments (observable_estimation.py:804-853), the reduction immediately before the
 * estimators: for each of n_settings settings, bits[s] is a [n_shots][n_qubits] array of 0/1 bytes
 * (one row per shot, as qc.run returns it), obs_mask[s][q] != 0 where the setting's observable acts
 * on column q, coefs[s] its real coefficient (NULL = 1).  mean_out[s] = coef * mean of the +-1
 * products, var_out[s] = variance of that mean (coef^2 (1 - m^2) / n_shots), or the Beta(n+ + 1,
 * n- + 1) moments when beta_prior != 0 (use_beta_dist_unbiased_prior).  An all-zero mask is the
 * identity term: (coef, 0). */
int fbx_shots_to_moments(int n_qubits, int64_t n_settings, int64_t n_shots, const uint8_t* bits,
                         const uint8_t* obs_mask, const double* coefs, int beta_prior,
                         double* mean_out, double* var_out);
int fbx_shots_to_moments_dev(int n_qubits, int64_t n_settings, int64_t n_shots, const uint8_t* d_bits,
                             const uint8_t* d_obs_mask, const double* d_coefs, int beta_prior,
                             double* d_mean_out, double* d_var_out);

/* Readout-calibration rescale, the arithmetic of calibrate_observable_estimates
 * (observable_estimation.py:1028-1037) with ratio_variance (:1052-1090), for B experiments x m settings:
 * mean_out = expect / cal_mean[c], err_out = sqrt(std_err^2 / cal_mean[c]^2 + expect^2 cal_var[c] /
 * cal_mean[c]^4) with c = cal_index[k] (the calibration of setting k's observable; cal_index NULL =
 * one calibration per setting, n_cal == m).  cal_mean / cal_var are the shots_to_obs_moments of the
 * calibration runs (fbx_shots_to_moments). */
int fbx_calibrate_expectations(int64_t B, int64_t m, const double* expect, const double* std_err,
                               const int32_t* cal_index, int64_t n_cal, const double* cal_mean,
                               const double* cal_var, double* mean_out, double* err_out);
int fbx_calibrate_expectations_dev(int64_t B, int64_t m, const double* d_expect,
                                   const double* d_std_err, const int32_t* d_cal_index, int64_t n_cal,
                                   const double* d_cal_mean, const double* d_cal_var,
                                   double* d_mean_out, double* d_err_out);

/* estimate_dfe (direct_fidelity_estimation.py:224-307), batched: for each of B experiments with m
 * settings on n_qubits, expect[B][m] and std_err[B][m] -> the direct fidelity estimate and its standard
 * error; kind = FBX_KIND_STATE (state fidelity) or FBX_KIND_PROCESS (average gate fidelity). */
int fbx_dfe_estimate(int n_qubits, int kind, int64_t B, int64_t m, const double* expect,
                     const double* std_err, double* mean_out, double* err_out);

/* Bootstrap resampling of expectations, _resample_expectations_with_beta (tomography.py:378-409), for
 * all resamples at once: out[r][i] = 2 Beta(n_plus_i + prior, n_minus_i + prior) - 1 with
 * n_plus = (expect_i + 1) / 2 * counts_i, i < n (= batch * settings, flattened), r < R.  The reference
 * draws from numpy's global stream; here element (r, i) owns a counter-based Philox4x32-10 stream
 * keyed by `seed` (counter = (r * n + i, draw number)), so results are reproducible and independent
 * of R and of the launch shape; parity with the reference is distributional.  Elements whose Beta
 * parameters are not positive come out as NaN.  The _dev form optionally also writes
 * d_counts_out[r][i] = counts[i] so that the R * batch resampled experiments can be handed to a
 * batched estimator as one launch. */
int fbx_beta_resample(int64_t n, int64_t R, const double* expect, const double* counts,
                      double prior_counts, uint64_t seed, double* out);
int fbx_beta_resample_dev(int64_t n, int64_t R, const double* d_expect, const double* d_counts,
                          double prior_counts, uint64_t seed, double* d_out, double* d_counts_out);

/* ---------------------------------------------------------------- random operators
 * operator_tools/random_operators.py:21-157 for batches, generated on the device.  Item b (global id
 * first_item + b) owns a counter-based Philox4x32-10 stream keyed by `seed`, so an item's matrices
 * depend on (seed, item id) only -- not on B, the launch shape or the split over GPUs.  The
 * reference draws from numpy's global Mersenne-Twister stream: parity is distributional (the
 * host generators of the Python shim keep the reference's draw order for seeded reproduction).
 *   FBX_RAND_GINIBRE        out[B][dim][cols]  ginibre_matrix_complex(dim, cols)           :21-46
 *   FBX_RAND_UNITARY        out[B][dim][dim]   haar_rand_unitary(dim)                      :49-72
 *   FBX_RAND_STATE_VECTOR   out[B][dim]        haar_rand_state(dim)                        :75-89
 *   FBX_RAND_GINIBRE_STATE  out[B][dim][dim]   ginibre_state_matrix(dim, rank)             :92-112
 *   FBX_RAND_BURES_STATE    out[B][dim][dim]   bures_measure_state_matrix(dim)             :115-132
 * dim in {2, 4, 8} except for FBX_RAND_GINIBRE (any shape); complex128 outputs.
 * fbx_random_kraus: CPTP Kraus sets out[B][K][d][d], K_j = G_j S^{-1/2}, S = sum_j G_j^H G_j with
 * Ginibre G_j -- kraus2choi of a set is a rand_map_with_BCSZ_dist(d, K) draw (:135-157). */
#define FBX_RAND_GINIBRE        0
#define FBX_RAND_UNITARY        1
#define FBX_RAND_STATE_VECTOR   2
#define FBX_RAND_GINIBRE_STATE  3
#define FBX_RAND_BURES_STATE    4
int fbx_random_operators(int kind, int dim, int cols_or_rank, int64_t B, uint64_t seed,
                         int64_t first_item, double* out);
int fbx_random_operators_dev(int kind, int dim, int cols_or_rank, int64_t B, uint64_t seed,
                             int64_t first_item, double* d_out);
int fbx_random_kraus(int n_qubits, int64_t B, int K, uint64_t seed, int64_t first_item,
                     double* kraus_out);
int fbx_random_kraus_dev(int n_qubits, int64_t B, int K, uint64_t seed, int64_t first_item,
                         double* d_kraus_out);

/* out[b] = op(a[b]) diag(scale[b]) op(b[b]) for stacks of N x N complex matrices, N in 1..1024: op = the matrix itself
 * (conj_t = 0) or its conjugate transpose (conj_t = 1); scale is [B][N] real or NULL.  The products around fbx_eigh:
 * V f(lambda) V^H (sqrtm_psd, calculational.py:77-91), sqrt(rho) sigma sqrt(rho) (fidelity, distance_measures.py:64-84). */
int fbx_matmul(int N, int64_t B, const double* a, int conj_t_a, const double* scale, const double* b, int conj_t_b,
               double* out);
int fbx_matmul_dev(int N, int64_t B, const double* d_a, int conj_t_a, const double* d_scale, const double* d_b,
                   int conj_t_b, double* d_out);

/* partial_trace (calculational.py:5-35) of operators on A (x) B, any dimensions with dim_a * dim_b <= 4096:
 * keep = 0 traces out B (out [B][dim_a][dim_a]), keep = 1 traces out A (out [B][dim_b][dim_b]); in is
 * [B][dim_a dim_b][dim_a dim_b].  Tr_out of a Choi matrix (trace preservation, validate_superoperator.py:80-97)
 * is keep = 0 with dim_a = dim_b = d; its action on the identity (unitality, :130-145) is keep = 1. */
int fbx_partial_trace(int dim_a, int dim_b, int keep, int64_t B, const double* in, double* out);
int fbx_partial_trace_dev(int dim_a, int dim_b, int keep, int64_t B, const double* d_in, double* d_out);

/* Batched Hermitian eigendecomposition with numpy.linalg.eigh / scipy.linalg.eigh semantics (the
 * LOWER triangle of a[B][N][N] is read, eigenvalues ascending).  The host form takes any N in 1..1024
 * (up to 64: in LDS, sizes that are not a power of two -- a qutrit, a 9 x 9 Choi matrix -- embedded in the
 * next power of two with decoupled zero padding; above 64: the same Jacobi with matrix and eigenvectors in
 * HBM -- one workgroup per matrix for batches, a cooperative launch over the whole chip for a few matrices; 4- and
 * 5-qubit Choi matrices, not a fast path: 25 ms for 256 x 256, 0.8 s for 1024 x 1024); the _dev form N in
 * {2, 4, 8, 16, 32, 64} or even in 66..1024.  This is the
 * primitive under choi2kraus (superoperator_transformations.py:325-336), the PSD validators
 * (validate_operator.py:118-150), proj_choi_to_unitary (project_superoperators.py:147-175),
 * sqrtm_psd (calculational.py:77-91) and the spectral distance measures (distance_measures.py:153-195,440-460).
 * w_out[B][N]; v_out[B][N][N] holds the eigenvectors as columns (phases arbitrary), may be NULL. */
int fbx_eigh(int N, int64_t B, const double* a, double* w_out, double* v_out);
int fbx_eigh_dev(int N, int64_t B, const double* d_a, double* d_w_out, double* d_v_out);

/* choi2kraus (superoperator_transformations.py:325-336) for a batch of n-qubit Choi matrices, n in 1..5 (D = 4^n, d = 2^n):
 * kraus_out[B][D][d][d] complex128, row-major d x d operators; operator i of an item is sqrt(lambda_i) unvec(v_i) for the
 * i-th eigenpair with |lambda_i| > tol in ascending eigenvalue order (the order of the reference's list; numpy's scimath
 * square root: i sqrt(|lambda|) for a negative eigenvalue), the remaining slots are zero; count_out[B] (may be NULL) = the
 * number of operators kept.  Kraus operators are defined up to the phase of each eigenvector: it is fixed so that the
 * eigenvector's first component above 1e-12 of its norm is real and positive -- the convention under which the reference's
 * entry-by-entry tests hold (tests/test_superoperator_transformations.py:215-216); compare |K| or kraus2choi(K) otherwise,
 * as :219-224, :263-271 do.  Also the tail of superop2kraus (:229-238), pauli_liouville2kraus (:280-288) and chi2kraus
 * (:195-204): fbx_convert to Choi first.  fbx_eigh + one assembling kernel; the result may alias nothing. */
int fbx_choi2kraus(int n_qubits, int64_t B, const double* choi, double tol, double* kraus_out, int32_t* count_out);
int fbx_choi2kraus_dev(int n_qubits, int64_t B, const double* d_choi, double tol, double* d_kraus_out, int32_t* d_count_out);

#ifdef __cplusplus
}
#endif
#endif /* FBX_H */
