/*
 * mpbp_hip.h - C ABI of libmpbp_hip.so: the MPBP message-update hot path on MI355X (gfx950).
 *
 * Drop-in boundary for stecrotti/MatrixProductBP.jl v0.9.0.  The reference has no FFI layer; the
 * seam this library replaces is the body of the sweep loop
 *
 *     Threads.@threads for i in nodes; onebpiter!(bp, i, eltype(bp.w[i]); svd_trunc, damp); end
 *                                                     (reference src/mpbp.jl:189-192)
 *
 * for factors that are `RecursiveBPFactor`s (reference src/recursive_bp_factor.jl:146-165), plus the
 * observables the host reads afterwards (beliefs src/mpbp.jl:237, pair_beliefs src/mpbp.jl:202-235,
 * bethe_free_energy src/mpbp.jl:298).  The host (Julia `ccall`, or the Python ctypes mirror in
 * matrixproductbp.jl_amd/) keeps the graph, evaluates the factor interface into dense tables, owns the
 * sweep loop / convergence callback, and calls in here once per sweep.
 *
 * Conventions: every call returns 0 on success or a negative MPBP_E* code and never throws or aborts
 * across the boundary; mpbp_last_error() gives the message.  All host arrays are caller-owned,
 * column-major ("first index fastest", as Julia stores them), float64 / int32, 0-based indices.
 * A context is used from one host thread at a time; calls are blocking (stream-synchronised before
 * returning) unless stated otherwise.
 */
#ifndef MPBP_HIP_H
#define MPBP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpbp_ctx mpbp_ctx;

enum {
  MPBP_OK = 0,
  MPBP_EINVAL = -1,      /* bad argument / inconsistent sizes */
  MPBP_ENOMEM = -2,      /* device or host allocation failed */
  MPBP_EHIP = -3,        /* a HIP runtime call failed */
  MPBP_EUNSUPPORTED = -4,/* valid in the reference, not implemented on the device path yet */
  MPBP_ECAPACITY = -5    /* a truncated bond exceeded the context's max_bond capacity */
};

/* SVDTrunc functors of TensorTrains.jl as used at reference src/mpbp.jl:118,186; src/mpems.jl:161. */
enum {
  MPBP_TRUNC_THRESH = 0,      /* TruncThresh(eps):          keep sigma_k > eps*||sigma||_2          */
  MPBP_TRUNC_BOND = 1,        /* TruncBond(mprime):         keep min(len, mprime)                    */
  MPBP_TRUNC_BOND_MAX = 2,    /* TruncBondMax(mprime):      as BOND, records the worst relative err  */
  MPBP_TRUNC_BOND_THRESH = 3  /* TruncBondThresh(mprime,eps): min of both                            */
};

typedef struct {
  int32_t kind;    /* MPBP_TRUNC_* */
  int32_t mprime;  /* bond cap (ignored by THRESH) */
  double eps;      /* threshold (ignored by BOND / BOND_MAX) */
} mpbp_trunc;

/*
 * Graph + sizes.  Replaces the `g::IndexedBiDiGraph` field of `MPBP` (reference src/mpbp.jl:1-33).
 * Neighbour lists are given explicitly so that the reference's aliased graphs (InfiniteRegularGraph,
 * src/infinite_graph.jl:8-20: k copies of edge (1,1,1)) are expressible: for node i, position
 * p in [nbr_ptr[i], nbr_ptr[i+1]) is its p-th neighbour, in_edge[p] the id of the message it reads,
 * out_edge[p] the id of the message it writes (same neighbour order in both, as
 * inedges(g,i)/outedges(g,i) guarantee, reference src/recursive_bp_factor.jl:149).
 */
typedef struct {
  int32_t n_nodes;          /* nv(g) */
  int32_t n_edges;          /* ne(g): number of stored messages */
  int32_t T;                /* final time; T+1 cores per message */
  int32_t q;                /* states per variable (uniform over nodes on the device path) */
  const int32_t* nbr_ptr;   /* [n_nodes+1] */
  const int32_t* in_edge;   /* [nbr_ptr[n_nodes]] */
  const int32_t* out_edge;  /* [nbr_ptr[n_nodes]] */
  int32_t max_bond;         /* capacity of every stored bond (>= the truncation cap you will use) */
  int32_t device;           /* HIP device ordinal */
  /* Storage slot of each message in the message slab (multi-GPU: ranks own contiguous slot ranges so
   * that one all-gather per sweep exchanges them).  NULL => slot = edge id, n_slots = n_edges. */
  const int32_t* slot_of_edge; /* [n_edges] or NULL */
  int32_t n_slots;             /* 0 => n_edges */
  /* Optional caller-owned device memory for the message slab (e.g. torch tensors that the host
   * all-gathers with RCCL).  NULL => the context allocates.  Sizes: see mpbp_slab_layout(). */
  void* ext_cores;   /* double[n_slots * core_slot_doubles] */
  void* ext_bonds;   /* int32 [n_slots * (T+2)] */
  void* stream;      /* hipStream_t to launch on; NULL => the context creates its own */
  /* 1: chains periodic in time (`periodic_mpbp`, reference src/mpbp.jl:399-409): the factor of the last time couples
   * x^{T+1} back to x^1 (`w[i][end](x^1, x_nbrs^{T+1}, x^{T+1})`, src/exact.jl:24-26; `_f_bp_partial` for periodic trains,
   * src/recursive_bp_factor.jl:89-101).  Messages stay OPEN trains on the device (bond-1 ends): the value of x_i^1 is
   * carried through the chain as one more factor q of the bond of the MPEM3 -> MPEM2 embedding and closed on the last
   * site - the same functions the reference represents as trace-closed PeriodicMPEM2.  Needs q*q*max_bond <= 256. */
  int32_t periodic;
} mpbp_desc;

typedef struct {
  int64_t core_slot_doubles;  /* doubles per message slot = (T+1)*max_bond*max_bond*q*q */
  int64_t core_stride;        /* doubles between consecutive cores of one message */
  int32_t bonds_per_slot;     /* T+2 */
  int32_t n_slots;
} mpbp_layout;

typedef struct {
  double maxerr;       /* TruncBondMax.maxerr equivalent: worst sqrt(sum dropped s^2 / sum s^2) */
  int64_t n_compress;  /* compress! calls executed (cavity ops + message finalisations) */
  int32_t nan_flag;    /* non-finite value met (mirrors the reference's `@error "NaN in tensor train"`) */
  int32_t capacity_flag; /* a bond was clamped to max_bond (results then differ from the reference) */
  int32_t jacobi_not_converged;
  float ms_total;      /* device time of the call (HIP events on the context's stream) */
  float ms_orth;       /* time in the dominant kernel family (orthogonalisation sweeps of `op`) */
  int32_t n_orth_launches;
  int64_t jacobi_sweeps; /* total one-sided Jacobi sweeps / calls of the call (diagnostics) */
  int64_t jacobi_calls;
} mpbp_stats;

int mpbp_create(mpbp_ctx** out, const mpbp_desc* desc);
void mpbp_destroy(mpbp_ctx* ctx);
const char* mpbp_last_error(const mpbp_ctx* ctx); /* ctx may be NULL: error of the last failed create */
int mpbp_slab_layout(const mpbp_ctx* ctx, mpbp_layout* out);
/* Device pointers of the message slab (for the host's collective); valid until mpbp_destroy. */
int mpbp_slab_pointers(const mpbp_ctx* ctx, void** cores, void** bonds);

/*
 * Factor of node `node` as dense tables: the RecursiveBPFactor interface
 * (reference src/recursive_bp_factor.jl:11-27) evaluated on the host.
 *   deg        = degree of the node (must equal nbr_ptr[node+1]-nbr_ptr[node])
 *   nstates[l] = nstates(w, l), l = 0..deg                                            (:11)
 *   nt         = 1 if the factor is the same at every time (w[i] = fill(w, T+1)), else T+1
 * Tables, all with the FIRST index fastest, one block per time if nt == T+1:
 *   prob_y  [x_next(q)][x(q)][y(nstates[deg])]                = prob_y(w, x_next, x, y, deg)         (:16)
 *   prob_xy [k(deg)] blocks of [y(nstates[1])][x_k(q)][x_i(q)] = prob_xy(w, y, x_k, x_i, k)          (:20-22)
 *   prob_yy blocks for every (d1,d2), d1 = 0..deg, d2 = 0..deg-d1, in that nesting order (d2 inner),
 *           each [y(nstates[d1+d2])][y1(nstates[d1])][y2(nstates[d2])][x_i(q)]
 *                                                              = prob_yy(w, y, y1, y2, x_i, d1, d2)  (:25-26)
 *   prob_y0 [y(nstates[0])][x_i(q)]                            = prob_y0(w, y, x_i)                  (:27)
 * prob_y_partial (:49-54) and prob_y_dummy (:59-61) are composed inside the library.
 */
int mpbp_set_factor(mpbp_ctx* ctx, int32_t node, int32_t deg, const int32_t* nstates, int32_t nt,
                    const double* prob_y, const double* prob_xy, const double* prob_yy,
                    const double* prob_y0);

/*
 * Factor of node `node` as a GENERIC `BPFactor` (reference src/bp_core.jl:1-10: only the functor
 * `w(x_next, x_neighbours, x)` exists, e.g. `GenericGlauberFactor` src/Models/glauber/glauber_bp.jl:1-20, `GenericFactor`
 * src/test_factors.jl).  The node is then updated by the exhaustive-trace path instead of the recursive one:
 * `f_bp` (src/bp_core.jl:18-57), `f_bp_dummy_neighbor` (:60-93) and the generic `onebpiter!` (src/mpbp.jl:117-154) -
 * for every out-neighbour j the product of the other z-1 incoming messages, summed over their joint states with the
 * transition table, `mpem2 |> compress!(is_orthogonal=:left)`, `normalize!`, stored WITHOUT damping (src/mpbp.jl:131);
 * the belief from all z messages, compressed before it is marginalised (:145-154).
 *   w[t][x_next + q (x + q (x_1 + q (x_2 + ... + q x_deg)))] = w_i^t(x_next | x_1 .. x_deg, x)   (neighbours in position
 *   order, 0-based states; one block if nt == 1, else T+1 blocks; the block of the last time is not read: src/bp_core.jl:41)
 * Cost and size are exponential in the degree (bond max_bond^(deg-1) before compression): the call to `mpbp_sweep`
 * fails with MPBP_EUNSUPPORTED if q^3 max_bond^(deg-1) or q^2 max_bond^deg exceeds 2048.  Not available on chains periodic in time.
 */
int mpbp_set_generic_factor(mpbp_ctx* ctx, int32_t node, int32_t deg, int32_t nt, const double* w);

/* Heterogeneous `nstates(bp, i)` (reference src/mpbp.jl:22-26: every node has its own number of states q_i; the messages
 * of edge i->j are MPEM2s over q_i x q_j, src/recursive_bp_factor.jl:155).  `mpbp_desc::q` is the LARGEST q_i; node i uses
 * its first q_node[i] states and the others are padding that carries exactly zero weight: the library zeroes phi, psi
 * and the message tables there, so normalisations, beliefs, pair beliefs and free energies are those of the unpadded
 * model (factor tables are passed for q states; their entries on padding states are ignored).  Messages are reset to the
 * reference's initial state (uniform over the real states).  Stored bond dimensions may exceed the reference's by
 * directions of zero weight where the reference's bound q_i q_j (not the truncation) limits a bond. */
int mpbp_set_node_states(mpbp_ctx* ctx, const int32_t* q_node /* [n_nodes], 1 <= q_node[i] <= q */);

/* phi[i][t][x]: double[q][T+1][n_nodes] (x fastest)  - `bp.ϕ`, reference src/mpbp.jl:4 */
int mpbp_set_phi(mpbp_ctx* ctx, const double* phi);
/* psi[e][t][x_src][x_dst]: double[q][q][T+1][n_edges] (x_src fastest) - `bp.ψ`, src/mpbp.jl:5 */
int mpbp_set_psi(mpbp_ctx* ctx, const double* psi);

/*
 * Messages `bp.μ[e]` (MPEM2, cores A[t][m,n,x_src,x_dst], reference src/mpems.jl:15-17) in packed form:
 *   bonds[e*(T+2) + t]  = left bond of core t (t = 0..T), bonds[..+T+1] = 1
 *   data: for each edge, cores in time order, each core column-major [b_t, b_{t+1}, q, q];
 *   offsets[e] = index in `data` (in doubles) of edge e's first core.
 * The messages must be normalised (z = 1), as `bp.μ` always is after set_msg!.
 * offsets[e] < 0 keeps edge e's current message (partial upload).
 */
int mpbp_set_messages(mpbp_ctx* ctx, const int32_t* bonds, const int64_t* offsets, const double* data);
int mpbp_get_bonds(mpbp_ctx* ctx, int32_t* bonds /* [n_edges*(T+2)] */);
/* offsets[e] < 0 skips edge e (partial download) */
int mpbp_get_messages(mpbp_ctx* ctx, const int64_t* offsets, double* data);
/* reset_messages!(bp) (src/mpbp.jl:72-80): uniform bond-1 messages */
int mpbp_reset_messages(mpbp_ctx* ctx);

/*
 * One pass of onebpiter! over `nodes` (reference src/mpbp.jl:190-192 with
 * src/recursive_bp_factor.jl:146-165 as the body).  All listed nodes are updated from the messages as
 * they are at entry (a Jacobi step over the list): listing an independent set of nodes is therefore
 * identical to the reference's sequential order, and single-node lists reproduce it exactly.
 * Writes the outgoing messages of the listed nodes, their belief marginals and f[i].
 * If an out-edge id occurs more than once for a node (InfiniteRegularGraph) the last occurrence is
 * the one stored, as in the reference's loop (src/recursive_bp_factor.jl:154-159).
 * Blocking: returns after the context's stream has been synchronised, so the slab may be handed to a collective on
 * any stream right away (mpbp_allgather_slots, or the caller's own RCCL call).  Node lists whose work trains do not
 * fit the device are split internally; the pieces read a snapshot, taken at entry, of the in-edge slots of the listed
 * nodes, results are those of one pass (tests/test_gpu_parity.py::test_split_sweep_equals_one_pass_and_oracle).
 * Device ownership: the batched gauge sweep (products whose Y_t exceeds 2048 rows) factors tall panels with a
 * cooperative kernel whose row-chunk workgroups wait for each other inside one launch; it is only launched when the grid
 * fits 3/4 of the CUs, which guarantees co-residency IF this process is the only one computing on the device.  With
 * other work resident (a second process, another stream of the caller) an arrival counter can time out: the kernel then
 * leaves its panel untouched, the library repeats the batch with one launch per column step and keeps the context in
 * that mode - slower, never wrong.  MPBP_DEBUG_NO_COOP_PANEL=1 starts in that mode (ranks that share a device).
 * Contexts on ONE device share two internal CU-masked streams (the look-ahead of the batched QR; created on first use,
 * destroyed when the last context of the device is destroyed): one host thread per context is fine, but calls into
 * different contexts of the same device must not run concurrently.
 */
int mpbp_sweep(mpbp_ctx* ctx, const int32_t* nodes, int32_t n_nodes, mpbp_trunc trunc, double damp,
               mpbp_stats* stats /* may be NULL */);

/* beliefs(bp) (src/mpbp.jl:237): out[x + q*(t + (T+1)*i)] */
int mpbp_beliefs(mpbp_ctx* ctx, double* out);
/* `bp.b[node]` as a normalised MPEM1 train (cores [b_t, b_{t+1}, q], z = 1): what twovar_marginals /
 * autocorrelations (src/mpbp.jl:245-255) read.  bonds: int32[T+2]; data == NULL only fills `bonds` (size query). */
int mpbp_get_belief_train(mpbp_ctx* ctx, int32_t node, int32_t* bonds, double* data, int64_t data_capacity);
/* pair_beliefs(bp) (src/mpbp.jl:202-235): out[x_src + q*(x_dst + q*(t + (T+1)*e))];
 * logz_pair[e] = log z_ij of the edge (the host folds (1/d_j - 1/2) weights, src/mpbp.jl:230) */
int mpbp_pair_beliefs(mpbp_ctx* ctx, double* out, double* logz_pair);
/* bp.f (src/recursive_bp_factor.jl:163); bethe_free_energy(bp) = sum (src/mpbp.jl:298) */
int mpbp_free_energy(mpbp_ctx* ctx, double* f_node /* [n_nodes] */);
/* log z_i and sum_j log z_{i->j} of the last update of every node (diagnostics / parity tests) */
int mpbp_logz(mpbp_ctx* ctx, double* logz_node /* [n_nodes] */, double* logz_msg /* [n_edges] */);

/* Two-time marginals of the beliefs of the listed nodes, computed on the device from the belief trains `bp.b[i]`
 * (TensorTrains `twovar_marginals`; callers: autocorrelations / autocovariances / alternate_marginals, reference
 * src/mpbp.jl:245-286): out[k][t][u][x + q*y] = p_{nodes[k]}(x^t = x, x^u = y) for t < u <= t + maxdist (maxdist <= 0:
 * all), each q x q block normalised, zero elsewhere.  `out`: host, n_nodes * (T+1)^2 * q^2 doubles. */
int mpbp_twovar_marginals(mpbp_ctx* ctx, const int32_t* nodes, int32_t n_nodes, int32_t maxdist, double* out);

/* Exchange step of the multi-GPU path (replaces the visibility of `bp.mu[idx(e)] = muj`,
 * src/recursive_bp_factor.jl:177, across GPUs): one in-place RCCL all-gather of the rank-major message slab and one of
 * the bond table on the context's stream, then a stream synchronise.  The context must have been created with
 * n_slots = world * slots_per_rank and slot_of_edge mapping every edge into its owner's slot range
 * [rank * slots_per_rank, (rank+1) * slots_per_rank).  `nccl_comm` is an ncclComm_t of the caller's RCCL. */
int mpbp_allgather_slots(mpbp_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world, int32_t slots_per_rank);

/* on = 1: per-launch HIP-event timing of the dominant kernel family (mpbp_stats.ms_orth; costs a sync per launch);
 * on = 2: additionally the in-kernel phase timers read by mpbp_phase_profile; 0: off. */
int mpbp_set_profiling(mpbp_ctx* ctx, int32_t on);
/* Workgroup-seconds spent per engine phase since the last reset (profiling on), summed over workgroups;
 * phases in the order of wg::PH_* (csrc/wg_blocks.h). */
int mpbp_phase_profile(mpbp_ctx* ctx, double* seconds, int32_t n, int32_t reset);

/* Self-test entry points used by tests/ (device building blocks against host references). */
int mpbp_selftest_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const double* A, const double* B,
                       double* C);
int mpbp_selftest_qr(int32_t device, int32_t rows, int32_t cols, const double* A, double* R);
int mpbp_selftest_qr_bench(int32_t device, int32_t rows, int32_t cols, int32_t nblocks, int32_t reps, double* ms_out);
int mpbp_selftest_jacobi_bench(int32_t device, int32_t m, int32_t n, int32_t nblocks, int32_t variant,
                               int32_t reps, double* ms_out, double* avg_sweeps);
int mpbp_selftest_svd(int32_t device, int32_t rows, int32_t cols, const double* A, double* sigma,
                      double* V);
/* the multi-launch (grid-level) one-sided Jacobi of the batched truncating sweep on one m x n matrix (n <= m <= 1024):
 * column norms after convergence and the number of sweeps (-1: not converged) */
int mpbp_selftest_jacobi_grid(int32_t device, int32_t m, int32_t n, const double* A, double* sigma, int32_t maxsweeps,
                              int32_t* sweeps);
/* the same through the two-level (block) Jacobi: column blocks, block pairs LDS resident, one launch per round of the
 * block tournament - the form the truncating sweep takes for factors of ~100 columns and more */
int mpbp_selftest_jacobi_block(int32_t device, int32_t m, int32_t n, const double* A, double* sigma, int32_t maxsweeps,
                               int32_t* sweeps);
/* nprob independent rows x cols matrices (A: [nprob][rows x cols] column-major) through the grid-level batched QR of
 * the gauge sweep (csrc/v2_kernels.h); R: [nprob][min(rows,cols) x cols]; force_tall: column-step panels always. */
int mpbp_selftest_qr_batched(int32_t device, int32_t rows, int32_t cols, int32_t nprob, int32_t force_tall,
                             const double* A, double* R, double* ms_out);
/* a SEQUENCE of single matrices rows[s] x cols through the same batched QR on ONE shared scratch sized for the largest
 * (what consecutive time steps of a gauge sweep do); A, R concatenated; path[s] = the form taken (1 communication-
 * avoiding tree, 2 look-ahead panels, 0 launch per panel) */
int mpbp_selftest_qr_batched_seq(int32_t device, int32_t nshape, const int32_t* rows, int32_t cols, const double* A,
                                 double* R, int32_t* path);

#ifdef __cplusplus
}
#endif
#endif /* MPBP_HIP_H */
