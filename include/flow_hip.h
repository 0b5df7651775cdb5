/*
 * flow_hip.h -- C ABI of libflow_hip.so, the MI355X (gfx950) implementation of
 * the Navier-Stokes pressure-correction / heat hot path of nschloe/flow.
 *
 * The reference has no FFI: its boundary is the Python API of
 * flow/navier_stokes/pressure_correction.py (Chorin/IPCS/Rotational.step) and
 * flow/heat.py (Heat).  Every numerical operation the reference delegates to
 * FFC-generated tabulate_tensor + DOLFIN Assembler + PETSc is exported here as
 * a plain-C entry point; each declaration cites the reference lines whose work
 * it replaces (paths relative to the reference root).  The Python host code in
 * flow_amd/ binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *     the library never allocates or frees device memory and keeps no state
 *     (one exception: a 64-byte host-coherent mailbox per host thread, through
 *     which a kernel of the stream hands scalar results to the host);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it; only the *_solve entry points and the *_host
 *     readbacks synchronise it;
 *   - fp64 values, int32 indices; vector fields are component-blocked
 *     (dof (a, i) -> a*n + i);
 *   - per-cell arrays are SoA with the cell index fastest ([k][cell]);
 *   - return codes: 0 ok, 1 not converged, 2 invalid argument, 3 HIP error;
 *     flow_last_error() returns the message of the last failure on the
 *     calling thread.
 */
#ifndef FLOW_HIP_H
#define FLOW_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOW_OK 0
#define FLOW_NOT_CONVERGED 1
#define FLOW_INVALID 2
#define FLOW_HIP_ERROR 3

/* CSR-stream tiling of the SpMV kernels (row blocks are built on the host): at
 * most FLOW_SPMV_ROWS_PER_BLOCK rows and flow_spmv_tile_nnz(kind) nonzeros per
 * block -- FLOW_SPMV_NNZ_PER_BLOCK for the kinds that park two products per
 * nonzero in LDS (2 and 4), as many or twice as many for the others (a build
 * constant of the library: ask). */
#define FLOW_SPMV_ROWS_PER_BLOCK 256
#define FLOW_SPMV_NNZ_PER_BLOCK 1022
int flow_spmv_tile_nnz(int kind);

const char* flow_last_error(void);
int flow_abi_version(void);
/* Kernel launches the library has issued so far in this process (all entry
 * points, all streams): differences over a window give launches per step. */
int flow_launch_count(unsigned long long* count_host);
/* Iteration bodies as HIP graphs (flow_amd/csrc/graph_replay.hip) -- OFF by
 * default.  The Krylov and defect-correction loops of this library
 * (flow_cg_solve*, flow_gmres_solve, flow_mass_solve*) issue the same chain of
 * launches every iteration; each can be captured once and replayed, with
 * bit-identical results.  Measured on MI355X / ROCm 7.2 the replay is SLOWER
 * than the launches it replaces at every size of the workload (2.61 against
 * 2.38 ms per step on the eighth-size proxy; graph_replay.hip has the table),
 * hence the default.
 * mode 0 (default, FLOW_AMD_GRAPHS unset): never; 1 (FLOW_AMD_GRAPHS=1): always;
 * 2 (FLOW_AMD_GRAPHS=auto): for systems of up to auto_rows rows (< 0: leave;
 * default 1.5e6, FLOW_AMD_GRAPH_ROWS); mode | (sites << 4) with sites != 0
 * also chooses the loops (1 CG, 2 GMRES, 4 mass solver; default all).  A graph
 * is captured the first time a body is seen and kept under a hash of every
 * value that reaches its kernels (the step size of the matrix-free Jacobian
 * travels through device memory instead); a replay counts as ONE launch in
 * flow_launch_count.
 * flow_graph_stats: stats_host[0..7] = graphs kept, captures so far, replays
 * so far, kernel nodes those replays carried, captures by loop (CG, GMRES,
 * mass solver), 0. */
int flow_graph_mode(int mode, long long auto_rows);
int flow_graph_stats(unsigned long long* stats_host);
/* Host copy of the workgroup -> tile mapping the CSR-stream kernels use
 * (XCD-aware, a permutation of [0, nblocks)); for tests. */
int flow_xcd_tile_host(int block, int nblocks);

/* A linear operator over ONE scalar CSR pattern (n rows, nnz entries).
 *   kind 0: scalar              y = A0 x                      (size n)
 *   kind 1: block diagonal      y_a = A_a x_a, a = 0,1        (size 2n)
 *   kind 2: full 2x2 blocks     y_a = sum_c A_{2a+c} x_c      (size 2n)
 *   kind 3: matrix-free         y = J(ui) x, `matfree` points to a
 *           flow_momentum_jvp (below); no pattern, no value planes (size 2n)
 *   kind 4: one plane, both components, Dirichlet rows by mask   (size 2n)
 *           y_a[i] = rowmask[a*n+i] ? (A0 x_a)[i] : x_a[i]
 *           The "identity rows" form of a component-wise Dirichlet-eliminated
 *           operator whose blocks are the same matrix (the vector mass matrix
 *           of the velocity correction, pressure_correction.py:451-464): on
 *           vectors that vanish on the Dirichlet dofs -- every CG direction
 *           when the start vector carries the boundary values -- it IS the
 *           symmetrically eliminated operator, and the matrix is streamed once
 *           for both components (0.76 GB instead of 1.35 GB per product on the
 *           headline workload).
 * Replaces the PETSc AIJ matrices behind `solve`/`assemble`
 * (pressure_correction.py:224-254, 326-339, 419-432, 451-464; heat.py:88,106). */
typedef struct {
  int kind;
  int n;
  int nnz;
  int nblocks;             /* number of CSR-stream row blocks */
  const int* rowptr;       /* n+1 */
  const int* cols;         /* nnz */
  const int* rowblocks;    /* nblocks+1 */
  const double* vals[4];   /* value planes, nnz each; every plane 16-B aligned
                              and readable up to index nnz (the SpMV loads
                              value PAIRS); cols likewise readable at nnz */
  const void* matfree;     /* kind 3 only: const flow_momentum_jvp* */
  const unsigned char* rowmask;   /* kind 4 only: 2n bytes, 0 = Dirichlet row */
} flow_operator;

/* ---- K8: SpMV (PETSc MatMult inside every Krylov solve; heat.py:101) ---- */
int flow_operator_apply(const flow_operator* A, const double* x, double* y,
                        void* stream);
/* dinv[a*n+i] = 1 / A_aa[i,i]  (K10 Jacobi; replaces 'hypre_amg',
 * pressure_correction.py:331,414,456). diag_idx[i] = position of (i,i). */
int flow_operator_diag_inv(const flow_operator* A, const int* diag_idx,
                           double* dinv, void* stream);

/* Per-kernel timer of the roofline kernel in solver context: from
 * _begin on, up to max_launches launches of the fused-dot CSR-stream SpMV of a
 * scalar operator with `rows` rows -- the product with A inside a CG iteration,
 * with whatever the rest of the iteration left in the caches -- are bracketed by
 * HIP events on the launch stream; _end waits for them and returns the summed
 * durations (microseconds) and the number of launches seen.  One profile at a
 * time, not thread safe: a measuring aid for bench.py. */
int flow_profile_spmv_begin(int rows, int max_launches);
int flow_profile_spmv_end(double* total_us, int* launches);
/* What such an event pair reads around a NULL kernel (median of 33, behind a
 * busy stream): the dispatch latency every bracketed launch includes on top of
 * the kernel's execution time -- a profiler's kernel duration does not. */
int flow_profile_event_overhead(double* overhead_us, void* stream);
/* Launches an empty kernel `profile_marker_kernel` with `id` workgroups (1 ..
 * 1024) on the stream: bench.py brackets its timed steps with ids 1 and 2, and
 * profiles/summarize.py cuts a rocprofv3 kernel trace at those launches instead
 * of at guessed offsets. */
int flow_profile_marker(int id, void* stream);
/* dst[0..n) = src[0..n) with a plain 16-byte-per-lane streaming kernel (n even,
 * buffers 16-byte aligned): the measured copy ceiling of the box,
 * 16 n bytes of traffic per launch; and the same for a kernel that only reads
 * (8 n bytes; sink: one double, never written for finite data): what a
 * read-dominated kernel like the SpMV can be held against. */
int flow_profile_stream_copy(size_t n, const double* src, double* dst,
                             void* stream);
int flow_profile_stream_read(size_t n, const double* src, double* sink,
                             void* stream);

/* ---- K9: BLAS-1 (PETSc VecDot/VecAXPY/VecNorm) -------------------------- */
int flow_dot_host(int n, const double* x, const double* y, double* work,
                  double* result_host, void* stream);
/* vals[k] *= d[row(k)] for a scalar CSR plane of n rows: row equilibration
 * before a Krylov solve that stands in for the reference's sparse LU
 * (flow/heat.py:117-121) */
int flow_scale_rows(int n, const int* rowptr, const double* d, double* vals,
                    void* stream);

/* kind 0: l2, 1: linf (norm(vec,'linf'), tests/test_karman_vortex_street.py:268) */
int flow_norm_host(int n, const double* x, int kind, double* work,
                   double* result_host, void* stream);
int flow_axpby(int n, double a, const double* x, double b, double* y,
               void* stream);                       /* y = a x + b y */
int flow_vmul(int n, double a, const double* x, const double* y, double* out,
              void* stream);                        /* out = a x .* y */
int flow_fill(int n, double value, double* y, void* stream);   /* y = value */
/* 64-bit fingerprints of nfields <= 2 device vectors (sum of the entries' bit
 * patterns times odd multipliers of their index, modulo 2^64 -- integer
 * arithmetic: equal values give equal fingerprints bit for bit; beyond 2^22
 * entries every k-th 64-byte line): which trajectory a call continues is a
 * matter of the VALUES of the fields it is handed (callers copy what a step
 * returns: `u0.assign(u1)`, tests/test_karman_vortex_street.py:241-242).
 * Asynchronous: slots (device, 2*nfields doubles) receive the low and the high
 * 32 bits of each fingerprint as exact integers; x_host / n_host are HOST
 * arrays; work: FLOW_REDUCE_WORK doubles.  flow_read_doubles brings n <= 64
 * doubles of device memory to the host in stream order (one synchronisation). */
int flow_fingerprint(int nfields, const double* const* x_host,
                     const int* n_host, double* work, double* slots,
                     void* stream);
int flow_read_doubles(const double* dev, int n, double* host, void* stream);
/* y = sum_k coef_host[k] * x_host[k], k < nterms <= 6, in one pass (coef_host
 * and the pointer list x_host are HOST arrays; y must not be one of the x): the
 * start vectors a time loop extrapolates from its previous increments */
int flow_lincomb(int n, int nterms, const double* coef_host,
                 const double* const* x_host, double* y, void* stream);
/* Its weights for an extrapolation in time (HOST arithmetic, no GPU): for past
 * increments over steps of sizes dts_host[0 .. m) (newest first, m <= 6), what
 * is smooth in time is the rate increment / dt^power at the mid point of its
 * step; a polynomial of `degree` (<= 0 or >= m: m - 1, interpolation) is fitted
 * through the rates by least squares and evaluated at the middle of the new
 * step of size dt:  increment(dt) ~ sum_i w_host[i] * increment_i. */
int flow_extrapolation_weights(int m, const double* dts_host, double dt,
                               int power, int degree, double* w_host);
/* dst[a*dst_stride + k] = src[a*src_stride + idx[k]], k < m, a < ncomp: the
 * vertex values of a P2 field (the linearisation point of the P1 level of
 * flow_pmg) */
int flow_gather_rows(int ncomp, const int* idx, int m, const double* src,
                     int src_stride, double* dst, int dst_stride, void* stream);
#define FLOW_REDUCE_WORK 4096   /* doubles of `work` the reductions need */

/* Two-level additive preconditioner  z = D^-1 r + P Ac^-1 P^T r  for scalar
 * SPD systems (stands in for the reference's 'hypre_amg', pressure_correction.py
 * :331,414-418): Jacobi plus a piecewise-constant aggregate coarse space whose
 * Galerkin operator Ac = P^T A P is inverted once on the host (dense; for the
 * singular Neumann operator the pseudo-inverse).  Aggregates: agg_of[i] in
 * [0,nc) or -1 (Dirichlet dofs); agg_ptr/agg_dofs list the dofs per aggregate. */
typedef struct {
  int n;                   /* fine dofs */
  int nc;                  /* aggregates */
  const int* agg_ptr;      /* nc+1 */
  const int* agg_dofs;     /* fine dofs sorted by aggregate */
  const int* agg_of;       /* n */
  int lda;                 /* row stride of Ainv in floats, multiple of 4, >= nc */
  const float* Ainv;       /* nc rows of lda floats (pad = 0), 16-B aligned: the
                              coarse inverse is only a preconditioner, so it is
                              kept in fp32 (half the HBM bytes of the per-
                              iteration dense product); sums run in fp64 */
} flow_coarse;

/* Smoothed-aggregation multigrid V(1,1)-cycle as a CG preconditioner for scalar
 * SPD systems (the AMG-class stand-in for 'hypre_amg', pressure_correction.py
 * :331,414-418).  The hierarchy is built once per (operator, BC
 * set) on the host (flow_amd/fem/multigrid.py): aggregates of ~3x3 vertices,
 * prolongation P = (I - 4/(3 rho) D^-1 A) P0, Galerkin operators R A P with
 * R = P^T, until the coarsest level is small enough for a dense (pseudo-)
 * inverse.  One application is the textbook cycle (zero start, one damped-
 * Jacobi sweep before and after the coarse correction)
 *   x = w D^-1 r ; r' = R (r - A x) ; [coarse] ; x += P x' ; x += w D^-1 (r - A x)
 * regrouped so that a level costs three launches and ONE product with A:
 *   t = r - Ah r ; r' = R t ; [coarse] ; x = Ps x' + w D^-1 (r + t)
 * with Ah = A w D^-1 (columns scaled; shares A's pattern arrays) and
 * Ps = (I - w D^-1 A) P, both formed at setup -- or, when the hierarchy also
 * carries C = R (I - Ah) (formed at setup like the Galerkin products) and row
 * blocks common to Ps and Ah, TWO launches:
 *   r' = C r ; [coarse] ; x = Ps x' + w D^-1 (2 r - Ah r)
 * (the second kernel does both products of its rows): no intermediate vector,
 * one launch less per level -- what the small levels' cost consists of --, and
 * on the strips of K15 the restriction needs no ghost rows of r at all.
 * All operators are kind-0 flow_operators (Ps, R, C rectangular: `n` = rows).
 * Levels 0 .. nlevels-1; the last one is solved with the dense inverse. */
#define FLOW_MG_MAX_LEVELS 8
typedef struct {
  int nlevels;
  flow_operator Ah[FLOW_MG_MAX_LEVELS];   /* A[l] w D[l]^-1, l < nlevels-1 (level 0 = the system) */
  const double* dinv[FLOW_MG_MAX_LEVELS]; /* 1 / diag A[l] */
  flow_operator Ps[FLOW_MG_MAX_LEVELS];   /* n_l x n_{l+1} */
  flow_operator R[FLOW_MG_MAX_LEVELS];    /* n_{l+1} x n_l, R = P^T */
  double* r[FLOW_MG_MAX_LEVELS];          /* work vectors of level l >= 1 (n_l) */
  double* x[FLOW_MG_MAX_LEVELS];
  double* t[FLOW_MG_MAX_LEVELS];          /* all levels l < nlevels-1 (n_l) */
  int nc, lda;                            /* coarsest level: size, row stride */
  const float* Ainv;                      /* nc rows of lda floats, as flow_coarse */
  double omega;                           /* Jacobi damping w (0.8) */
  /* the two-launch form: used for ALL levels when C[0].rowptr != NULL */
  flow_operator C[FLOW_MG_MAX_LEVELS];    /* n_{l+1} x n_l: R[l] (I - Ah[l]) */
  const int* up_rowblocks[FLOW_MG_MAX_LEVELS];  /* row blocks of level l that hold
                                             <= flow_spmv_tile_nnz(0) nonzeros of
                                             Ps[l] AND of Ah[l] */
  int up_nblocks[FLOW_MG_MAX_LEVELS];
} flow_mg;
/* z = D^-1 r + P Ac^-1 P^T r: one application of the two-level preconditioner
 * (for callers that drive their own Krylov loop: the preconditioned MINRES of
 * flow_amd/stokes.py).  work: 2*coarse->lda doubles, 16-byte aligned. */
int flow_two_level_apply(const flow_coarse* coarse, const double* dinv,
                         const double* r, double* z, double* work,
                         void* stream);

/* z = V-cycle(r) on level 0 (n = its size): one preconditioner application */
int flow_mg_apply(const flow_mg* mg, int n, const double* r, double* z,
                  void* stream);

/* ---- K11: multicolour ILU(0) ----------------------------------------------
 * (replaces the sparse LU of the Newton solve, pressure_correction.py:224-254,
 * and `LUSolver`, heat.py:117-121, as a BiCGStab preconditioner).  The plan is
 * built once per pattern on the host (flow_amd/fem/ilu.py): rows are ordered by
 * colour (independent sets), so factorisation and both triangular sweeps are
 * one launch per colour.  The sweep streams are sliced ELL: inside a colour the
 * rows are sorted by their L / U row lengths and cut into slices of 64 rows (one
 * wavefront); a slice is stored column-major, padded to its longest row (pad:
 * value 0, column 0), so lane i reads entry k of its row at off + 64 k + i --
 * every load coalesced, the row sum stays in registers, no LDS, no barrier.
 * The *_host members are HOST arrays; everything else is device memory.
 * Factor buffer of one block (lu_size doubles, 16-B aligned):
 *   [0, nnz) combined L\U in the permuted CSR | [off_l, +nnz_l) L stream |
 *   [off_u, +nnz_u) U stream | [off_d, +n) inverse pivots. */
#define FLOW_ILU_SLICE 64
typedef struct {
  int n, nnz, ncolors;
  int nnz_l, nnz_u;          /* entries of the sliced streams INCLUDING padding */
  int off_l, off_u, off_d, lu_size;
  int max_row;               /* longest row of the pattern */
  int nslices;               /* slices over all colours (L and U share them) */
  const int* color_ptr_host; /* ncolors+1 (host): row range of every colour */
  const int* slice_ptr_host; /* ncolors+1 (host): slice range of every colour */
  const int* rowptr;         /* n+1, permuted (colour-major) numbering */
  const int* cols;           /* nnz, ascending per row, permuted numbering */
  const int* diag;           /* n: position of the diagonal entry */
  const int* src_pos;        /* nnz: entry -> position in the operator plane */
  const int* old_of_new;     /* n: permuted row -> original dof */
  const int* new_of_old;     /* n: original dof -> permuted row */
  const int* slice_row;      /* nslices: first row of the slice */
  const int* l_slice_off;    /* nslices+1: entry offsets (multiples of 64) */
  const int* l_cols;         /* nnz_l */
  const int* l_pos;          /* nnz_l: position in the combined factor, -1 = pad */
  const int* u_slice_off;
  const int* u_cols;         /* nnz_u */
  const int* u_pos;
} flow_ilu_plan;
typedef struct {
  const flow_ilu_plan* plan;
  int nblocks;               /* 1 (scalar) or 2 (diagonal blocks of a 2-field op) */
  const double* lu;          /* nblocks * lu_size */
  const float* packed;       /* NULL, or (nnz_l + nnz_u) * nblocks floats filled
                                by flow_ilu0_pack: the sweep streams rounded to
                                fp32, blocks interleaved -- half the bytes per
                                application (fp64 arithmetic throughout; a
                                preconditioner only) */
  int single_vector;         /* with packed: the sweep vector is kept in fp32 as
                                well (8 instead of 16 B per gather; the row
                                sums accumulate in fp64).  The application is
                                then not exactly linear: for FLEXIBLE Krylov
                                methods only -- flow_gmres_solve is one (it
                                keeps Z_j = M^-1 V_j and updates x with it) */
  const void* cycle;         /* NULL, or a const flow_tl* (K19 below): wherever a
                                solver applies this preconditioner it runs the
                                two-level cycle that flow_tl describes instead
                                of the bare sweeps (the flow_ilu a flow_tl names
                                as its smoothers must have cycle == NULL) */
} flow_ilu;
/* HOST routine (setup, no GPU needed): first-fit greedy colouring of the graph
 * of a CSR pattern in row order; colour: n ints out, *ncolors <= 63 */
int flow_color_greedy_host(int n, const int* rowptr, const int* cols,
                           int* colour, int* ncolors);
/* HOST routine: `rounds` passes of Culberson's iterated greedy on a proper
 * colouring (colour / *ncolors in and out): the rows are recoloured first-fit
 * class by class -- classes in reverse order on even passes, largest first on
 * odd ones, rows of a class in row order --, which can never need more colours
 * and usually needs fewer (P2 triangulations: 9 -> 7): every colour less is
 * two dependent launches less per ILU(0) application (K11). */
int flow_color_iterate_host(int n, const int* rowptr, const int* cols,
                            int* colour, int* ncolors, int rounds);
/* factor nblocks (1|2) value planes over the plan's pattern into lu
 * (nblocks * lu_size); the blocks share one pass over the index structure */
int flow_ilu0_factor(const flow_ilu_plan* plan, int nblocks,
                     const double* avals0, const double* avals1, double* lu,
                     void* stream);
/* fill `packed` (16-byte aligned) from the factors in ilu->lu; ilu->packed may
 * point at it from then on.  Call again after every flow_ilu0_factor. */
int flow_ilu0_pack(const flow_ilu* ilu, float* packed, void* stream);
/* z = blockdiag(LU)^-1 r in the ORIGINAL numbering; r, z, work: nblocks*n */
int flow_ilu0_solve(const flow_ilu* ilu, const double* r, double* z,
                    double* work, void* stream);

/* ---- K17: p-multigrid / Chebyshev preconditioner ---------------------------
 * (a second stand-in for the sparse LU of the Newton solve,
 * pressure_correction.py:224-254, as the right preconditioner of
 * flow_gmres_solve).  Two levels on the SAME mesh, both made of CSR-stream
 * products only (no dependent sweeps):
 *   fine    the two diagonal blocks of the assembled Jacobian J = dF1/dui over
 *           the scalar P2 pattern, smoothed with `pre` / `post` steps of the
 *           Chebyshev iteration for D^-1 A on [lam_min, lam_max];
 *   coarse  the P1 discretisation of the same linearised operator (P1 is a
 *           subspace of P2: a vertex dof copies its vertex, an edge dof
 *           averages its two end points), packed the same way;
 *           `coarse_steps` Chebyshev steps from a zero start.
 * One application z = M^-1 r:
 *   x = cheb_pre(r) ; rc = P^T (r - A x) ; xc = cheb_coarse(rc) ;
 *   x += P xc ; x += cheb_post(r - A x).
 * The matrices are stored row-scaled (D^-1 A, entries of size <= ~1) in fp16,
 * the two blocks interleaved: vals = nnz half2 (J00, J11)/diag per nonzero -- 8 B
 * per nonzero with the index; a smoother needs no more (same GMRES counts as
 * with fp64 entries).  Vectors inside are fp32, both components interleaved
 * (float2 per dof): the application is not exactly linear in r -- for FLEXIBLE
 * Krylov methods only (flow_gmres_solve is one).  Rows of Dirichlet dofs are
 * identity rows of J: z = r there (bc_fine); the coarse residual is zeroed on
 * the Dirichlet rows of the coarse operator (bc_coarse), whose rows are
 * identity rows too.
 * Level arrays: CSR-stream row blocks of at most FLOW_SPMV_ROWS_PER_BLOCK rows
 * and FLOW_PMG_NNZ_PER_BLOCK nonzeros (a lane loads nonzeros in QUADS, 16-byte
 * loads: the tile base is aligned down to a multiple of four); cols and
 * vals 16-byte aligned and readable three entries past nnz. */
#define FLOW_PMG_NNZ_PER_BLOCK 2044
typedef struct {
  int n, nnz, nblocks;
  const int* rowptr;       /* n+1 */
  const int* cols;         /* nnz */
  const int* rowblocks;    /* nblocks+1: CSR-stream row blocks */
  const void* vals;        /* nnz half2 (filled by flow_pmg_pack) */
  const float* diag;       /* n float2: diagonal of the two blocks */
  const float* dinv;       /* n float2: 1 / diagonal */
  double lam_min, lam_max; /* Chebyshev interval for D^-1 A (flow_pmg_lambda_max
                              gives the upper end) */
  /* optional: the column indices as 16-bit offsets from the lowest column of
   * each row block (6 B per nonzero instead of 8; nnz ushorts, 16-byte aligned,
   * readable three past nnz); used instead of cols when not NULL.  Needs every
   * block's columns to span < 65536 (flow_pmg_cols16 reports otherwise). */
  const void* cols16;
  const int* cbase;        /* nblocks */
  /* optional: ONE plane for both components -- the mean of the two blocks
   * (they differ by the reaction term of the Newton linearisation, of the size
   * of the off-diagonal blocks the cycle drops; the mean is the Oseen operator)
   * -- as a 32-bit word per nonzero: fp16 value | 16-bit column offset from
   * cbase (4 B per nonzero; nnz words, 16-byte aligned, readable three past
   * nnz; filled by flow_pmg_pack1).  Used instead of vals / cols16 when not
   * NULL; idrows (2 n, component-blocked) flags the identity rows of each
   * component, which the plane cannot carry. */
  const void* packed;
  const unsigned char* idrows;
} flow_pmg_level;
typedef struct {
  flow_pmg_level fine, coarse;
  int pre, post, coarse_steps;     /* Chebyshev steps: pre, post 1..3, coarse >= 1 */
  const int* ends;                 /* fine.n int2: coarse rows of a fine dof */
  const int* rptr;                 /* coarse.n+1: restriction lists ... */
  const int* rsrc;                 /* ... of fine dofs; the FIRST entry of a list
                                      is the vertex's own dof (weight 1), the
                                      others are edge dofs (weight 1/2) */
  const unsigned char* bc_fine;    /* 2*fine.n bytes (component-blocked), NULL: none */
  const unsigned char* bc_coarse;  /* 2*coarse.n bytes, NULL: none */
  float* work;                     /* 12*fine.n + 8*coarse.n + 2 floats, 16-B
                                      aligned; the LAST float2 must be zero and
                                      is never written (dummy coarse row) */
  int scalar;                      /* != 0: a SCALAR operator of size fine.n (the
                                      heat system, flow/heat.py:103-122): both
                                      planes of the levels hold the same matrix,
                                      r and z have fine.n entries, the masks are
                                      duplicated; 0: the two velocity blocks */
} flow_pmg;
/* vals[k] = half2((a00, a11)[k] / their diagonal entries of row(k)), 0 where
 * keep[k] == 0 (keep = NULL: everything is kept);
 * diag[i] = (a00, a11)[diag_idx[i]], dinv[i] = 1 / diag[i].
 * K15, block Jacobi on a strip: the level is a rank's DIAGONAL BLOCK in local
 * numbering -- rowptr / diag_idx / a00 / a11 shifted to the block's first
 * nonzero, cols in local numbering with the couplings that leave the block
 * pointing at their own row and marked keep = 0; `ends` of a dof whose partner
 * vertex lies outside names the dummy coarse row coarse.n (work carries a zero
 * there), the restriction lists hold the block's own dofs only. */
int flow_pmg_pack(int n, int nnz, const int* rowptr, const int* diag_idx,
                  const double* a00, const double* a11,
                  const unsigned char* keep, void* vals, float* diag,
                  float* dinv, void* stream);
/* The one-plane stream of a level (flow_pmg_level.packed) from the same two
 * blocks: idrows[a n + i] = 1 where row i of block a has no off-diagonal entry
 * (a Dirichlet dof of component a: found here, the caller passes no mask);
 * entry (i, j) = mean of a00 / a11 over the components in which neither row i
 * nor column j is such a dof, divided by the mean diagonal over the free
 * components of row i, rounded to fp16 and packed with the column offset from
 * cbase[tile] (flow_pmg_cols16 of the same row blocks, no overflow); diag /
 * dinv as flow_pmg_pack's with the free components' entries replaced by the
 * mean.  keep as in flow_pmg_pack. */
int flow_pmg_pack1(int n, int nnz, int nblocks, const int* rowblocks,
                   const int* rowptr, const int* cols, const int* diag_idx,
                   const double* a00, const double* a11,
                   const unsigned char* keep, const int* cbase,
                   unsigned char* idrows, void* packed, float* diag,
                   float* dinv, void* stream);
/* cols16 / cbase of a level's pattern; *overflow_dev (zeroed by the caller) is
 * set when an offset does not fit in 16 bits */
int flow_pmg_cols16(int nblocks, const int* rowblocks, const int* rowptr,
                    const int* cols, int* cbase, void* cols16,
                    int* overflow_dev, void* stream);
/* spectral radius of D^-1 A of a level by `iterations` (>= 2) steps of the
 * power method (normalised on the device: one read-back, at the end).  work:
 * 6*n floats, dwork: FLOW_REDUCE_WORK doubles.  start (optional, 2*n floats,
 * zero before the first call): the iterate the previous call left there is
 * where this one starts (a rebuild of the same level a few time steps later
 * needs a few iterations, not 32) and leaves its own. */
int flow_pmg_lambda_max(const flow_pmg_level* level, int iterations, float* work,
                        double* dwork, float* start, double* result_host,
                        void* stream);
/* z = M^-1 r (r, z: 2*fine.n doubles, component-blocked) */
int flow_pmg_apply(const flow_pmg* pmg, const double* r, double* z, void* stream);

/* HOST routine (setup, no GPU needed): ALGEBRAIC aggregation of the rows of a
 * scalar CSR matrix for the smoothed-aggregation hierarchy (flow_mg below;
 * `hypre_amg` of pressure_correction.py:331, 414 needs no coordinates either).
 * j is strongly coupled to i when |a_ij| >= theta sqrt(|a_ii a_jj|), i != j.
 * Three passes in row order (Vanek, Mandel, Brezina 1996): (1) a free row none
 * of whose strong neighbours is aggregated yet founds an aggregate with all of
 * them; (2) every remaining row joins the aggregate of its strongest coupled
 * aggregated neighbour; (3) what is left forms aggregates with its own
 * unaggregated strong neighbours (isolated rows: singletons).  free[i] == 0
 * (Dirichlet rows; NULL: all free): agg[i] = -1.  agg: n ints out,
 * *naggregates the count.  Deterministic. */
int flow_aggregate_host(int n, const int* rowptr, const int* cols,
                        const double* vals, double theta,
                        const unsigned char* free_rows, int* agg,
                        int* naggregates);

/* ---- K19: two-level cycle with ILU(0) smoothing ----------------------------
 * (the stand-in for the sparse LU of the Newton solve, pressure_correction.py:
 * 224-254, and of the heat solve, heat.py:117-121, WHERE THE CHEBYSHEV CYCLE OF
 * K17 IS REJECTED by its acceptance test: cell Peclet numbers beyond ~3 at
 * CFL-sized steps put the spectrum of D^-1 A off the real axis; a polynomial
 * smoother amplifies there, a multicolour ILU(0) sweep does not.)  The levels are
 * those of K17 -- the diagonal block(s) of the assembled operator on the P2
 * pattern and the P1 discretisation of the same operator on the same mesh, the
 * same transfer tables -- with one ILU(0) application (K11) as the smoother on
 * the fine level and `coarse_sweeps` of them as the treatment of the P1 level.
 * One application z = M^-1 r (all vectors fp64, component-blocked):
 *   x = pre ? ILU_f^-1 r : 0 ;  t = r - A_f x ;  rc = P^T (rscale .* t) ;
 *   xc = ILU_c^-1 rc ; (coarse_sweeps - 1) x [ xc += ILU_c^-1 (rc - A_c xc) ] ;
 *   x += P xc ;  post: x += ILU_f^-1 (r - A_f x) ;  z = x.
 * Measured (tools/smoother_lab.py, flexible GMRES applications to 1e-8):
 * structured channel at cell Peclet 3.5 27 -> 13, graded unstructured channel
 * 48 -> 17, heat system at cell CFL 6 / 14 / 40: 56 / 80 / 146 -> 19 / 23 / 35
 * (two coarse sweeps: 14 / 17 / 28).
 * Dirichlet dofs: identity rows of both operators (ILU: z = r there); the
 * prolongation leaves them alone (bc_fine), the restricted residual is zeroed
 * on those of the P1 level (bc_coarse).  rscale (optional): the residual is
 * multiplied row by row before it is restricted -- the heat solve works on the
 * ROW-SCALED system (ILU(0) is invariant under a row scaling, the rediscretised
 * P1 level is not).  Reached through flow_ilu.cycle by every solver that takes
 * a flow_ilu. */
typedef struct {
  const flow_ilu* fine;          /* ILU(0) of the fine diagonal block(s); cycle == NULL */
  const flow_ilu* coarse;        /* ILU(0) of the P1 level; same nblocks; cycle == NULL */
  const flow_operator* fine_op;  /* A_f, size nblocks * fine->plan->n (kind 0, 1, 2 or 3) */
  const flow_operator* coarse_op;/* A_c (needed for coarse_sweeps > 1) */
  int pre, post;                 /* 0 | 1, not both 0 */
  int coarse_sweeps;             /* 1 .. 8 */
  const int* ends;               /* fine n int2: the coarse rows of a fine dof  } as in */
  const int* rptr;               /* coarse n + 1: restriction lists ...         } flow_pmg */
  const int* rsrc;               /* ... of fine dofs, the vertex's own dof first */
  const unsigned char* bc_fine;  /* nblocks * n bytes or NULL */
  const unsigned char* bc_coarse;/* nblocks * n1 bytes or NULL */
  const double* rscale;          /* nblocks * n or NULL */
  double* work;                  /* nblocks * (4 n + 5 n1) doubles, 16-B aligned */
} flow_tl;
/* z = M^-1 r (r, z: nblocks * n doubles, r != z) */
int flow_tl_apply(const flow_tl* tl, const double* r, double* z, void* stream);

/* ---- K18: mass-matrix solves by mixed-precision defect correction ---------
 * The velocity correction (pressure_correction.py:436-465: `solve(a3 == L3,
 * u1, bcs)`, CG + hypre_amg, relative tolerance tol) and the callers' L2
 * projections (tests/test_karman_vortex_street.py:262-267) are solves with a
 * finite-element MASS matrix.  Its Jacobi-scaled spectrum is known a priori on
 * any mesh (Wathen 1987: the extreme eigenvalues of D^-1 M lie between those
 * of the scaled element matrices -- P1 triangles [1/2, 2], P2 triangles
 * [0.3924, 2.0598]), so a fixed Chebyshev polynomial of D^-1 M is an
 * approximate inverse B with a KNOWN contraction |I - B M| <= eps_k =
 * 2 s^k / (1 + s^2k), s = 0.39 for P2 (k = 6 steps: 7e-3).  The solve is the
 * defect correction
 *     r = b - M x  (fp64, the CSR-stream SpMV of the operator, with the
 *                   residual scaled by D^-1 and rounded to fp32 in its epilogue)
 *     z = B r      (k Chebyshev steps = k - 1 products with a copy of D^-1 M
 *                   rounded to fp16 -- ONE value plane, 6 B per nonzero with
 *                   the index, also when it serves both velocity components --,
 *                   vectors fp32, components interleaved)
 *     x += z       (fp64, in the last product's epilogue)
 * No dot product steers the iteration; one launch per product.  Stopping test:
 * z_k = B r_k IS the preconditioned residual of PETSc's KSPCG test with a
 * preconditioner as strong as the reference's AMG (B ~ M^-1), and B b ~ x:
 *     contraction * |z_k| <= max(rtol |x_k + z_k|, atol)
 * -- since B r_{k+1} = (I - B M) z_k, the iterate x_{k+1} = x_k + z_k then
 * satisfies |B r_{k+1}| <= rtol |B b| (contraction = the bound on |I - B M|
 * the caller vouches for; 1: test |z_k| itself, one more iteration).  Decided
 * on the device like the Krylov drivers below (sticky flag, exact count).
 * A: kind 0 (scalar) or kind 4 (one plane for both components, identity rows
 * by mask; x should carry b on those rows on entry, else it converges there
 * like everywhere).  dinv: op-size doubles (1 on masked rows).
 * vals16: nnz halfs (D^-1 M)_ij (flow_mass_pack), 16-byte aligned, readable
 * three entries past nnz like A->cols (quads of nonzeros per lane);
 * rowblocks16: CSR-stream row blocks of <= FLOW_PMG_NNZ_PER_BLOCK nonzeros.
 * work16: 5 * ncomp * n floats, 16-byte aligned. */
typedef struct {
  const flow_operator* A;
  const double* dinv;
  int nblocks16;
  const int* rowblocks16;
  const void* vals16;
  double lam_min, lam_max;     /* Chebyshev interval of D^-1 M */
  int steps;                   /* Chebyshev steps per defect correction, 3..16 */
  double contraction;          /* bound on |I - B M| in (0, 1] */
  float* work16;
  /* the PACKED stream (used instead of vals16 + A->cols when not NULL): one
   * 32-bit word per nonzero, fp16 value | (column - cbase16[tile]) << 16 --
   * 4 B per nonzero and one 16-byte load per quad of nonzeros; nnz words,
   * 16-byte aligned, readable three past nnz.  Needs every tile's columns to
   * span < 65536 (flow_mass_pack16 reports otherwise). */
  const void* packed16;
  const int* cbase16;          /* nblocks16: lowest column of each tile */
  int work16_rows;             /* rows the fp32 work vectors cover (0: A->n); K15:
                                  the rank's window, flow_shard_mass_solve */
} flow_mass;
/* vals16[k] = half(vals[k] / vals[diag_idx[row(k)]]) */
int flow_mass_pack(int n, const int* rowptr, const int* diag_idx,
                   const double* vals, void* vals16, void* stream);
/* the packed stream of the same matrix over the row blocks rowblocks16;
 * *overflow_dev (an int the caller zeroes) is set to 1 when a column offset
 * does not fit in 16 bits: keep the plain stream then */
int flow_mass_pack16(int n, int nblocks16, const int* rowblocks16,
                     const int* rowptr, const int* cols, const int* diag_idx,
                     const double* vals, int* cbase16, void* packed16,
                     int* overflow_dev, void* stream);
/* x holds the initial guess.  maxit / *iters_host count defect corrections;
 * first_check: as flow_cg_solve (then one at a time).  *resid_host = |z| of
 * the last correction.  work: FLOW_REDUCE_WORK + 2 * nblocks16 doubles. */
int flow_mass_solve(const flow_mass* M, const double* b, double* x, double rtol,
                    double atol, int maxit, int first_check, double* work,
                    size_t work_len, int* iters_host, double* resid_host,
                    void* stream);

/* The same for a solve whose start xbase is close, in INCREMENT form:
 * M (x - xbase) = g with the defect g = b - M xbase GIVEN by the caller -- the
 * velocity correction's is -dt/rho (grad phi, v), assembled without the
 * (ui, v) term (flow_assemble_correction_rhs, flag bit 1) -- and the increment
 * iterated from zero in a vector of its own: the first correction needs no
 * product with M, and fp64 only has to resolve the increment.  delta0 (or
 * NULL): a start vector for the increment (a time loop's previous increments,
 * extrapolated; the first defect is then g - M delta0).  The stopping
 * test is the one above with |x| = |xbase + increment|.  x = xbase + increment
 * on return (x may be xbase; the rows A masks as identity rows need g = 0).
 * work: FLOW_REDUCE_WORK + 2 * nblocks16 + 1 + op-size doubles. */
int flow_mass_solve_increment(const flow_mass* M, const double* g,
                              const double* xbase, const double* delta0,
                              double* x, double rtol,
                              double atol, int maxit, int first_check,
                              double* work, size_t work_len, int* iters_host,
                              double* resid_host, void* stream);

/* ---- K12: Krylov drivers -------------------------------------------------
 * Device-resident loops.  Convergence is decided ON THE DEVICE: the kernel
 * that computes the solver scalars compares the residual norm with the target
 * and freezes the solution at the first iterate that passes (everything
 * enqueued behind it returns at once), so *iters_host is the exact count and
 * the solution never moves past the accepted iterate, however late the host
 * looks: it reads the state every check_every iterations -- the first time
 * after first_check iterations when that is > 0 (a time loop knows how many the
 * previous step needed; every read-back drains the stream).
 * Stopping test: CG tests the PRECONDITIONED residual like PETSc's KSPCG (the
 * default the reference's `solve` runs with, pressure_correction.py:326-339,
 * 419-432, 451-464): ||B r||_2 <= max(rtol*||B b||_2, atol), B = the
 * preconditioner (Jacobi, two-level, V-cycle; identity without one) -- with
 * Dirichlet rows carrying boundary VALUES in b next to O(h^2) interior
 * entries, the unpreconditioned norm would let the interior equations off at
 * rtol*|g|/h^2.  *resid_host is that norm.  BiCGStab and GMRES are right-
 * preconditioned: ||r||_2 <= max(rtol*||b||_2, atol).  Return
 * FLOW_NOT_CONVERGED after maxit iterations (dolfin raises RuntimeError:
 * 'error_on_nonconvergence', pressure_correction.py:337,424,462).
 * dinv may be NULL (no preconditioner).  x holds the initial guess.
 * coarse may be NULL (Jacobi only); mg != NULL selects the multigrid V-cycle
 * instead (coarse must then be NULL); ilu may be NULL (Jacobi), else it replaces
 * dinv as the (right) preconditioner of BiCGStab.
 * work (16-byte aligned): FLOW_REDUCE_WORK + 5*N + B + 2 [+ 2*coarse->lda]
 * [+ 2*mg->Ps[0].nblocks] doubles (cg; B = nblocks, twice that for kind 1: the
 * SpMV leaves its z.Az partials there, the V-cycle's last kernel its r.z and
 * z.z partials), FLOW_REDUCE_WORK + 7*N [+ n] (bicgstab); N = operator size. */
int flow_cg_solve(const flow_operator* A, const double* dinv,
                  const flow_coarse* coarse, const flow_mg* mg,
                  const double* b, double* x,
                  double rtol, double atol, int maxit, int check_every,
                  int first_check, double* work, size_t work_len,
                  int* iters_host, double* resid_host, void* stream);
/* flow_cg_solve from a GUARDED start vector (the start vectors a time loop
 * extrapolates from its previous calls, flow_amd/navier_stokes/
 * start_vectors.py; the reference starts every solve from a fresh, zero
 * Function, pressure_correction.py:313).  A start that leaves a larger
 * preconditioned residual than x = 0 would, ||B(b - A x)|| > ||B b|| --
 * decided on the device by the kernel that forms the first iteration's scalars,
 * from numbers it forms anyway; everything enqueued behind the verdict returns
 * at once, x still holds the start -- is dropped for x_fallback (guarded the
 * same way; NULL: none; must not alias x) and that for zero.
 * *starts_dropped_host: 0, 1 or 2.  Everything else as flow_cg_solve. */
int flow_cg_solve_guarded(const flow_operator* A, const double* dinv,
                          const flow_coarse* coarse, const flow_mg* mg,
                          const double* b, double* x, const double* x_fallback,
                          double rtol, double atol, int maxit, int check_every,
                          int first_check, double* work, size_t work_len,
                          int* iters_host, double* resid_host,
                          int* starts_dropped_host, void* stream);
int flow_bicgstab_solve(const flow_operator* A, const double* dinv,
                        const flow_ilu* ilu, const double* b, double* x,
                        double rtol, double atol, int maxit, int check_every,
                        int first_check, double* work, size_t work_len,
                        int* iters_host, double* resid_host, void* stream);
/* GMRES(restart), right-preconditioned with the p-multigrid / Chebyshev cycle
 * (pmg != NULL), ILU(0) (ilu != NULL), Jacobi (dinv != NULL) or nothing -- at
 * most one of pmg / ilu: the Newton systems of the tentative velocity
 * (pressure_correction.py:224-254) -- one operator + preconditioner application
 * per iteration, ~15 % fewer of them than BiCGStab needs there.  Classical
 * Gram-Schmidt with one reduction per iteration; the Hessenberg matrix, the
 * small least-squares problem and the stopping test live on the device (one
 * workgroup behind the dot products), so the Arnoldi steps are enqueued
 * without the host in between: expected_its of them (what the caller expects
 * the solve to need -- the count of the previous solve in a time loop; 0:
 * unknown) before the host looks for the first time, then one at a time.  The
 * accepted iterate does not depend on it: everything enqueued behind it returns
 * at once.  A cycle stops on the least-squares residual estimate.  verify != 0:
 * the iterate is then checked with the true residual b - A x (one more
 * operator application and read-back, not counted): accepted within a factor
 * 10 of the target, with the TRUE norm in *resid_host, continued otherwise --
 * for solves nothing else checks (flow/heat.py:117-121); verify = 0: the
 * estimate is trusted (*resid_host = the estimate) -- the Newton systems, whose
 * outer iteration recomputes the nonlinear residual anyway.  `iters_host` =
 * operator applications of the Arnoldi steps.
 * x_is_zero != 0: the caller guarantees x = 0 on entry (Newton increments), the
 * initial residual is then b without an operator application.
 * work: FLOW_REDUCE_WORK + (2*restart + 2)*N + FLOW_GMRES_PARTIALS +
 * FLOW_GMRES_STATE doubles (the preconditioned basis vectors are kept: no extra
 * preconditioner application for the solution update). */
#define FLOW_GMRES_MAX_RESTART 30
#define FLOW_GMRES_PARTIALS ((FLOW_GMRES_MAX_RESTART + 2) * 1024)
#define FLOW_GMRES_STATE 1280
int flow_gmres_solve(const flow_operator* A, const double* dinv,
                     const flow_ilu* ilu, const flow_pmg* pmg, const double* b,
                     double* x,
                     double rtol, double atol, int maxit, int restart,
                     int x_is_zero, int expected_its, int verify, double* work,
                     size_t work_len, int* iters_host, double* resid_host,
                     void* stream);

/* ---- K15: domain decomposition over the GPUs of one node -------------------
 * (nothing in the reference: DOLFIN/PETSc would do this implicitly under
 * mpirun).  One process per GPU.  The mesh is cut into strips along the channel:
 * with the x-major numbering every scalar space splits into contiguous row
 * blocks, rank g OWNS the rows [r0, r1) and needs, for anything its rows are
 * coupled to (matrix columns, dofs of incident cells), the GHOST rows [e0, r0)
 * and [r1, e1) -- owned by its left and right neighbour.
 *
 * The library needs ONE communication primitive, handed in as a callback:
 *   allreduce(user, count): sum the first `count` doubles of comm->buf over the
 *   ranks, in stream order (kernels enqueued before it have written buf, kernels
 *   enqueued after it see the sums).
 * The host binds it to torch.distributed.all_reduce on RCCL (flow_amd/
 * parallel.py).  Everything travels in that buffer: dot products, the partial
 * coarse residual of the multigrid cycle, and the halos -- every rank writes
 * its boundary rows into its own slots and zeros into all others, so the sum
 * is the concatenation (bitwise the owners' values: x + 0).  One kind of
 * collective, a fixed reduction order, identical results on every rank.
 *
 * Vectors inside the sharded solvers are EXT-COMPACT: ncomp * (e1 - e0)
 * doubles, entry (a, row) at a*(e1-e0) + row - e0.  Kernels that index by
 * global row get the base pointer shifted by -e0 and the component stride
 * e1 - e0; fields handed in by the caller (b, x, dinv, ...) are global-length
 * (stride n), valid on the rank's owned + ghost rows. */
typedef int (*flow_allreduce_fn)(void* user, int count);
/* Halos from neighbour to neighbour instead of through the all-reduce (round
 * 6; optional).  Every rank owns a peer block its two neighbours have mapped
 * with hipIpcOpenMemHandle (xGMI peers on a node; processes sharing one device
 * in a rehearsal): FLOW_PEER_FLAGS 64-bit words, then two landing buffers of
 * land_cap doubles.  One exchange is then a PUSH launch (my send slots of the
 * packed buffer -> the neighbour's landing buffer, then its `arrived` word), the
 * all-reduce of what is really summed (nothing for a pure halo), and a PULL
 * launch (wait for my `arrived` words, landing buffer -> my packed buffer,
 * acknowledge in the neighbour's `consumed` word: it may reuse the buffer two
 * exchanges on).  Sequence numbers count the exchanges of the communicator:
 * every rank issues the same sequence.  Waits are bounded (spin_limit polls):
 * a neighbour that never arrives sets word 4 of the waiting rank's block
 * ((seq << 8) | what it waited for) and the kernel goes on -- the device
 * cannot hang; flow_peer_status reads that word. */
#define FLOW_PEER_FLAGS 16
typedef struct {
  unsigned long long* flags;       /* this rank's block: [arrived L, arrived R,
                                      consumed L, consumed R, error, ...] */
  double* land;                    /* this rank's 2 * land_cap doubles */
  int land_cap;
  int spin_limit;                  /* polls before a wait gives up */
  unsigned long long* nb_flags[2]; /* the neighbours' blocks as mapped HERE ... */
  double* nb_land[2];              /* ... and their landing buffers; NULL: no
                                      neighbour on that side (0 left, 1 right) */
  unsigned long long* seq_host;    /* HOST, 5 words kept by the library: the
                                      counter of this communicator's exchanges,
                                      then per side and landing buffer the
                                      exchange of this rank's last push into it */
} flow_peer;
typedef struct {
  int rank, world;
  double* buf;               /* exchange buffer (device), `capacity` doubles */
  int capacity;
  flow_allreduce_fn allreduce;
  void* user;
  const flow_peer* peer;     /* NULL: halos travel in the all-reduce */
} flow_comm;
/* A peer block of FLOW_PEER_FLAGS words + 2 * land_cap doubles, zeroed, and its
 * 64-byte IPC handle (handle_out); the neighbours map it with flow_peer_open
 * and unmap it with flow_peer_close before the owner frees it. */
int flow_peer_alloc(int land_cap, void** base_out, char* handle_out);
int flow_peer_open(const char* handle, void** base_out);
int flow_peer_close(void* mapped_base);
int flow_peer_free(void* base);
/* the error word of this rank's block (0: every wait so far was served) */
int flow_peer_status(const flow_peer* peer, unsigned long long* error_host,
                     void* stream);

/* The same primitive issued by the library itself: ncclAllReduce on `stream`,
 * on a communicator of its own (flow_amd/csrc/rccl_direct.hip).  The RCCL
 * shared object is dlopen'ed by path (the instance the host already loaded),
 * the 128-byte unique id of rank 0 is distributed by the host.  Set
 * comm->allreduce = flow_rccl_allreduce and comm->user = &binding. */
#define FLOW_RCCL_ID_BYTES 128
typedef struct {
  void* comm;                /* from flow_rccl_comm_create */
  double* buf;               /* = flow_comm.buf */
  void* stream;              /* hipStream_t of the solvers */
} flow_rccl_binding;
int flow_rccl_load(const char* librccl_path);
int flow_rccl_unique_id(char* id_host);                      /* 128 bytes out */
int flow_rccl_comm_create(const char* id_host, int rank, int world,
                          void** comm_out);                  /* collective */
int flow_rccl_comm_destroy(void* comm);
int flow_rccl_allreduce(void* user, int count);              /* flow_allreduce_fn */

/* the rows of one scalar space as rank `comm->rank` sees them; index 0 = left
 * neighbour, 1 = right neighbour, len 0 = none */
typedef struct {
  int n;                     /* global rows */
  int r0, r1;                /* owned */
  int e0, e1;                /* owned + ghost: e0 <= r0 < r1 <= e1 */
  int nhalo;                 /* halo slots of ONE component, all ranks together */
  int send_row[2], send_len[2], send_slot[2];  /* x[row..+len) -> slots */
  int recv_row[2], recv_len[2], recv_slot[2];  /* slots -> x[row..+len) */
} flow_rows;

/* make the ghost rows of x current (ncomp components, x[a*stride + row] with
 * GLOBAL row: a global-length field has stride n; for an ext-compact vector v
 * pass x = v - e0, stride = e1 - e0).  One collective. */
int flow_shard_halo(const flow_comm* comm, const flow_rows* rows, int ncomp,
                    double* x, int stride, void* stream);
/* kind 0: sum over the owned rows of x_a[row] * y_a[row], summed over the
 * ranks; kind 1: max |x| over the owned rows and the ranks (y unused).
 * work: FLOW_REDUCE_WORK doubles.  One collective + one read-back. */
int flow_shard_reduce_host(const flow_comm* comm, const flow_rows* rows,
                           int ncomp, const double* x, const double* y,
                           int stride, int kind, double* work,
                           double* result_host, void* stream);

/* CG + Jacobi on the strips: the mass-matrix solves of the velocity correction
 * (operator kind 4: both components, pressure_correction.py:451-464) and of the
 * callers' step-size projection (kind 0).  A carries the CSR-stream row blocks
 * of the OWNED rows only; b, x, dinv are global-length fields (b valid on the
 * owned rows, x on owned rows -- its ghosts are made current on entry and on
 * exit).  Chronopoulos-Gear recurrences on the owned AND ghost rows (pointwise
 * given w = A z there), so ONE collective per iteration carries the three dot
 * products and the halo of w; stopping test, device-side convergence flag and
 * the meaning of first_check / check_every as flow_cg_solve.  Every rank runs
 * the same number of iterations (the sums are bitwise identical).
 * start_rejected_host != NULL: the start vector is guarded as in
 * flow_cg_solve_guarded -- ||B(b - A x)|| > ||B b|| (sums over all ranks: the
 * same verdict everywhere) leaves x untouched, sets *start_rejected_host = 1
 * and returns FLOW_OK: the caller swaps in its fallback and calls again.
 * work (16-B aligned): FLOW_REDUCE_WORK + 10 * ncomp * (e1 - e0) + A->nblocks
 * + 2 doubles, ncomp = 1 (kind 0) or 2 (kind 4). */
int flow_shard_cg_solve(const flow_comm* comm, const flow_rows* rows,
                        const flow_operator* A, const double* dinv,
                        const double* b, double* x, double rtol, double atol,
                        int maxit, int check_every, int first_check,
                        double* work, size_t work_len, int* iters_host,
                        double* resid_host, int* start_rejected_host,
                        void* stream);

/* The pressure solve on the strips: CG preconditioned with the SAME smoothed-
 * aggregation V(1,1) cycle as flow_cg_solve (same iteration counts).  The
 * finest level is row-sharded: Ah0 / Ps0 are mg->Ah[0] / mg->Ps[0] with the row
 * blocks of the owned rows, Rg is mg->R[0] restricted to the owned COLUMNS
 * (column index minus r0; all coarse rows): every rank restricts its own part
 * of the fine residual, the partial coarse residuals are summed by the
 * collective, and the levels below (a tenth of the rows, latency-bound on one
 * GPU already) run replicated on every rank.  Three collectives per iteration:
 * [dots + halo of w], [coarse residual], [halo of z] -- or, when the hierarchy
 * carries the two-launch form (mg->C) and Cg / up_rowblocks0 are given, TWO:
 * the coarse residual rc = C r is then carried by the recurrences of CG itself,
 *     rc_s = rc_w + beta rc_s ;  rc_r -= alpha rc_s ,   rc_w = C w = sum over
 *     the ranks of Cg w_owned,
 * whose only collective part, rc_w, rides with the dots and the halo of w (C is
 * restricted by COLUMNS: a rank needs no ghost rows for its share).  Every
 * eighth iteration rc_r and rc_s are recomputed from r and s themselves (one
 * more collective then): carried by recurrence alone they drift away from the
 * rank-local vectors and CG stagnates short of a tight target.
 * work: FLOW_REDUCE_WORK + 11 * (e1 - e0) + A->nblocks +
 * 2 * max(Ps0.nblocks, up_nblocks0) + 2 [+ 3 * Cg.n]. */
typedef struct {
  const flow_mg* mg;
  flow_operator Ah0, Ps0, Rg;
  flow_operator Cg;             /* mg->C[0] restricted to the owned columns
                                   (column index minus r0); rowptr NULL: the
                                   three-collective form */
  const int* up_rowblocks0;     /* mg->up_rowblocks[0] for the owned rows */
  int up_nblocks0;
  /* ONE collective per iteration (two-collective form only): z_hi > z_lo --
   * up_rowblocks0 then covers the rows [z_lo, z_hi) = the owned rows plus ONE
   * ghost layer, and the flow_rows handed to the solver reach TWO layers out:
   * the halo of w (in the collective that carries the dots) is two layers
   * deep, so r is current two layers out, so the up-sweep can form z on the
   * first ghost layer itself -- every rank recomputing those few rows instead
   * of asking for them -- and w = A z needs no halo of z.  x comes back on
   * [z_lo, z_hi). */
  int z_lo, z_hi;
} flow_mg_shard;
int flow_shard_mgcg_solve(const flow_comm* comm, const flow_rows* rows,
                          const flow_operator* A, const double* dinv,
                          const flow_mg_shard* mgs, const double* b, double* x,
                          double rtol, double atol, int maxit, int check_every,
                          int first_check, double* work, size_t work_len,
                          int* iters_host, double* resid_host,
                          int* start_rejected_host, void* stream);

/* The mass solver of K18 on the strips.  A correction's polynomial reaches
 * steps - 1 matrix hops: with the scaled defect known on a ghost zone `steps`
 * vertex columns deep the rank computes product j on the rows within
 * steps - 1 - j hops of its own (rowblocks16[j]: CSR-stream row blocks of that
 * row range, <= FLOW_PMG_NNZ_PER_BLOCK nonzeros each, global row numbers) and
 * ends with x advanced on its own rows and its first ghost layer (the rows
 * [row_lo_last, row_hi_last) of the last product) -- no communication inside a
 * correction, ONE collective per correction: the deep halo of the defect with
 * the norms of the correction before riding along (so a solve that needs m
 * corrections issues m + 1).  `rows`: the owned rows with the DEEP ghost range
 * [e0, e1) and its halo slots; M->A: the operator with the row blocks of the
 * owned rows; M->work16: 5 * ncomp * (e1 - e0) floats (work16_rows = e1 - e0);
 * the plain fp16 stream M->vals16 is used (the packed one is tiled once).
 * xbase != NULL: the increment form (b = the defect of xbase; delta0 or NULL
 * as flow_mass_solve_increment; x = xbase + increment on the rows of the last
 * product); xbase == NULL: x holds the start, valid on own + first ghost rows.
 * work: FLOW_REDUCE_WORK + 2 * nblocks16[last] + 1 [+ op size] doubles. */
typedef struct {
  int nlevels;                    /* = steps - 1 */
  const int* rowblocks16[16];
  int nblocks16[16];
  int row_lo_last, row_hi_last;
} flow_mass_strips;
int flow_shard_mass_solve(const flow_comm* comm, const flow_rows* rows,
                          const flow_mass* M, const flow_mass_strips* levels,
                          const double* b, const double* xbase,
                          const double* delta0, double* x, double rtol,
                          double atol, int maxit, int first_check, double* work,
                          size_t work_len, int* iters_host, double* resid_host,
                          void* stream);

/* GMRES(restart) on the strips: for the Newton systems -- the operator is the
 * matrix-free Jacobian action (kind 3) whose flow_mesh / flow_space carry the
 * rank's cell and row ranges (below) -- and for scalar systems (kind 0 with the
 * row blocks of the owned rows, one component: the heat system,
 * flow/heat.py:103-122; verify as flow_gmres_solve).  The preconditioner is
 * block Jacobi, no
 * communication: EITHER ilu, the factors of the rank's own diagonal block (plan
 * over the owned rows in local numbering), OR pmg, the two-level cycle of K17
 * on that block (levels in local numbering, flow_pmg_pack).  The Krylov vectors hold the owned rows only
 * (2 * (r1 - r0) doubles); per iteration one halo collective (the operator's
 * input) and one for the dot products, whose sums the device-side step of
 * flow_gmres_solve takes straight from the exchange buffer: no read-back per
 * iteration here either (expected_its as there; every rank must pass the same
 * value -- it decides how many collectives are enqueued).  b, x: global-length
 * fields (owned rows).  work: FLOW_REDUCE_WORK + (2*restart + 4) * 2*(r1-r0)
 * + 2*(e1-e0) + FLOW_GMRES_PARTIALS + FLOW_GMRES_STATE doubles. */
int flow_shard_gmres_solve(const flow_comm* comm, const flow_rows* rows,
                           const flow_operator* A, const flow_ilu* ilu,
                           const flow_pmg* pmg, const double* b, double* x,
                           double rtol, double atol,
                           int maxit, int restart, int x_is_zero,
                           int expected_its, int verify, double* work,
                           size_t work_len, int* iters_host, double* resid_host,
                           void* stream);

/* ---- assembly ------------------------------------------------------------
 * Two-phase, atomic-free: a cell kernel writes local tensors to `scratch`
 * ([entry][cell]); a gather kernel sums, per CSR nonzero / per dof, the
 * contributions listed in the contribution maps built once per mesh on the
 * host (flow_amd/fem/space.py). */
typedef struct {
  int nc;                  /* cells */
  const double* xy;        /* (2,3,nc): vertex coordinates */
  int c0, c1;              /* K15: the cell kernels run over [c0, c1) only
                              (c1 = 0: all cells) -- a rank's cells are those
                              that touch its owned rows */
} flow_mesh;

typedef struct {
  int deg;                 /* 1 | 2 */
  int n;                   /* scalar dofs */
  int nnz;
  const int* cell_dofs;    /* (nloc, nc) */
  const int* cptr;         /* nnz+1  matrix contribution map */
  const int* csrc;
  const int* vptr;         /* n+1    vector contribution map */
  const int* vsrc;
  int r0, r1;              /* K15: the gathers fill the rows [r0, r1) only
                              (r1 = 0: all rows) ... */
  int nnz0, nnz1;          /* ... and the nonzeros [nnz0, nnz1) = [rowptr[r0],
                              rowptr[r1]) of matrix planes */
} flow_space;

/* per-cell P_k lattice values of a coefficient (Constant / Expression(degree=k)
 * / Function), `dim` components: values[(a*nl + l)*nc_eff + c*cell_stride];
 * G = (nl x nloc_test) reference matrix int psi_l phi_i. */
typedef struct {
  int nl;
  int cell_stride;         /* 0: spatially constant, 1: per cell */
  const double* values;
  const double* G;
} flow_coef;

/* K1: a2 = dot(grad(p), grad(q))*dx            (pressure_correction.py:317)
 * K3: a3 = inner(u2, v)*dx per component       (pressure_correction.py:442)
 * kind 0: stiffness, 1: mass, 2: vertex-quadrature lumped mass (heat.py:39-45).
 * scratch: nloc^2 * nc doubles. */
int flow_assemble_scalar_matrix(int kind, const flow_mesh* mesh,
                                const flow_space* V, double* scratch,
                                double* vals, void* stream);

/* K2: L2 (pressure_correction.py:318-323):
 *   b_i = -alpha*rho/dt (div u, q_i) + (grad p0, grad q_i)
 *         [- mu (grad div u, grad q_i)  if rotational].
 * W: velocity scalar space (deg 1|2), P: P1.  scratch: 3*nc. */
int flow_assemble_pressure_rhs(const flow_mesh* mesh, const flow_space* W,
                               const flow_space* P, const double* u,
                               const double* p0, double alpha_rho_dt, double mu,
                               int rotational, double* scratch, double* b,
                               void* stream);

/* K4: L3 (pressure_correction.py:444-449):
 *   b_(a,i) = (u_a, v_i) - dt/rho (d_a phi, v_i),
 *   phi = p1 - p0 [+ mu div u  if rotational & 1].   scratch: 2*nloc*nc.
 * rotational & 2: WITHOUT the (u_a, v_i) term -- the defect b - M u of the
 * start u = ui, the right-hand side of the correction's increment
 * (flow_mass_solve_increment). */
int flow_assemble_correction_rhs(const flow_mesh* mesh, const flow_space* W,
                                 const flow_space* P, const double* u,
                                 const double* p1, const double* p0,
                                 double dt_rho, double mu, int rotational,
                                 double* scratch, double* b, void* stream);

/* K5+K6: F1 and J = derivative(F1, ui) (pressure_correction.py:169-202) with
 * the integrand of _rhs_weak (:135-144), exterior-facet terms included:
 *   F = (ui - u0, v) - dt/rho [theta_i R(ui; f1) + theta_e R(u0; f0)],
 *   J = dF/dui.
 * bfmask[c]: bit i set iff local facet i of cell c is a boundary facet.
 * F (2n) and/or Jvals (4 planes, j_plane_stride >= nnz apart) may be NULL to
 * skip.
 * scratch: max(2*nloc, 4*nloc^2) * nc doubles. */
typedef struct {
  double dt, rho, mu, theta_i, theta_e;
} flow_ns_params;
int flow_assemble_momentum(const flow_mesh* mesh, const flow_space* W,
                           const flow_space* P, const int* bfmask,
                           const double* ui, const double* u0,
                           const double* p0, const flow_coef* f0,
                           const flow_coef* f1, const flow_ns_params* prm,
                           double* scratch, double* F, double* Jvals,
                           size_t j_plane_stride, void* stream);

/* Matrix-free action of that Jacobian, out = J(ui) v = dF1/dui [v] (the
 * directional derivative of the form, evaluated cell by cell like the residual;
 * `derivative(F1, ui)` applied instead of assembled, pressure_correction.py:202):
 * costs what the assembled 2x2-block SpMV costs and makes the per-Newton-step
 * Jacobian assembly unnecessary -- the assembled diagonal blocks are then only
 * needed when the (lagged) ILU(0) is refactored.  Rows of the nbc Dirichlet
 * dofs are identity rows (out[d] = v[d]), as flow_bc_identity_rows makes them.
 * scratch: 2*nloc*nc doubles, private to this operator while a solve runs.
 * A flow_operator of kind 3 carries a pointer to this struct in `matfree`. */
typedef struct {
  const flow_mesh* mesh;
  const flow_space* W;       /* velocity space, needs vptr/vsrc */
  const int* bfmask;         /* nc: exterior-facet bits (as flow_assemble_momentum) */
  const double* ui;          /* 2*W->n: linearisation point */
  flow_ns_params prm;
  double* scratch;
  int nbc;
  const int* bc_dofs;        /* nbc Dirichlet dofs (component-blocked numbering) */
  const unsigned char* bc_mask;   /* 2n bytes, 1 on those dofs, or NULL: with it
                                     the gather writes the identity rows itself
                                     (one launch less per application) */
} flow_momentum_jvp;
int flow_momentum_jvp_apply(const flow_momentum_jvp* J, const double* v,
                            double* out, void* stream);

/* (f, v) for a `dim`-component coefficient: the load vector behind
 * dolfin.project (tests/test_navier_stokes.py:296-308).  scratch: dim*nloc*nc. */
int flow_assemble_source(const flow_mesh* mesh, const flow_space* V, int dim,
                         const flow_coef* f, double* scratch, double* b,
                         void* stream);

/* Stokes bootstrap (flow/stokes.py:40-42): adjoint of the
 * divergence coupling, out_(a,i) = - int p d_a phi_i  (`- p*div(v)*dx`); the
 * coupling itself, - int q div u, is flow_assemble_pressure_rhs with p0 = 0,
 * alpha_rho_dt = 1.  scratch: 2*nloc*nc. */
int flow_assemble_div_adjoint(const flow_mesh* mesh, const flow_space* W,
                              const flow_space* P, const double* p,
                              double* scratch, double* out, void* stream);

/* K16: b_i = int m(u) phi_i with m = sqrt(ux^2+uy^2) (mode 0) or |ux|+|uy|
 * (mode 1): the load vector of the callers' `project(sqrt(ux**2 + uy**2), ...)`
 * step-size control (tests/test_karman_vortex_street.py:262-268,
 * tests/test_boussinesq.py:268-273).  scratch: nloc*nc. */
int flow_assemble_magnitude(const flow_mesh* mesh, const flow_space* W, int mode,
                            const double* u, double* scratch, double* b,
                            void* stream);

/* ---- K7: Dirichlet conditions (bcs= in solve, pressure_correction.py:226,
 * 327,452; bc.apply(A, b), heat.py:113-114).  dofs sorted, in operator
 * numbering (a*n + i). ------------------------------------------------------ */
/* Newton form: rows -> identity (all planes), F[d] = u[d] - g[d]. */
int flow_bc_identity_rows(const flow_operator* A, double* vals_planes,
                          const int* diag_idx, int nbc, const int* dofs,
                          void* stream);
int flow_bc_residual(int nbc, const int* dofs, const double* g, const double* u,
                     double* F, void* stream);
int flow_bc_set_values(int nbc, const int* dofs, const double* g, double* x,
                       void* stream);
/* symmetric elimination (assemble_system semantics, 'symmetric': True):
 * vals_out = vals_in with rows AND columns of marked dofs replaced by identity;
 * isbc: byte mask of length n for the plane's component. */
int flow_bc_symmetric_matrix(int n, const int* rowptr, const int* cols,
                             const double* vals_in, const unsigned char* isbc,
                             double* vals_out, void* stream);

/* ---- K13/K14: heat operator (heat.py:20-89) and SUPG tau
 * (stabilization.py:50-143) -------------------------------------------------
 * A = matrix of  -kappa/(rho cp) (grad u, grad v) - (conv . grad u, v)
 *     [+ SUPG terms], Msupg = (u, tau conv . grad v) (added to the lumped M).
 * Q: temperature space (deg 1|2), W: scalar space of `conv` (2 comps).
 * tau_out (3*nc, optional): tau at the three vertices of every cell.
 * status_dev: int, set to 1 on device if any tau > 1e3 (the reference throws).
 * scratch: 2*nloc^2*nc. */
int flow_assemble_heat(const flow_mesh* mesh, const flow_space* Q,
                       const flow_space* W, const double* conv, double kappa,
                       double rho_cp, int supg, double* scratch, double* Avals,
                       double* Msupg_vals, double* tau_out, int* status_dev,
                       void* stream);

/* SUPG part of the heat load vector with a non-zero source (heat.py:79-86:
 * the `source / rho_cp` term of R2 times tau conv.grad(v)):
 *   b_i = int (source / rho_cp) tau (conv . grad v_i);
 * source: a scalar P0/P1/P2 interpolant per cell (nl = 1, 3, 6; G unused).
 * scratch: nloc*nc. */
int flow_assemble_heat_supg_source(const flow_mesh* mesh, const flow_space* Q,
                                   const flow_space* W, const double* conv,
                                   double kappa, double rho_cp,
                                   const flow_coef* source, double* scratch,
                                   double* b, int* status_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOW_HIP_H */
