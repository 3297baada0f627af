/* =============================================================================
 * amg_hip.h -- C ABI of the MI355X-native V-cycle (libamg_hip.so)
 *
 * Drop-in boundary for the V-cycle hot path of jfdev001/algebraic-multigrid.
 * The reference is header-only C++ (no FFI of its own); these entry points are
 * what a binding of its public interface binds.  Each one cites the reference
 * interface it replaces (paths relative to the reference repository).
 * include/amg/ *.hpp is the C++ drop-in layer (AMG::Multigrid<> etc.) written on
 * top of exactly these functions; INTEGRATION.md shows both.
 *
 * Conventions
 *  - plain pointers + sizes, no C++/torch types; all indices int32 (Eigen's
 *    default StorageIndex), all values fp64 (the only EleType the reference
 *    instantiates, test/testlib.cpp throughout).
 *  - sparse matrices are COMPRESSED COLUMN (Eigen::SparseMatrix<double> default):
 *    colptr[cols+1], rowind[nnz] ascending inside a column, val[nnz].
 *  - every function returns an amg_hip_status; amg_hip_last_error() gives the
 *    thread-local message of the last failure.  The C++ layer re-throws
 *    AMG_HIP_EINVAL as std::invalid_argument with the reference's message text
 *    (multigrid.hpp:165-178, smoother.hpp:286-293).
 *  - there is NO CPU fallback: without a usable HIP device every compute entry
 *    point fails with AMG_HIP_EHIP.
 *  - one handle = one host thread at a time (the reference is not thread-safe
 *    either); different handles are independent.
 * ===========================================================================*/
#ifndef AMG_HIP_H
#define AMG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  AMG_HIP_OK = 0,
  AMG_HIP_EINVAL = 1,       /* bad argument (-> std::invalid_argument)        */
  AMG_HIP_EHIP = 2,         /* HIP runtime / device failure                    */
  AMG_HIP_ENOMEM = 3,       /* host or device allocation failed                */
  AMG_HIP_EUNSUPPORTED = 4, /* valid request this build cannot run on device   */
  AMG_HIP_ECOMM = 5         /* RCCL failure (amg_hip_comm_*, the sharded cycles) */
} amg_hip_status;

/* Smoother plug-ins (smoother.hpp).  0-2 are the reference's three classes;
 * 3-4 are build-side additions the reference does not contain.   */
typedef enum {
  AMG_HIP_SM_SPGS = 0,          /* AMG::SparseGaussSeidel smoother.hpp:86-216:
                                   n_iters x (forward + backward lexicographic
                                   sweep), column-as-row walk, exact order      */
  AMG_HIP_SM_REF_JACOBI = 1,    /* AMG::Jacobi smoother.hpp:223-264 (an in-place
                                   forward Gauss-Seidel, row addressed)         */
  AMG_HIP_SM_SOR = 2,           /* AMG::SuccessiveOverRelaxation :271-373       */
  AMG_HIP_SM_JACOBI = 3,        /* true two-buffer weighted Jacobi              */
  AMG_HIP_SM_MULTICOLOR_GS = 4  /* symmetric multicolour Gauss-Seidel           */
} amg_hip_smoother;

/* Device layout of the level matrices (results are bit-identical in all).    */
typedef enum {
  AMG_HIP_LAYOUT_AUTO = 0, /* DICT when the matrix qualifies, else SELL-64 unless
                              its padding exceeds 25 %, else CSR                 */
  AMG_HIP_LAYOUT_CSR = 1,  /* plain CSR, LDS-staged kernel (K-CSR)               */
  AMG_HIP_LAYOUT_SELL = 2, /* CSR sliced in 64-row lane-interleaved panels       */
  AMG_HIP_LAYOUT_DICT = 3  /* dictionary-coded rows (K-Dict): one byte per entry
                              indexing a table of the matrix's distinct
                              (column offset, value) pairs; needs <= 255 pairs
                              and rows of <= 16 entries, else falls back to SELL.
                              With <= 255 distinct rows the whole row is one byte
                              (amg_hip_set_row_types)                             */
} amg_hip_layout;

/* Options of amg_hip_create.  Zero-initialise, then amg_hip_default_options. */
typedef struct {
  int32_t smoother;        /* amg_hip_smoother; default AMG_HIP_SM_SPGS         */
  int32_t smoother_iters;  /* SmootherBase::n_iters (smoother.hpp:37); default 1
                              (= SparseGaussSeidel(), smoother.hpp:183-187)     */
  double omega;            /* SOR / weighted-Jacobi relaxation; default 1.0     */
  int32_t device;          /* HIP device ordinal; -1 = current device           */
  int32_t use_graph;       /* 1: replay the V-cycle as one hipGraph (default)   */
  int32_t stencil_transfers; /* 1 (default): when P/R are the built-in
                              LinearInterpolator, use the matrix-free stride-2
                              kernels (bit-identical to the CSR path)           */
  int32_t layout;          /* amg_hip_layout of the level matrices on the device  */
  int32_t host_only;       /* 1: build the hierarchy on the host only (no device is
                              touched); getters work, compute entry points fail.
                              Used by the row-block multi-GPU driver, where every
                              rank slices its own rows out of the hierarchy.       */
  int32_t keep_structural_zeros; /* 0 (default): device copies of the level matrices
                              drop entries that are exactly 0.0 (Eigen's Galerkin
                              product keeps them, 22 % of level 1); results are
                              bit-identical, the getters still return them.       */
  int32_t no_fusion;       /* 0 (default): the true-Jacobi V-cycle takes the first
                              pre-smoothing sweep of a coarse level, whose input is the
                              zero vector, from f and the diagonal alone.  Same
                              operations on the same values: bit-identical.       */
  int32_t fuse_prolong;    /* 1: apply the prolongation inside the first post-smoothing
                              Jacobi sweep (bit-identical; measured slower, off)  */
  int32_t fast_coarse_solve; /* 1: solve the coarsest system with the partitioned
                              (parallel) form of the banded LDL^T whenever it applies
                              (half-bandwidth <= 63): depth ~2c+P steps instead of n.
                              Same direct solve, different rounding order: agrees with
                              the sequential substitution to ~1e-14 relative, not bit
                              for bit.  Default 0 = by size: partitioned from 4096
                              coarsest rows on (see exact_coarse_solve).             */
  int32_t host_galerkin;   /* 1: build the Galerkin products R (A P) on the host instead of
                              on the device (same entries, same bits; the device path
                              is taken for LinearInterpolator operators only)        */
  int32_t keep_residual;   /* 1: keep the level residuals r = f - A u readable through
                              amg_hip_get_vec(level, 2).  The reference holds them as
                              private workspace without a getter (multigrid.hpp:107), so
                              by default (0) the fused residual+restriction kernel does
                              not store r on the levels where it runs, and reading it
                              there returns AMG_HIP_EINVAL.                            */
  int32_t exact_coarse_solve; /* 1: always the sequential substitution (bit-exact against
                              the row-oriented LDL^T solve of the oracle), whatever the
                              size of the coarsest level.  Default 0.                   */
  int32_t exact_gs;        /* SparseGaussSeidel / AMG::Jacobi / SOR (lexicographic sweeps,
                              smoother.hpp:148-174).  1: always the dependency-scheduled
                              exact kernel (bit-exact, one latency-bound workgroup).
                              0 (default): levels of more than 65536 rows whose matrix
                              is dictionary-coded with one chained lower neighbour run
                              the line-scan form (the in-line recurrence u_k = c_k +
                              q_k u_{k-1} solved by a parallel affine scan: same sweep,
                              different rounding order, ~1e-13 relative); smaller
                              problems keep the exact kernel.                          */
  void* stream;            /* hipStream_t to run on; NULL (default) = the solver creates
                              and owns a non-blocking stream.  A caller that already
                              orders its device work on a stream (torch's current
                              stream in the multi-GPU driver) passes it here so that
                              no host synchronisation is needed between the two.   */
  int32_t window;          /* 1: the solver is one rank's WINDOW of a sharded hierarchy
                              ("window sharding" below): its coarsest level is the level
                              the ranks all-gather, so it is neither factored nor solved
                              here; amg_hip_vcycle / solve / apply / pcg are refused and
                              the cycle runs by parts (amg_hip_window_run).  Default 0. */
  int32_t reserved0;       /* 0 */
} amg_hip_options;

typedef struct amg_hip_solver amg_hip_solver; /* opaque; owns device memory    */

const char* amg_hip_last_error(void);
void amg_hip_default_options(amg_hip_options* o);

/* Layout used by amg_hip_default_options and by the stand-alone host-array
 * operations below (process-wide; tests flip it to run both kernels).         */
void amg_hip_set_default_layout(int32_t layout);

/* SELL-64 column indices as 16-bit offsets from the row's diagonal column when
 * every entry of a matrix is within +-32767 of it (10 instead of 12 bytes per
 * entry; bit-identical results).  Process-wide, default on.                    */
void amg_hip_set_index16(int32_t on);
/* Stream matrices larger than ~192 MB with non-temporal loads (process-wide, default
 * on): the once-read matrix then does not evict the x lines the gathers re-use.  */
void amg_hip_set_nontemporal(int32_t on);
/* K-Dict: give every XCD one contiguous run of row tiles, so that the +-bandwidth
 * re-reads of x hit the L2 that fetched them; on wide bands (3-D levels) the same slab of lines of
 * every grid plane instead (default on = 1; 2 = contiguous runs everywhere, the round-2 order;
 * 0 = plain order).  Tuning switch.                                                */
void amg_hip_set_xcd_mapping(int32_t on);
/* K-Dict second level: when a dictionary-coded matrix has at most 255 distinct rows, store
 * one byte per row into a table of code words instead of the code words themselves
 * (process-wide, default on; bit-identical results).  Applies to matrices uploaded
 * after the call.                                                                */
void amg_hip_set_row_types(int32_t on);
/* K-Dict rows per lane: 2 (default; 16-byte lane accesses) or 1.  Process-wide;
 * bit-identical results, a tuning / test switch.                                */
void amg_hip_set_dict_rows(int32_t rows_per_lane);
/* K-Dict: waves whose 128 rows all share one 7-point ({-M,-m,-1,0,1,m,M}) or 15-point
 * ({-M,-m,0,m,M} x {-1,0,1}) row type -- the interior of a 3-D level -- fetch x as aligned pairs
 * (7 / 15 loads per lane instead of 16 / 32 gathers) with the values from a per-type scalar table.
 * Process-wide, default on (AMG_HIP_DICT_STENCIL=0 in the environment turns it off);
 * bit-identical results, a tuning / test switch.                                                */
void amg_hip_set_dict_stencil(int32_t on);
/* K-Patch (temporal blocking of the 2+2 true-Jacobi cycle: a level's down-leg and up-leg in
 * one launch each over 2-D patches of the level) is used on levels of at least this many rows
 * (default 10^6; negative = never).  Process-wide, read when a solver is created: existing
 * solvers keep theirs.  Bit-identical results; tests set 0.                              */
void amg_hip_set_patch_min_rows(int64_t rows);

/* K-Patch per-tile row-type flags (patches whose rows all share one type skip the row-type
 * loads and the bounds checks) on / off; process-wide, read at launch.  Same bits either way. */
void amg_hip_set_patch_tile_flags(int32_t on);
/* K-BandChain (coarsest solve kind 3) on / off; process-wide, read when a solver is created.
 * Same bits either way: an A/B switch for tests and tuning.                             */
void amg_hip_set_band_chain(int32_t on);

/* K-Tail (the deepest levels of the 2+2 true-Jacobi cycle -- at most 4095 rows each --, the
 * coarsest solve and the way back up in ONE launch of one workgroup) on / off; process-wide, read
 * when a cycle is enqueued (a captured graph keeps what it was captured with).  Same bits either
 * way.  OFF by default (AMG_HIP_TAIL_FUSION=1 in the environment turns it on): on MI355X the one
 * launch takes as long as the launches it replaces.                       */
void amg_hip_set_tail_fusion(int32_t on);

/* Number of usable HIP devices (0 when none; never fails). */
int amg_hip_device_count(void);

/* ---- AMG::Multigrid<double> (multigrid.hpp:151-244, ctor = SETUP) ----------
 * A (n x n CSC) and b are copied (multigrid.hpp:193,201).  Level sizes follow
 * n_H = (n_h+1)/2 - 1 (multigrid.hpp:127-130); transfer operators are the
 * built-in LinearInterpolator (interpolator.hpp:106-141) unless custom P/R are
 * given; coarse operators are Galerkin R*(A*P) in Eigen's summation order with
 * structural zeros kept (multigrid.hpp:219-223); the coarsest level is factored
 * (banded LDL^T, replaces Eigen::SimplicialLDLT multigrid.hpp:240-243).
 * Everything is uploaded; u_0 = 0.                                            */
amg_hip_status amg_hip_create(int64_t n, const int32_t* colptr,
                              const int32_t* rowind, const double* val,
                              const double* b, int32_t n_levels,
                              const amg_hip_options* opts,
                              amg_hip_solver** out);

/* Same, for a user InterpolatorBase subclass (interpolator.hpp:43-44): the C++
 * layer runs make_operators(n_h, n_H, level) on the host for every level and
 * hands over P_l (n_l x n_{l+1}) and R_l (n_{l+1} x n_l), l = 0..n_levels-2,
 * as arrays of CSC triples.                                                    */
amg_hip_status amg_hip_create_custom(int64_t n, const int32_t* colptr,
                                     const int32_t* rowind, const double* val,
                                     const double* b, int32_t n_levels,
                                     const int32_t* const* P_colptr,
                                     const int32_t* const* P_rowind,
                                     const double* const* P_val,
                                     const int32_t* const* R_colptr,
                                     const int32_t* const* R_rowind,
                                     const double* const* R_val,
                                     const amg_hip_options* opts,
                                     amg_hip_solver** out);

/* Strength-based C/F coarsening (/root/reference README.md:104-109 names
 * Ruge-Stueben as the alternative it did not build: NO reference counterpart, pinned to the
 * oracle twin only).  Classical first-pass C/F splitting on the strength graph
 * (threshold `theta`, typically 0.25) with direct interpolation, R = P^T, Galerkin R(AP) in
 * the same summation order as the reference's chain; the hierarchy ends at `max_levels` or
 * where a level has at most `min_coarse` rows or stops coarsening (amg_hip_n_levels tells).
 * The same V-cycle (multigrid.hpp:263-305) then runs on it: general CSR transfer kernels,
 * the smoother of `opts`, banded LDL^T of whatever width on the coarsest level.          */
amg_hip_status amg_hip_create_rs(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                 const double* val, const double* b, int32_t max_levels,
                                 double theta, int64_t min_coarse, const amg_hip_options* opts,
                                 amg_hip_solver** out);

/* The same constructor for the reference's own model problem, A = Grid::laplacian(n) and
 * b = Grid::rhs(n) (grid.hpp:88-98,108-140; dim = 3: the 7-point analogue), with the built-in
 * LinearInterpolator -- SETUP ON THE DEVICE END TO END: the
 * matrix generator, the Galerkin chain, the dictionary encoder, the diagonal and the symmetry
 * check are kernels; no host copy of any level matrix is made unless a getter asks for it
 * (the right-hand side is evaluated on the host threads: its values are libm's exp()).
 * Identical hierarchy, identical results (tests compare level matrices and V-cycles bitwise
 * with amg_hip_create on Grid-generated arrays).  Options that need host structures -- exact
 * lexicographic schedules (exact_gs or small problems), multicolouring, non-dictionary
 * layouts, host_only -- silently take the host path: generate, then amg_hip_create.       */
amg_hip_status amg_hip_create_poisson(int32_t dim, int64_t n, int32_t n_levels,
                                      const amg_hip_options* opts, amg_hip_solver** out);

void amg_hip_destroy(amg_hip_solver* s); /* ~Multigrid, multigrid.hpp:135 */

/* One V-cycle, multigrid.hpp:263-305 (including the smoothing + residual on
 * the coarsest level).  Asynchronous on the solver's stream.      */
amg_hip_status amg_hip_vcycle(amg_hip_solver* s);
/* n back-to-back V-cycles (what bench.py times); returns after enqueueing.   */
amg_hip_status amg_hip_vcycles(amg_hip_solver* s, int32_t n);
/* Block until the solver's stream is idle. */
amg_hip_status amg_hip_sync(amg_hip_solver* s);

/* ---- row-block ("slab") sharding of one V-cycle over several GPUs ---------
 * The reference is a serial program (multigrid.hpp:263-305 walks whole levels); this is the
 * build's multi-GPU form of that loop.  Every rank creates the SAME solver (whole hierarchy,
 * full-size vectors) and then runs the K-Patch levels 0..levels-1 only over its own block of
 * grid lines plus `halo_lines` lines either side, recomputed redundantly, so that the whole
 * cycle needs two exchanges, both done by the caller (RCCL / torch.distributed) on the
 * solver's stream:
 *   1. before part 1: lines [line_begin - halo_lines, line_begin) and [line_end, line_end +
 *      halo_lines) of u0 from the two neighbouring ranks (their owned lines, same offsets);
 *   2. between parts 1 and 2: all-gather of f_gather, rank g contributing the entries
 *      [g, g+1) * chunk_lines * gather_pitch (equal blocks; the allocation has room for
 *      world of them, entries past gather_rows are padding);
 * part 2 (levels >= `levels`, coarse solve included) runs replicated on every rank.  After
 * part 3 each rank holds its own lines of the level-0 solution; per-row arithmetic is the
 * single-GPU cycle's, so the assembled solution equals it bit for bit.
 * amg_hip_slab_plan is the pure host arithmetic (no device): per slab level l the lines
 * [down_lo[l], down_hi[l]) the down-leg runs over and [up_lo[l], up_hi[l]) for the up-leg. */
#define AMG_HIP_SLAB_MAX_LEVELS 8
typedef struct amg_hip_slab_info {
  int32_t levels;      /* slab levels (0 .. levels-1); level `levels` is the gathered one   */
  int32_t halo_lines;  /* lines of u0 each neighbour supplies before a cycle                */
  int64_t lines;       /* grid lines (the same on every slab level: x-only coarsening)      */
  int64_t chunk_lines; /* lines per rank = ceil(lines / world); the last rank may own fewer  */
  int64_t line_begin, line_end; /* this rank's lines                                        */
  int64_t pitch0;       /* entries per line on level 0                                      */
  int64_t gather_pitch; /* entries per line on level `levels`                               */
  int64_t gather_rows;  /* rows of level `levels`                                           */
  int64_t down_lo[AMG_HIP_SLAB_MAX_LEVELS], down_hi[AMG_HIP_SLAB_MAX_LEVELS];
  int64_t up_lo[AMG_HIP_SLAB_MAX_LEVELS], up_hi[AMG_HIP_SLAB_MAX_LEVELS];
  double* u0;           /* device: level-0 solution, room for world*chunk_lines*pitch0      */
  double* f_gather;     /* device: rhs of level `levels`, room for world*chunk_lines*gather_pitch */
} amg_hip_slab_info;
/* AMG_HIP_EUNSUPPORTED when a rank would own fewer lines than the halo depth. */
amg_hip_status amg_hip_slab_plan(int64_t lines, int32_t rank, int32_t world, int32_t levels,
                                 amg_hip_slab_info* out);
/* max_levels < 0: every K-Patch level.  AMG_HIP_EUNSUPPORTED when the solver has none.     */
amg_hip_status amg_hip_slab_setup(amg_hip_solver* s, int32_t rank, int32_t world,
                                  int32_t max_levels, amg_hip_slab_info* info);
/* part 1: down-legs of the slab levels; 2: the replicated rest of the cycle (from the gathered
 * rhs); 3: up-legs of the slab levels.  Asynchronous on the solver's stream (one hipGraph each). */
amg_hip_status amg_hip_slab_run(amg_hip_solver* s, int32_t part);

/* ---- window sharding: row blocks with sharded STORAGE and SETUP ------------
 * The reference's flat-index coarsening (multigrid.hpp:127-130, interpolator.hpp:116-129)
 * coarsens the fast axis only, so the grid lines (2-D) / x-y planes (3-D) -- "units" -- of every
 * level coincide and a block of whole units is a consistent cut of the whole hierarchy.  Rank g
 * creates an ORDINARY solver on the principal submatrix of its WINDOW of units [unit_begin,
 * unit_end) = its owned units plus `halo` units either side (A = the rows and columns of
 * Grid::laplacian(n) in the window, b = the window of Grid::rhs(n); grid.hpp:88-140): because
 * the window starts at an even flat index, coarse dof j of the window is coarse dof j + offset
 * of the whole problem, the window's LinearInterpolator is the window of the global one, and
 * every Galerkin row R (A P) (multigrid.hpp:219-223) whose fine rows lie two units inside the
 * window has the same entries, summed in the same order, as the global row -- bit-identical.
 * Nothing of the global problem is ever assembled on a rank above the gathered level.
 * One V-cycle (multigrid.hpp:263-305) on `k` distributed levels:
 *   1. halo: `halo` units of the level-0 solution from each neighbour (their owned units);
 *   2. amg_hip_window_run(s, 1): the down-legs of levels 0..k-1 over the window, whatever the
 *      smoother (each sweep / colour stage / residual / transfer makes one more unit at either
 *      end of the window stale; the halo depth is chosen so that the owned units of f_k and
 *      what the up-legs need of u_l stay valid: window_vcycle.py: WindowPlan);
 *   3. all-gather of the owned units of f_k; the levels >= k (their own solver, built from the
 *      all-gathered rows of A_k) run replicated, coarse solve included;
 *   4. the window of u_k goes back into level k of the window solver;
 *      amg_hip_window_run(s, 3): the up-legs of levels k-1..0.
 * Per-row arithmetic is the single-GPU kernels', so the owned units of the result equal the
 * single-GPU cycle bit for bit (tests/test_window_gloo.py, tests/test_gpu_window.py).
 * amg_hip_create_poisson_window: options as amg_hip_create_poisson; `n_levels` = k + 1 (level k
 * is only a container for f_k / u_k: opts->window is forced to 1).                         */
amg_hip_status amg_hip_create_poisson_window(int32_t dim, int64_t n, int64_t unit_begin,
                                             int64_t unit_end, int32_t n_levels,
                                             const amg_hip_options* opts, amg_hip_solver** out);
/* Line ranges (window-local grid lines) for the K-Patch legs of the distributed levels, or
 * NULL arrays: every leg runs over the whole window.  k = amg_hip_n_levels(s) - 1.           */
amg_hip_status amg_hip_window_setup(amg_hip_solver* s, const int64_t* down_lo,
                                    const int64_t* down_hi, const int64_t* up_lo,
                                    const int64_t* up_hi);
/* part 1: down-legs (u_0 .. f_k), part 3: up-legs (u_k .. u_0); one hipGraph each.           */
amg_hip_status amg_hip_window_run(amg_hip_solver* s, int32_t part);
/* Device pointer of a level vector (which as in amg_hip_get_vec) for zero-copy exchanges. */
amg_hip_status amg_hip_vec_dev_ptr(amg_hip_solver* s, int32_t level, int32_t which, void** ptr,
                                   int64_t* n);

/* The stream the solver's work is enqueued on (its own, or opts->stream). */
amg_hip_status amg_hip_get_stream(amg_hip_solver* s, void** stream);

/* ---- communicator: RCCL from inside the library --------------------------------
 * The reference is a serial program; this is how a C / C++ caller of the drop-in runs the sharded
 * V-cycle (multigrid.hpp:263-305 over row blocks) without Python.  librccl.so is opened at run
 * time (dlopen; AMG_HIP_EUNSUPPORTED when there is none).  Rank 0 calls amg_hip_comm_unique_id
 * and hands the 128 bytes to every rank by whatever bootstrap the application has (MPI_Bcast,
 * torch.distributed, a file); every rank then calls amg_hip_comm_create (collective).  One process
 * per GPU.  The sharded cycles below are ONE call each: everything -- grouped ncclSend / ncclRecv
 * of the halo, the captured legs, one ncclAllGather, the replicated rest -- is enqueued on the
 * solver's stream, nothing synchronises with the host.                                        */
#define AMG_HIP_COMM_ID_BYTES 128
typedef struct amg_hip_comm amg_hip_comm;
amg_hip_status amg_hip_comm_unique_id(uint8_t id[AMG_HIP_COMM_ID_BYTES]);
amg_hip_status amg_hip_comm_create(const uint8_t id[AMG_HIP_COMM_ID_BYTES], int32_t rank, int32_t world,
                                   int32_t device, amg_hip_comm** out);
void amg_hip_comm_destroy(amg_hip_comm* c);
int32_t amg_hip_comm_rank(const amg_hip_comm* c);
int32_t amg_hip_comm_world(const amg_hip_comm* c);
/* grouped send / recv with rank-1 and rank+1 (counts in doubles; a null pointer or a zero count
 * skips that direction); `stream` = hipStream_t                                                */
amg_hip_status amg_hip_comm_neighbor_exchange(amg_hip_comm* c, const double* send_prev, int64_t n_send_prev,
                                              double* recv_prev, int64_t n_recv_prev,
                                              const double* send_next, int64_t n_send_next,
                                              double* recv_next, int64_t n_recv_next, void* stream);
/* out = [the `count` doubles of rank 0 | of rank 1 | ...]; in == out + rank * count is allowed */
amg_hip_status amg_hip_comm_all_gather(amg_hip_comm* c, const double* in, double* out, int64_t count,
                                       void* stream);
/* One slab-sharded V-cycle (the slab section above): `info` is what amg_hip_slab_setup returned
 * for (amg_hip_comm_rank, amg_hip_comm_world).  With a communicator of one rank this is
 * amg_hip_vcycle cut into its three graphs: same bits.                                       */
amg_hip_status amg_hip_slab_cycle(amg_hip_solver* s, amg_hip_comm* c, const amg_hip_slab_info* info);
/* One window-sharded V-cycle (the window section above): `w` the window solver, `t` the solver of
 * the replicated levels >= k (built from the gathered rows of A_k), both on ONE stream.  All
 * offsets / counts in doubles:
 *   level-0 solution of the window: owned rows [own0_off, own0_end); send_prev_cnt of its first
 *   rows go to rank-1, send_next_cnt of its last rows to rank+1; recv_prev_cnt rows arrive below
 *   own0_off, recv_next_cnt rows from own0_end on;
 *   level k: owned rows [own_k_off, own_k_off + own_k_cnt) of the window's f_k are this rank's
 *   block (block_k doubles per rank, padded) of the all-gather; the window's u_k is rows
 *   [uk_off, uk_off + n) of the tail's solution.                                            */
typedef struct amg_hip_window_plan {
  int64_t own0_off, own0_end;
  int64_t send_prev_cnt, recv_prev_cnt, send_next_cnt, recv_next_cnt;
  int64_t own_k_off, own_k_cnt, block_k, uk_off;
} amg_hip_window_plan;
amg_hip_status amg_hip_window_cycle(amg_hip_solver* w, amg_hip_solver* t, amg_hip_comm* c,
                                    const amg_hip_window_plan* p);

/* Multigrid::solve(), multigrid.hpp:311-337: while (iter < n_iters && error >
 * tol) { vcycle(); if (++iter % every == 0) error = rss }.  error starts at 100.
 * *converged = (error <= tol).  Prints nothing (the C++ layer prints the
 * reference's "AMG converged after N iterations." line).                      */
amg_hip_status amg_hip_solve(amg_hip_solver* s, double tol, int64_t every,
                             int64_t n_iters, int64_t* iters, double* last_rss,
                             int32_t* converged);

/* ---- the V-cycle as a preconditioner (README.md:127 of the reference, its ref [7]: "a
 * single V-cycle used for preconditioner to get M^-1 v") ------------
 * amg_hip_apply: z = M^-1 v, i.e. ONE vcycle() (multigrid.hpp:263-305) started from the
 * zero vector with v as the level-0 right-hand side.  v_dev / z_dev are DEVICE pointers
 * (n_dofs(0) doubles, may alias each other); enqueued on the solver's stream; the solver's
 * own right-hand side and solution are left as they were.                             */
amg_hip_status amg_hip_apply(amg_hip_solver* s, const double* v_dev, double* z_dev);
/* Preconditioned conjugate gradients for A_0 x = b with M^-1 = amg_hip_apply, entirely on the
 * device (SpMV, dot products, updates; alpha / beta never leave it).  Starts from the current
 * level-0 solution, replaces it with the result.  Stops when ||b - A x||_2 <= rtol ||b||_2
 * (checked every iteration) or after max_iters.  The smoother must make the cycle a
 * symmetric operator: SparseGaussSeidel (forward + backward), true Jacobi or multicolour GS
 * with equal pre / post sweeps all do.  A is the reference's negative definite Laplacian
 * (grid.hpp:88-98): CG runs on it unchanged (r.z and p.Ap are both negative).            */
amg_hip_status amg_hip_pcg(amg_hip_solver* s, double rtol, int64_t max_iters, int64_t* iters,
                           double* relres);

/* AMG::rss(A_0, u_0, b), common.hpp:17-27 (device tree reduction; agrees with
 * the sequential sum to ~1e-15 relative). */
amg_hip_status amg_hip_rss(amg_hip_solver* s, double* out);

/* Getters, multigrid.hpp:339-354. */
int32_t amg_hip_n_levels(const amg_hip_solver* s);
int64_t amg_hip_get_n_dofs(const amg_hip_solver* s, int32_t level);
int64_t amg_hip_get_level_nnz(const amg_hip_solver* s, int32_t level);
/* get_coefficient_matrix(level): CSC copy out (arrays sized by the caller). */
amg_hip_status amg_hip_get_level_matrix(const amg_hip_solver* s, int32_t level,
                                        int32_t* colptr, int32_t* rowind,
                                        double* val);
/* which: 0 = P_level (n_l x n_{l+1}), 1 = R_level.  Returns nnz (<0: error). */
int64_t amg_hip_get_transfer_nnz(const amg_hip_solver* s, int32_t level, int32_t which);
amg_hip_status amg_hip_get_transfer(const amg_hip_solver* s, int32_t level,
                                    int32_t which, int32_t* colptr,
                                    int32_t* rowind, double* val);
/* which: 0 = get_soln, 1 = get_rhs, 2 = residual vector (multigrid.hpp:107).
 * Synchronises the stream, then copies n_dofs(level) doubles to/from host.    */
amg_hip_status amg_hip_get_vec(amg_hip_solver* s, int32_t level, int32_t which,
                               double* out);
amg_hip_status amg_hip_set_vec(amg_hip_solver* s, int32_t level, int32_t which,
                               const double* in);
/* Device-side copy between a level vector of the solver and caller-owned
 * DEVICE memory (n_dofs(level) doubles), enqueued on the solver's stream:
 * to_solver = 1 copies src -> solver vector, 0 copies solver vector -> dst.
 * which as in amg_hip_get_vec.  Lets a host that owns device buffers (the
 * multi-GPU driver's agglomerated coarse part) feed a right-hand side and take
 * the solution without a host round trip.                                      */
amg_hip_status amg_hip_copy_vec_dev(amg_hip_solver* s, int32_t level, int32_t which,
                                    void* dev_ptr, int32_t to_solver);
/* Zero-fill a level vector on the solver's stream. */
amg_hip_status amg_hip_zero_vec(amg_hip_solver* s, int32_t level, int32_t which);

/* Device layout the level's operator was uploaded in (amg_hip_layout, never AUTO)
 * and the bytes of its matrix stream (indices + values, or codes + table) that
 * one sweep reads -- what the format actually moves, next to the CSR-formula
 * figure of amg_hip_cycle_bytes.  host_only solvers: EINVAL.                    */
amg_hip_status amg_hip_level_layout(const amg_hip_solver* s, int32_t level, int32_t* layout,
                                    int64_t* matrix_stream_bytes);

/* Half-bandwidth of the factored coarsest operator. */
int64_t amg_hip_coarse_halfbw(const amg_hip_solver* s);
/* Which device form of the coarsest solve (multigrid.hpp:287-288) the solver uses:
 * 0 = one-wave sequential substitution (half-bandwidth <= 63, bit-exact),
 * 1 = partitioned / parallel (fast_coarse_solve or >= 4096 rows),
 * 2 = blocked sequential substitution for any half-bandwidth (bit-exact),
 * 3 = LDS-resident scalar recurrence for half-bandwidth <= 3 and <= 2048 rows (bit-exact;
 *     the coarsest level of a deep hierarchy).                                        */
int32_t amg_hip_coarse_solve_kind(const amg_hip_solver* s);
/* Multicolour smoother: colour of every dof of `level` (n_dofs int32) and the
 * colour count, so a CPU twin can replay the same colouring.                   */
amg_hip_status amg_hip_get_colors(const amg_hip_solver* s, int32_t level,
                                  int32_t* color, int32_t* n_colors);

/* Single steps of the V-cycle on one level, for a host that has to interleave its
 * own code with them: the C++ layer runs a USER-DEFINED SmootherBase subclass on
 * the host (vectors go device -> host -> smooth() -> device) and
 * everything else here.  op:
 *   0 smooth with the solver's built-in smoother           (multigrid.hpp:268,300)
 *   1 residual r_l = f_l - A_l u_l                          (:272-274)
 *   2 u_{l+1} = 0 and f_{l+1} = R_l r_l                     (:278-282)
 *   3 u_l = u_l + P_l u_{l+1}                               (:294-296)
 *   4 coarsest solve u_L = A_L^-1 f_L (level must be L-1)   (:287-288)        */
amg_hip_status amg_hip_level_op(amg_hip_solver* s, int32_t level, int32_t op);

/* Sum over levels of the algorithmic HBM bytes of one V-cycle (SURVEY 8(d)
 * formulae) and the per-sweep bytes of level 0; used by bench.py.             */
amg_hip_status amg_hip_cycle_bytes(const amg_hip_solver* s, double* cycle_bytes,
                                   double* fine_sweep_bytes);

/* Bytes the launches of ONE V-cycle have to move: for every launch what that kernel reads and
 * writes once in the layout it really streams (dictionary: 1 B row type per row; K-Patch leg:
 * 25 n + 24 n_H down, 25 n + 8 n_H up; SELL / CSR: indices + values; vectors 8 B per entry),
 * summed while the cycle is enqueued.  part 0 = the whole cycle (bench.py divides it by the
 * measured time per cycle: the whole-cycle fraction of the HBM roof), 1 / 2 / 3 = the parts of
 * amg_hip_slab_run / amg_hip_window_run once they have run.                                  */
amg_hip_status amg_hip_cycle_must_move(amg_hip_solver* s, int32_t part, double* bytes);

/* Measurement hook for bench.py: launches the level-0 smoother sweep kernel
 * (the dominant kernel of a V-cycle) n_launches times back to back on the
 * solver's own stream, each launch bracketed by a pair of HIP events on that
 * stream, and returns the average / minimum launch duration in milliseconds.
 * AMG_HIP_SM_JACOBI: one launch = one sweep over level 0 (K-Patch: the level's down-leg);
 * AMG_HIP_SM_MULTICOLOR_GS: the first launch of the symmetric pass (K-Patch form: two colour
 * stages over the level; colour kernels: colour 0).  The level-0 solution is restored
 * afterwards.                                                                  */
amg_hip_status amg_hip_profile_fine_sweep(amg_hip_solver* s, int32_t n_launches,
                                          double* avg_ms, double* min_ms);
/* Name of the kernel amg_hip_profile_fine_sweep times (as rocprofv3 prints it, without the
 * namespace), the number of Jacobi sweeps over level 0 one launch of it performs, and the
 * bytes one launch has to move (what the kernel reads and writes once: SURVEY 8(d)'s CSR
 * formula for the CSR / SELL layouts; row types + f + x + out for the dictionary-coded one;
 * for the K-Patch down-leg additionally f_H, the first coarse sweep and the coarse diagonal). */
amg_hip_status amg_hip_fine_sweep_info(const amg_hip_solver* s, char* name, int32_t name_cap,
                                       int32_t* sweeps_per_launch, double* bytes_per_launch);

/* ---- stand-alone plug-in operations on host arrays (run on the device) -----
 * SmootherBase::smooth(A, u, b), smoother.hpp:63-65, for the built-in kinds.
 * u is updated in place.  every = compute_error_every_n_iters (0 = never, the
 * SparseGaussSeidel() default); *iters and *converged mirror the reference's
 * loop (smoother.hpp:195-203).  For kinds 3-4 `n_iters` counts sweeps /
 * symmetric colour passes and tol/every are ignored.                          */
amg_hip_status amg_hip_smooth(int32_t kind, int64_t n, const int32_t* colptr,
                              const int32_t* rowind, const double* val,
                              double* u, const double* b, double omega,
                              double tol, int64_t every, int64_t n_iters,
                              int64_t* iters, int32_t* converged);
/* One lexicographic sweep, dir=+1 forward (smoother.hpp:148-157) or -1 backward
 * (:167-174). */
amg_hip_status amg_hip_spgs_sweep(int32_t dir, int64_t n, const int32_t* colptr,
                                  const int32_t* rowind, const double* val,
                                  double* u, const double* b);
/* r = f - A*u, multigrid.hpp:272-274 (Eigen order: ascending column per row). */
amg_hip_status amg_hip_residual(int64_t n, const int32_t* colptr,
                                const int32_t* rowind, const double* val,
                                const double* u, const double* f, double* r);
/* out = M*v for a rows x cols CSC matrix: InterpolatorBase::prolongation /
 * restriction, interpolator.hpp:52-56,64-68. */
amg_hip_status amg_hip_spmv(int64_t rows, int64_t cols, const int32_t* colptr,
                            const int32_t* rowind, const double* val,
                            const double* v, double* out);
/* Built-in LinearInterpolator transfers without a matrix (bit-identical to
 * amg_hip_spmv on make_operators' P/R): f_H = R r  and  u_h += P u_H
 * (interpolator.hpp:106-141, multigrid.hpp:281-282,294-296). */
amg_hip_status amg_hip_linear_restrict(int64_t n_h, int64_t n_H, const double* r,
                                       double* f_H);
amg_hip_status amg_hip_linear_prolong_add(int64_t n_h, int64_t n_H,
                                          const double* u_H, double* u_h);
/* AMG::rss(A, u, b), common.hpp:17-27. */
amg_hip_status amg_hip_rss_host(int64_t n, const int32_t* colptr,
                                const int32_t* rowind, const double* val,
                                const double* u, const double* b, double* out);
/* x = A^-1 f by the banded LDL^T used for the coarsest level
 * (multigrid.hpp:287-288); *halfbw returns the bandwidth found. */
amg_hip_status amg_hip_coarse_solve(int64_t n, const int32_t* colptr,
                                    const int32_t* rowind, const double* val,
                                    const double* f, double* x, int64_t* halfbw);

/* The same solve by the partitioned (parallel) algorithm of opt.fast_coarse_solve. */
amg_hip_status amg_hip_coarse_solve_fast(int64_t n, const int32_t* colptr,
                                         const int32_t* rowind, const double* val,
                                         const double* f, double* x, int64_t* halfbw,
                                         int32_t* partition_rows);

/* ---- Grid<double> problem generators (grid.hpp), host side -----------------
 * laplacian: grid.hpp:88-98 (dim 2) / 7-point analogue (dim 3).  Call with NULL
 * arrays to get nnz.  rhs: grid.hpp:108-140, default Gaussian forcing.        */
int64_t amg_hip_laplacian(int32_t dim, int64_t n, int32_t* colptr,
                          int32_t* rowind, double* val);
amg_hip_status amg_hip_rhs(int32_t dim, int64_t n, double* b);

/* ---- device-pointer kernel launchers ----------------------------------------
 * For hosts that own device memory and streams themselves (the multi-GPU
 * row-block driver: torch tensors + torch.distributed halo exchange).  All
 * pointers are DEVICE pointers; `stream` is a hipStream_t (NULL = default).
 * CSR here = row-major view (rowptr/col/val); for a symmetric A the CSC arrays
 * are the CSR arrays.  Column indices address `u` directly, so a rank passes a
 * halo-extended local vector and locally renumbered columns.                  */
/* Host-only probe of the dictionary coding (no device needed): encodes the CSR block
 * the way amg_hip_devmat_create / the level upload would (exact zeros are NOT dropped
 * here), decodes it again and compares with the input.  AMG_HIP_OK: the block qualifies
 * and the round trip is exact; n_pairs = distinct (column offset, value) pairs,
 * n_row_types = distinct rows as whole code words (0 when more than 255, i.e. first
 * level only), words = 64-bit code words per row.  AMG_HIP_EUNSUPPORTED: it does not
 * qualify (more than 255 pairs, a row longer than 16 entries, ...).              */
amg_hip_status amg_hip_dict_probe(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                                  const int32_t* col, const double* val, int64_t diag_shift,
                                  int32_t* n_pairs, int32_t* n_row_types, int32_t* words);
/* Launch parameters of the CSR kernels, computed from a HOST copy of rowptr:
 * max entries in any block of 256 consecutive rows, and the longest row.     */
amg_hip_status amg_hip_csr_shape(int64_t nrows, const int32_t* rowptr_host,
                                 int32_t* max_block_nnz, int32_t* max_row_nnz);
amg_hip_status amg_hip_dev_residual(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                    int32_t max_row_nnz, const int32_t* rowptr,
                                    const int32_t* col, const double* val,
                                    const double* u, const double* f, double* r,
                                    void* stream);
/* u_out[i] = u_in[i'] + omega*((b[i]-sum_{j!=i'} a_ij u_in[j])/a_ii' - u_in[i'])
 * with i' = i + diag_shift: the column that is row i's diagonal (a rank passes
 * the offset of its first owned row inside its halo-extended vector). b and
 * u_out are indexed by the local row i.                                       */
amg_hip_status amg_hip_dev_jacobi(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                  int32_t max_row_nnz, const int32_t* rowptr,
                                  const int32_t* col, const double* val,
                                  const double* u_in, const double* b,
                                  double* u_out, double omega, int64_t diag_shift,
                                  void* stream);
amg_hip_status amg_hip_dev_spmv(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                int32_t max_row_nnz, const int32_t* rowptr,
                                const int32_t* col, const double* val,
                                const double* v, double* out, void* stream);
/* The same three operations on a matrix the LIBRARY uploads in its own device
 * layout (amg_hip_layout; AUTO = dictionary-coded when the block qualifies, else
 * SELL-64 panels; exact zeros dropped):
 * rows x cols local CSR block given on the host, columns indexing the vector the
 * operation is applied to.  op: 0 = residual (out = f - A x), 1 = Jacobi sweep
 * (diagonal of row i at column i + diag_shift), 2 = SpMV (f unused).  The
 * diag_shift given at creation (0 for a plain matrix) must be the one passed to
 * apply (the 16-bit relative column indices are built against it).            */
typedef struct amg_hip_devmat amg_hip_devmat;
amg_hip_status amg_hip_devmat_create(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                                     const int32_t* col, const double* val, int32_t layout,
                                     int64_t diag_shift, int32_t device, amg_hip_devmat** out);
void amg_hip_devmat_destroy(amg_hip_devmat* m);
/* Layout the block was uploaded in and its matrix stream bytes (amg_hip_level_layout). */
amg_hip_status amg_hip_devmat_layout(const amg_hip_devmat* m, int32_t* layout,
                                     int64_t* matrix_stream_bytes);
amg_hip_status amg_hip_devmat_apply(const amg_hip_devmat* m, int32_t op, const double* x,
                                    const double* f, double* out, double omega,
                                    int64_t diag_shift, void* stream);
/* First Jacobi sweep from a zero vector: u_out[i] = 0 + omega*((b[i] - 0)/diag[i] - 0)
 * (diag[i] == 0 leaves 0); bit-identical to amg_hip_dev_jacobi on u_in == 0.   */
amg_hip_status amg_hip_dev_jacobi_from_zero(int64_t nrows, const double* diag,
                                            const double* b, double* u_out,
                                            double omega, void* stream);
/* y[i] += x[i] */
amg_hip_status amg_hip_dev_axpy1(int64_t n, const double* x, double* y, void* stream);
/* *out (device double) = sum_i r[i]^2 ; scratch >= 8 KiB device memory */
amg_hip_status amg_hip_dev_sumsq(int64_t n, const double* r, double* out,
                                 double* scratch, void* stream);

/* ---- peer-to-peer halo exchange over hipIpc (xGMI inside a node) --------------
 * The multi-GPU driver's neighbour exchange without a collective: every rank
 * allocates its exchangeable vectors inside one arena, exports it with hipIpc,
 * maps the arenas of rank-1 / rank+1, and then pushes its boundary unknowns
 * straight into the neighbour's halo slots with stream-ordered copies; arrival
 * and consumption are signalled with 32-bit epoch words in the arenas
 * (hipStreamWriteValue32 / hipStreamWaitValue32: no host synchronisation, no
 * spinning kernel).  Bootstrap (exchanging the 64-byte handles and the layout
 * tables) is done by the host through torch.distributed.                      */
typedef struct amg_hip_arena amg_hip_arena;
amg_hip_status amg_hip_arena_create(int64_t bytes, int32_t device, amg_hip_arena** out);
void amg_hip_arena_destroy(amg_hip_arena* a);
void* amg_hip_arena_base(const amg_hip_arena* a);
/* 64-byte hipIpcMemHandle_t of the arena */
amg_hip_status amg_hip_arena_export(const amg_hip_arena* a, uint8_t handle[64]);
amg_hip_status amg_hip_arena_open_peer(const uint8_t handle[64], void** peer_base);
amg_hip_status amg_hip_arena_close_peer(void* peer_base);

/* One exchange with both neighbours.  Any side may be absent (NULL / 0).
 * push_wait:  wait until the neighbours acknowledged epoch-1 (my previous data
 * has been consumed) -> copy my boundary into their halo slots -> publish
 * `epoch` in their arenas -> wait until their data of `epoch` has landed in
 * mine.  ack: tell the neighbours that the halo data of `epoch` has been
 * consumed (call it after the kernel that read the halos).                    */
typedef struct {
  void* dst_prev; const void* src_prev; int64_t bytes_prev;  /* my first rows -> rank-1's upper halo */
  void* dst_next; const void* src_next; int64_t bytes_next;  /* my last rows  -> rank+1's lower halo */
  uint32_t* data_flag_at_prev;  /* word in rank-1's arena: "data from my next has landed" */
  uint32_t* data_flag_at_next;  /* word in rank+1's arena: "data from my prev has landed" */
  uint32_t* my_data_from_prev;  /* words in MY arena the neighbours publish into          */
  uint32_t* my_data_from_next;
  uint32_t* ack_flag_at_prev;   /* word in rank-1's arena: "my next consumed my data"     */
  uint32_t* ack_flag_at_next;
  uint32_t* my_ack_from_prev;   /* words in MY arena: the neighbour consumed what I sent  */
  uint32_t* my_ack_from_next;
  uint32_t epoch;               /* 1, 2, 3, ... per channel                               */
  int32_t recv_from_prev;       /* 1 when rank-1 sends me data on this channel            */
  int32_t recv_from_next;
} amg_hip_halo_desc;
amg_hip_status amg_hip_halo_push_wait(const amg_hip_halo_desc* d, void* stream);
amg_hip_status amg_hip_halo_ack(const amg_hip_halo_desc* d, void* stream);

/* The same exchange as ONE graph-capturable kernel (no host-side stream memory
 * operations, which cost ~3 us each and cannot be captured): flags are 0/1 words
 * in the receiver's arena (DATA: "your halo slots hold my new values", FREE: "I
 * have consumed them"), accessed with system-scope atomics; spins are bounded
 * (~2 s) and report through *timeout (a device word the host can check).
 * FREE words must be initialised to 1 (amg_hip_fill_u32), DATA words to 0.      */
typedef struct {
  double* dst_prev; const double* src_prev; int64_t cnt_prev;   /* counts in doubles */
  double* dst_next; const double* src_next; int64_t cnt_next;
  uint32_t* my_free_from_prev; uint32_t* my_free_from_next;     /* in MY arena        */
  uint32_t* data_at_prev; uint32_t* data_at_next;               /* in the neighbours' */
  uint32_t* my_data_from_prev; uint32_t* my_data_from_next;     /* in MY arena        */
  int32_t recv_prev, recv_next;
  uint32_t* timeout;
} amg_hip_halo_kdesc;
amg_hip_status amg_hip_halo_exchange_kernel(const amg_hip_halo_kdesc* d, void* stream);
/* after the kernel that consumed the halos: FREE := 1 at the senders */
amg_hip_status amg_hip_halo_ack_kernel(uint32_t* free_at_prev, uint32_t* free_at_next,
                                       void* stream);
/* All-gather by direct pushes: my `cnt` doubles go to offset `off` of every rank's
 * full vector dst[g] (dst[rank] is my own); data_at[g] / free_at[g] are word [rank]
 * of rank g's DATA / FREE block, my_data_from / my_free_from my own blocks
 * (one word per rank).  world <= 16.                                          */
typedef struct {
  int32_t rank, world;
  const double* src; int64_t cnt, off;
  double* dst[16];
  uint32_t* data_at[16];
  uint32_t* free_at[16];
  uint32_t* my_data_from;
  uint32_t* my_free_from;
  uint32_t* timeout;
} amg_hip_gather_kdesc;
amg_hip_status amg_hip_gather_kernel(const amg_hip_gather_kdesc* d, void* stream);
amg_hip_status amg_hip_gather_ack_kernel(const amg_hip_gather_kdesc* d, void* stream);
amg_hip_status amg_hip_fill_u32(uint32_t* dev_ptr, int64_t count, uint32_t value, void* stream);

/* Stream capture helpers for hosts that drive the launchers above from another
 * language: everything enqueued on `stream` between begin and end becomes one
 * executable graph.                                                            */
amg_hip_status amg_hip_capture_begin(void* stream);
amg_hip_status amg_hip_capture_end(void* stream, void** graph_exec);
amg_hip_status amg_hip_graph_launch(void* graph_exec, void* stream);
void amg_hip_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* AMG_HIP_H */
