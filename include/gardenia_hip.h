/* gardenia_hip.h -- C-ABI of libgardenia_hip.so: the MI355X (gfx950) drop-in for the CSR hot
 * path of the GARDENIA benchmark (BFS, PageRank, SpMV, SSSP, TC, CC).
 *
 * Boundary: the reference has no FFI; its "plugin API" is link-time substitution of one
 * XxxSolver symbol per kernel directory (src/<k>/Makefile picks the object;
 * src/<k>/main.cc calls Solver then Verifier).  The entry points below have the raw-array
 * shape of the reference's legacy solvers (m, nnz, row_offsets, column_indices,
 * labels/values -- e.g. src/bfs/topo_base.cu:35, src/pr/vector.cu:83) with the widths of
 * the live Graph class (include/csr_graph.h:50-51: uint64_t offsets, int32_t vertex ids).
 * gardenia_amd/host/solvers.cc (declarations: gardenia_host.hpp) adapts them 1:1 to the live `XxxSolver(Graph&, ...)`
 * signatures, so a main.cc written like the reference's links unchanged.
 *
 * Conventions
 *   - every function returns 0 (GDN_OK) or a negative gdn_status; nothing calls exit()
 *     (the reference's CUDA_SAFE_CALL exits: include/cutil_subset.h:4-12);
 *     gdn_last_error() returns a thread-local message for the last failure.
 *   - "host API": all pointers are HOST pointers owned by the caller; label arrays are
 *     in/out exactly as in the reference mains (dist pre-filled, scores = 1/m, comp[i] = i,
 *     y accumulated into).
 *   - "_dev API": all array pointers are DEVICE pointers on the current device, `stream` is
 *     a hipStream_t passed as void* (NULL = default stream); calls are asynchronous.
 *   - no torch / C++ types in any signature.
 */
#ifndef GARDENIA_HIP_H_
#define GARDENIA_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gdn_status {
  GDN_OK = 0,
  GDN_ERR_INVALID = -1,   /* bad argument (null pointer, m <= 0, source out of range ...) */
  GDN_ERR_NO_DEVICE = -2, /* no HIP device: the product has no CPU fallback */
  GDN_ERR_HIP = -3,       /* a HIP runtime call failed */
  GDN_ERR_OOM = -4,       /* device allocation failed */
  GDN_ERR_OVERFLOW = -5   /* a device worklist overflowed (never silently dropped) */
} gdn_status;

/* Per-call report.  solve_ms has the boundary of the reference's Timer (graph resident,
 * first kernel .. final sync; src/pr/base.cu:108-128, src/bfs/linear_base.cu:62-80):
 * it excludes h2d_ms (graph + label upload) and includes prep_ms only where stated. */
typedef struct gdn_stats {
  int32_t iterations;       /* PR iterations / BFS levels / CC rounds / SSSP phases */
  int32_t reserved;         /* gdn_pr_multi: the exchange that ran (1 RCCL all-gather, 2 peer copies); else 0 */
  double solve_ms;
  double h2d_ms;
  double prep_ms;           /* solver-private layout preparation (tile tables, orientation) */
  uint64_t edges_traversed; /* BFS/SSSP: sum of out-degrees of reached vertices; else nnz*iters */
  double last_error;        /* PR: L1 change of the last iteration; SSSP: edges RELAXED over the solve (list passes count the
                             * out-edges of their list, a dense sweep every edge) -- SURVEY 8d's re-relaxation count */
} gdn_stats;

const char *gdn_last_error(void);
/* Library options.  An option is set process wide before the plan is created or the solver called that it concerns
 * (value NULL = unset); the environment variable of the same name OVERRIDES the stored value, so that a measurement or a test
 * can flip one without touching the caller.  gdn_option_get copies the effective value ("" when unset).
 *
 * THE PUBLIC OPTIONS (all of them; round 6 cut the surface from 145 names to these):
 *   layout / algorithm choices
 *     GDN_PR_LAYOUT     csr | pb      PageRank plans and gdn_pr: merge-path CSR or the propagation-blocked layout (default: by size
 *                                     and by the locality of the gather, gdn_pr_plan_create)
 *     GDN_PR_SQUISH     0             keep vertices without any edge in the per-iteration state (default: squished out)
 *     GDN_PR_FUSED      0 | 1         gdn_pr: the whole solve in one cooperative launch (default below 2^18 edges) or never / always
 *     GDN_PR_BATCH      n             gdn_pr: iterations queued per convergence check on the device (default 8)
 *     GDN_PR_SUM        reference     re-sum rows in the reference's fp32 order behind every pull (see gdn_pr_plan_refsum_info),
 *     GDN_PR_SUM_MIN_DEGREE n           the rows of >= n in-edges (default 0 = every row)
 *     GDN_PRD_LAYOUT    csr | pb      layout of gdn_pr_delta's pull plan;  GDN_PRD_PUSH_DIV n  its push / pull switch (frontier edges < nnz / n)
 *     GDN_SPMV_LAYOUT   csr | pb      SpMV plans;  GDN_SPMV_ONESHOT solve  gdn_spmv builds the blocked layout inside the call (prep_ms)
 *     GDN_BFS_COOP      0 | 1         light BFS levels on the cooperative grid: never / from the first one (default: after 8 light levels)
 *     GDN_BFS_ALPHA_DENSE n           a level goes to the bitmap engines from nnz / n frontier edges on (default 32)
 *     GDN_SSSP_DENSE_IN n             gdn_sssp_run enters its dense sweeps at m / n frontier vertices;  GDN_SSSP_COOP 0 | 1 as GDN_BFS_COOP
 *     GDN_SSSP_UNIT_BFS 0             equal-weight plans keep the sweeps instead of solving through the BFS plan
 *     GDN_CC_SV         1 | fused     Shiloach-Vishkin rounds (src/cc/omp_base.cc) instead of Afforest, as launches or one cooperative kernel
 *     GDN_CC_REVERSE    build         gdn_cc without a reverse graph builds one inside the call (prep_ms)
 *     GDN_TC_FORM       f | a | u | v | bs   forward count on the rank-ordered DAG / the reference's orientation / binary-search intersect
 *     GDN_TC_CORE       k             bit-matrix core of the forward count over the top k degree ranks (0 = none; default by graph size)
 *     GDN_PB_BUILDER    old           layout builds through the round-3 key sort instead of the tiered builder
 *   multi-GPU
 *     GDN_MULTI_DEVICES "0,1,.."      devices of gdn_pr_multi / gdn_spmv_multi;  GDN_MULTI_EXCHANGE rccl | peer  their exchange
 *   placement search of a plan's streamed arrays
 *     GDN_PR_PLACE n / GDN_SPMV_PLACE n   candidates per array (0 = off);  GDN_PR_PLACE_COPIES 1  also the arrays that need a copy
 *   memory
 *     GDN_SCRATCH_KEEP_GB n           cap of the build-scratch cache per device (default 32)
 *     GDN_ALLOC_FENCE   1             every buffer in a block of whole pages of its own, at its end (debugging: an overrun faults)
 *   reports (stderr)
 *     GDN_BFS_TRACE, GDN_SSSP_TRACE   per-level / per-phase timing;  GDN_BFS_TIME_INIT 1  stats.prep_ms = the per-search initialisation,
 *                                     stats.last_error = the closing depth pass (ms)
 *     GDN_PR_PLACE_TRACE, GDN_SPMV_PLACE_TRACE   the placement search, candidate by candidate;  GDN_TRACE_POLICY  gdn_pr's layout model
 *   GDN_TEST_HOOKS=1 makes the library honour its ~50 test hooks (thresholds that force a big-graph code path onto a small graph,
 *   layout variants the suite compares bit for bit; tests/conftest.py sets it, csrc/gdn_common.hpp names the classes).  The A/B
 *   knobs of closed experiments exist only in builds with -DGDN_EXPERIMENTS (make -C gardenia_amd/csrc EXPERIMENTS=1,
 *   tools/build_variant.sh). */
int gdn_option_set(const char *name, const char *value);
int gdn_option_get(const char *name, char *value, int32_t capacity);
int gdn_device_count(int *count);
int gdn_set_device(int device);

/* ------------------------------------------------------------------------------------------
 * Host API: one call == one reference XxxSolver call.
 * ---------------------------------------------------------------------------------------- */

/* replaces BFSSolver(Graph&, int source, DistT* dist): src/bfs/bfs.h:43; callers
 * src/bfs/main.cc:22.  dist: in = MYINFINITY (1e9) everywhere (main.cc:21), out = hop count,
 * unreachable stays 1e9.  in_rowptr/in_colidx nullable: when given (reverse or symmetrized
 * graph) the direction-optimising path (src/bfs/omp_beamer.cc:97) is used. */
int gdn_bfs(int32_t m, uint64_t nnz, const uint64_t *out_rowptr, const int32_t *out_colidx,
            const uint64_t *in_rowptr, const int32_t *in_colidx, int32_t source, int32_t *dist,
            gdn_stats *stats);

/* replaces PRSolver(Graph&, ScoreT* scores): src/pr/pr.h:31; caller src/pr/main.cc:19.
 * scores: in = 1/m (main.cc:17-18), out = PageRank.  Reference constants: damping kDamp 0.85
 * (pr.h:6), epsilon EPSILON 1e-4 (pr.h:5), max_iter MAX_ITER 100 (pr.h:12).  out_degree[v] =
 * Graph::get_degree(v) on the OUT-CSR (csr_graph.h:295); the gather runs over the IN-CSR. */
int gdn_pr(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx,
           const int32_t *out_degree, float *scores, float damping, double epsilon, int32_t max_iter,
           gdn_stats *stats);

/* The per-iteration L1 changes of the calling thread's last gdn_pr / gdn_pr_multi solve: what the reference prints as it
 * iterates (" %2d    %lf", src/pr/omp_base.cc:35; the only golden it ships is that trace,
 * test/reference/graph-pr.mtx.out:13-27).  *n = iterations recorded; the first min(*n, capacity) go to diff. */
int gdn_pr_last_trace(int32_t capacity, int32_t *n, double *diff);
/* The layout the calling thread's last gdn_pr ran on (GDN_LAYOUT_CSR or GDN_LAYOUT_PB below; -1 before the first call).
 * gdn_pr picks it by predicted wall time -- the blocked layout costs ~100 ps per edge to build (stats.prep_ms) and saves
 * ~8 ps per edge and iteration -- unless the option GDN_PR_LAYOUT forces c(sr) or p(b); GDN_PR_ONESHOT=solve: the blocked
 * layout from 2^22 edges on whatever the iteration count.  No counterpart in the reference (src/pr/pr.h:31 has one
 * layout per binary). */
int gdn_pr_last_layout(int32_t *layout);

/* PRSolver on `ngpus` devices of this node (SURVEY 8b `gdn_pr(..., int ngpus, gdn_stats*)`, 8e; supersedes the
 * edge-list slicing stub of include/graph_gpu.h:145-165 -- the reference has no multi-GPU hot path).  Same arguments
 * and results as gdn_pr.  One process, one host thread per device; the rows of the in-CSR are cut into ngpus contiguous
 * vertex ranges of about nnz / ngpus edges (binary search on the row offsets); per iteration every device pulls its
 * rows and the slices of the next contribution vector are exchanged -- an in-place RCCL all-gather over xGMI (librccl is
 * dlopen'ed on first use), or peer copies pipelined behind the pull kernels (GDN_MULTI_EXCHANGE=p2p, and whenever a
 * device is listed twice: RCCL refuses that, and it is how a 1-GPU box exercises this path); the 8-byte L1 changes are
 * summed on the host in rank order.  devices: ngpus HIP device ids (NULL = 0 .. ngpus-1).  ngpus > m is clamped.
 * With the propagation-blocked layout (shards of >= 2^22 edges, or GDN_PR_LAYOUT=pb) the scores are bit-identical to
 * gdn_pr's for every ngpus.  stats.reserved reports the exchange that ran. */
int gdn_pr_multi(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx,
                 const int32_t *out_degree, float *scores, float damping, double epsilon, int32_t max_iter,
                 int32_t ngpus, const int32_t *devices, gdn_stats *stats);
/* the vertex ranges gdn_pr_multi / gdn_spmv_multi cut a host CSR into: bounds[ngpus + 1] (range r = [bounds[r],
 * bounds[r+1])), *chunk (nullable) = slot length of the padded vertex space */
int gdn_multi_ranges(int32_t m, const uint64_t *rowptr, int32_t ngpus, int32_t *bounds, int32_t *chunk);

/* replaces SpmvSolver(Graph&, const ValueT* Ax, const ValueT* x, ValueT* y):
 * src/spmv/spmv.h:29; caller src/spmv/main.cc:39.  y[i] += sum_k Ax[k]*x[Aj[k]] over the rows
 * of (Ap, Aj) = g.in_rowptr()/g.in_colidx() (src/spmv/omp_base.cc:10-11). */
int gdn_spmv(int32_t m, uint64_t nnz, const uint64_t *Ap, const int32_t *Aj, const float *Ax,
             const float *x, float *y, gdn_stats *stats);

/* SpmvSolver on `ngpus` devices (SURVEY 8b `gdn_spmv(..., ngpus, ...)`): row ranges as for gdn_pr_multi, x replicated
 * (one upload per device: a single multiply needs no exchange), every device multiplies its rows and returns its slice
 * of y.  Same results as gdn_spmv bit for bit (the merge-path row sums do not depend on the cut). */
int gdn_spmv_multi(int32_t m, uint64_t nnz, const uint64_t *Ap, const int32_t *Aj, const float *Ax, const float *x,
                   float *y, int32_t ngpus, const int32_t *devices, gdn_stats *stats);

/* replaces SSSPSolver(Graph&, int source, DistT* weight, DistT* dist, int delta):
 * src/sssp/sssp.h:47; caller src/sssp/main.cc:27.  dist: in = kDistInf (INT_MAX,
 * sssp.h:46), out = shortest distance.  weight[nnz] parallel to colidx, delta >= 1: the bucket width the schedule starts
 * with (src/sssp/omp_base.cc:12 takes it as a tuning parameter too); behind a run of light buckets the schedule widens it
 * by itself (option GDN_SSSP_ADAPT=0: the caller's width throughout).  Distances are exact whatever the widths. */
int gdn_sssp(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx,
             const int32_t *weight, int32_t source, int32_t delta, int32_t *dist, gdn_stats *stats);

/* replaces TCSolver(Graph&, uint64_t& total): src/tc/tc.h:7; caller src/tc/main.cc:17.
 * oriented == 0: (rowptr, colidx) is a symmetric graph and the DAG orientation of
 * src/common/graph.cc:67-113 is applied on the device first (as `Graph g(prefix, USE_DAG)`
 * does, src/tc/main.cc:12); oriented != 0: already a DAG. */
int gdn_tc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, int32_t oriented,
           uint64_t *total, gdn_stats *stats);

/* Betweenness centrality from ONE source == BCSolver(g, source, scores) of src/bc/bc.h:37 (src/bc/main.cc:22;
 * SURVEY 8f rank 4): Brandes' forward BFS with 32-bit path counts + backward dependency sweep, scores[v] += delta[v],
 * then every score divided by the largest (all 0/0 = NaN when nothing lies between, like the reference).
 * scores: m floats, in/out (the reference main zero-fills).  stats.iterations = BFS levels. */
int gdn_bc(int32_t m, uint64_t nnz, const uint64_t *out_rowptr, const int32_t *out_colidx, int32_t source, float *scores,
           gdn_stats *stats);

/* Delta PageRank == the PRSolver of src/pr/delta.cu:140 / src/pr/omp_delta.cc:52 (raw-array signature
 * PRSolver(m, nnz, in_row_offsets, in_column_indices, out_row_offsets, out_column_indices, degrees, scores); degrees are
 * the out-CSR's row lengths and are taken from it).  scores in: 1/m (src/pr/main.cc:17), out.  An iteration pulls the
 * deltas of all vertices over the in-CSR while the frontier (|delta| > epsilon2 * score; pr.h:8 epsilon2 = 1e-3) holds
 * at least m / push_div vertices (8 in delta.cu:178, 10 in omp_delta.cc:69) and pushes the frontier's deltas over the
 * out-CSR otherwise; stops on an empty frontier, after max_iter iterations or when the L1 norm of the deltas < epsilon.
 * stats.iterations = iterations executed (omp_delta.cc:105 prints one more), stats.last_error = the last L1 norm. */
int gdn_pr_delta(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx, const uint64_t *out_rowptr,
                 const int32_t *out_colidx, float *scores, float damping, double epsilon, float epsilon2, int32_t max_iter,
                 int32_t push_div, gdn_stats *stats);

/* replaces CCSolver(Graph&, CompT* comp): src/cc/cc.h:28; caller src/cc/main.cc:16.
 * comp: in = i (main.cc:15), out = component label = minimum vertex id of the (weakly)
 * connected component (fixpoint of src/cc/omp_base.cc:24-43).  in_* nullable (directed
 * graphs pass the reverse graph like src/cc/omp_afforest.cc:64-76). */
int gdn_cc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx,
           const uint64_t *in_rowptr, const int32_t *in_colidx, int32_t *comp, gdn_stats *stats);

/* Device buffers for callers of the _dev API that do not bring their own allocator (blocking). */
int gdn_dev_alloc(uint64_t bytes, void **d_ptr);
int gdn_dev_free(void *d_ptr);
/* The library keeps the temporaries of its graph / layout builds in a process-level cache of device blocks (a fresh
 * multi-GB hipMalloc stalls for seconds on this runtime when memory of that size was freed shortly before; up to
 * GDN_SCRATCH_KEEP_GB = 32 GB per device stay cached).  gdn_dev_trim hands every cached block back to the driver -- for a
 * caller that shares the device with another allocator (torch, a second library) and wants the memory; the library does so
 * itself before it reports GDN_ERR_OOM.  *freed_bytes (nullable) = what went back.  Blocking. */
int gdn_dev_trim(uint64_t *freed_bytes);
/* Sets a block of device memory aside (bytes 0: gives it back).  Called as the FIRST device call of a process -- before any
 * build has allocated and freed memory -- it gives the first blocked PageRank plan whose per-iteration scratch (`vals`) fits
 * a block of pages no build has used; the plan owns it from then on.  Measurement hook of DESIGN.md 4.1 (placement). */
int gdn_dev_reserve(uint64_t bytes);
int gdn_dev_upload(void *d_dst, const void *h_src, uint64_t bytes);
int gdn_dev_download(void *h_dst, const void *d_src, uint64_t bytes);

/* Stable radix sort of n 64-bit keys by their bits [begin_bit, end_bit) on the device (gdn_sort.hip: the primitive under
 * every graph and layout build here, where the reference's builds call std::sort and its kernels CUB,
 * include/worklistc.h:6).  d_keys holds the input, d_tmp n more keys; *d_sorted is whichever of the two holds the
 * result.  Blocking. */
int gdn_sort_u64_dev(uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int32_t begin_bit, int32_t end_bit, uint64_t **d_sorted);

/* The worklist push of the traversal kernels on its own (replaces Worklist::push / Worklist2::push_1item,
 * include/worklistc.h:44-50, :66-89): every index i < n with d_flags[i] != 0 is appended to d_queue, one counter add per
 * wavefront step (staged = 0) or per ~256 items collected in LDS (staged != 0).  *d_count = items pushed, including the
 * ones beyond `capacity`, which are not stored and set *d_overflow (the reference drops them silently, :46-47).  Blocking. */
int gdn_worklist_filter_dev(const int32_t *d_flags, int32_t n, int32_t staged, int32_t *d_queue, uint32_t capacity,
                            uint32_t *d_count, uint32_t *d_overflow);

/* ------------------------------------------------------------------------------------------
 * Resident graphs (the reference re-uploads per Solver call: src/bfs/linear_base.cu:42-49).
 * ---------------------------------------------------------------------------------------- */
typedef struct gdn_graph gdn_graph;

int gdn_graph_upload(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx,
                     gdn_graph **out);
/* gdn_graph_upload checks the arrays on the device (offsets ascending within [0, nnz], column ids within [0, m)) and
 * returns GDN_ERR_INVALID for a malformed CSR instead of letting a solver read out of bounds.  gdn_graph_upload_rows:
 * rows [row_lo, row_hi) of a host CSR as an independent graph (offsets rebased, column ids < n_cols unchanged) -- the
 * shard one device of gdn_pr_multi / gdn_spmv_multi holds.  gdn_graph_validate: the same check for wrapped arrays. */
int gdn_graph_upload_rows(int32_t m, const uint64_t *rowptr, const int32_t *colidx, int32_t row_lo, int32_t row_hi,
                          int32_t n_cols, gdn_graph **out);
int gdn_graph_validate(const gdn_graph *g, int32_t n_cols);
/* wrap device arrays owned by the caller (e.g. torch tensors); nothing is copied (and nothing checked) */
int gdn_graph_wrap_dev(int32_t m, uint64_t nnz, const uint64_t *d_rowptr, const int32_t *d_colidx,
                       gdn_graph **out);
int gdn_graph_free(gdn_graph *g);
int gdn_graph_info(const gdn_graph *g, int32_t *m, uint64_t *nnz, const uint64_t **d_rowptr,
                   const int32_t **d_colidx);
/* out_degree[v] = rowptr[v+1]-rowptr[v] as int32 (Graph::get_degree, csr_graph.h:295) */
int gdn_graph_degrees_dev(const gdn_graph *g, int32_t *d_degree, void *stream);
/* reverse graph, csr_graph.h:170-194 build_reverse_graph (rows ascending) */
int gdn_graph_transpose(const gdn_graph *g, gdn_graph **out);
/* undirected closure: every edge in both directions, duplicates dropped, rows ascending (the loader's
 * symmetrize = true, csr_graph.h:112-115) */
int gdn_graph_symmetrize(const gdn_graph *g, gdn_graph **out);
/* rows [row_lo,row_hi) as an independent graph (column ids stay global): the vertex-range
 * shard one GPU holds in the multi-GPU PageRank/SpMV path */
int gdn_graph_slice_rows(const gdn_graph *g, int32_t row_lo, int32_t row_hi, gdn_graph **out);
/* nnz-balanced vertex ranges of a resident graph (binary search on row_offsets): bounds[world + 1],
 * bounds[0] = 0, bounds[world] = m, range r = [bounds[r], bounds[r+1]) holds about nnz / world edges and at least one row */
int gdn_graph_balanced_ranges(const gdn_graph *g, int32_t world, int32_t *bounds);
/* rows of range `rank` as an independent graph whose column ids are moved into the PADDED vertex space of a sharded
 * run: vertex v of range r -> r * chunk + (v - bounds[r]) (chunk >= the longest range).  Every rank's slice of a
 * replicated per-vertex vector then starts at a multiple of chunk: equal all-gather slots for unequal ranges. */
int gdn_graph_slice_padded(const gdn_graph *g, int32_t world, const int32_t *bounds, int32_t chunk, int32_t rank,
                           gdn_graph **out);
/* The second half of gdn_graph_slice_padded on a shard that was never part of a whole graph (gdn_rmat_build_range +
 * gdn_pr_squish_range): its column ids, in the vertex space `bounds` cuts, moved into the padded space -- in place. */
int gdn_graph_pad_columns(gdn_graph *shard, int32_t world, const int32_t *bounds, int32_t chunk);
/* Device ingest: edge list (host arrays, 0-based ids) -> resident CSR with the clean-up of the reference
 * loader (include/csr_graph.h:108 self loops dropped, :127 rows sorted ascending, :132-143 duplicates
 * dropped; symmetrize != 0 also inserts every reverse edge, :112-115).  One radix sort on the device
 * replaces the per-row std::sort + O(deg^2) erase loop.  Ids outside [0,m) are an error. */
int gdn_graph_from_edges(int32_t m, uint64_t n_edges, const int32_t *src, const int32_t *dst, int32_t symmetrize,
                         gdn_graph **out);
/* download to caller-provided host arrays ((m+1) x u64, nnz x i32) */
int gdn_graph_download(const gdn_graph *g, uint64_t *rowptr, int32_t *colidx);

/* Device R-MAT generator + CSR builder (measurement input): Graph500 recipe of
 * include/generator.h:81-114 on the counter-based RNG documented in gardenia_amd/graphio.py,
 * cleaned like csr_graph.h:108,127,132-143 (self loops and duplicates dropped, rows ascending).
 * Either output may be NULL. */
int gdn_rmat_build(int32_t scale, int32_t edge_factor, uint64_t seed, int32_t permute,
                   gdn_graph **out_csr, gdn_graph **in_csr);
/* ONE vertex range of gdn_rmat_build_ex's graph, for a rank of a sharded run that must not hold the whole graph (17 GB of keys
 * at scale 27, 137 GB at scale 30): the in-CSR rows [v_lo, v_hi) -- row i = vertex v_lo + i, column ids global -- of exactly
 * the graph gdn_rmat_build_ex(scale, n_edges, a, b, c, seed, flags) builds.  Every edge of the stream is generated (twice: a
 * counting pass sizes the key buffer), only the keys whose destination lies in the range are kept, sorted and cleaned of
 * duplicates as there: memory = the range's own edges + 8 (2^scale + 1) bytes of offsets.  d_out_degree_partial (nullable;
 * int32[2^scale], zeroed by the caller) += 1 per kept edge at its source: summed over the ranges of a partition -- one
 * all-reduce across the ranks -- it is the graph's out-degree vector (the DISTRIBUTED degree count).  flags: GDN_RMAT_PERMUTE
 * only (compaction is a property of the whole graph: gdn_pr_squish_range below does it for the ranks together). */
int gdn_rmat_build_range(int32_t scale, uint64_t n_edges, double a, double b, double c, uint64_t seed, int32_t flags,
                         int32_t v_lo, int32_t v_hi, gdn_graph **in_rows, int32_t *d_out_degree_partial);
/* The same generator with its knobs open: n_edges draws (not a multiple of 2^scale necessarily), quadrant probabilities
 * a / b / c (d = 1 - a - b - c; include/generator.h:88-90 fixes .57 / .19 / .19), flags GDN_RMAT_PERMUTE (ids shuffled,
 * generator.h:52-62) and GDN_RMAT_COMPACT: the ids that occur in no edge are dropped and the others renumbered in ascending
 * order, so the graph has few isolated vertices like the real graphs of BASELINE configs 2 and 4 (soc-LiveJournal1, com-Orkut,
 * datasets/test.mk:5,8 -- wget lines); *_csr then have fewer than 2^scale rows (gdn_graph_info).  gardenia_amd/graphio.py
 * (rmat_graph_ex, lj_like / orkut_like) holds the bit-identical numpy twin and the two stand-in recipes. */
#define GDN_RMAT_PERMUTE 1
#define GDN_RMAT_COMPACT 2
int gdn_rmat_build_ex(int32_t scale, uint64_t n_edges, double a, double b, double c, uint64_t seed, int32_t flags,
                      gdn_graph **out_csr, gdn_graph **in_csr);

/* ------------------------------------------------------------------------------------------
 * _dev API -- PageRank (graph resident, vectors are device pointers)
 * ---------------------------------------------------------------------------------------- */
typedef struct gdn_pr_plan gdn_pr_plan;

/* in_csr: IN-CSR of the m_local rows this device owns (global row ids row_base ..
 * row_base+m_local-1; column ids in [0,m_global)).  d_out_degree: m_local entries.
 * Single GPU: row_base = 0, m_global = m_local. */
/* layout: how the plan stores the edges of its rows.
 *   GDN_LAYOUT_CSR  merge-path over the caller's CSR (no copy; bitwise reproducible sums)
 *   GDN_LAYOUT_PB   propagation blocking: edges re-grouped into (source chunk, destination bin)
 *                   tiles with 16-bit local ids, both slices LDS resident (the reference's
 *                   precedent: include/prop_blocking.h, src/pr/push_pb.cu; built once per graph,
 *                   outside the timed loop like segmenting() in src/pr/partition.cu)
 *   GDN_LAYOUT_AUTO PB for graphs with >= 2^22 edges (env GDN_PR_LAYOUT=csr|pb overrides)
 *   GDN_LAYOUT_PB_SQUISHED (whole graphs only; never picked by AUTO)  the PB layout in a vertex space WITHOUT the
 *                   vertices that have neither in- nor out-edges (53 % of RMAT-27): such a vertex keeps the constant
 *                   base score and nobody reads its contribution, so the per-vertex STATE arrays of an iteration
 *                   (d_scores, d_contrib_in/out of the calls below) have gdn_pr_plan_state_size() entries, state
 *                   index k <-> the k-th live vertex in id order; gdn_pr_import_dev / gdn_pr_export_dev convert
 *                   between the caller's m-entry score vector and the state at the boundary of a solve.  Same bits
 *                   as GDN_LAYOUT_PB for every vertex. */
enum { GDN_LAYOUT_AUTO = -1, GDN_LAYOUT_CSR = 0, GDN_LAYOUT_PB = 1, GDN_LAYOUT_PB_SQUISHED = 2 };
/* From 3 x 2^28 edges on, creating a PB plan ends with a PLACEMENT SEARCH: fresh allocations of the streamed
 * arrays are timed on scratch vectors and the fastest ones kept -- 0.4-2.5 s more plan build for up to 9 % faster
 * iterations, same results.  Option GDN_PR_PLACE=<tries per array> (default 3, 0 = off); the one-shot gdn_pr never runs it.
 * gdn_spmv_plan_create does the same (GDN_SPMV_PLACE). */
int gdn_pr_plan_create(const gdn_graph *in_csr, const int32_t *d_out_degree, int32_t m_global,
                       int32_t row_base, int32_t layout, gdn_pr_plan **plan);
/* log_blk: 0 for CSR; for PB 100*log2(chunk ids) + log2(bin rows) */
int gdn_pr_plan_layout(const gdn_pr_plan *plan, int32_t *layout, int32_t *log_blk);
/* PB accumulates in 2^-62 fixed point (contributions must lie in [0,1]): returns
 * GDN_ERR_OVERFLOW if any launch on this plan saw a value outside that range (blocking) */
/* hub tier of the PB layout (0 / 0 when the plan has none): number of hub sources and of the edges that leave them */
int gdn_pr_plan_hubs(const gdn_pr_plan *plan, int32_t *n_hubs, uint64_t *hub_edges);
/* mid tiers of the PB layout (the degree levels below the hubs, read by phase B as 32-bit (source, row) records):
 * number of tiers, their sources and their edges (0 / 0 / 0 when the plan has none) */
int gdn_pr_plan_mid(const gdn_pr_plan *plan, int32_t *n_tiers, int32_t *n_sources, uint64_t *n_edges);
/* Placement of a blocked plan's streamed arrays: those named in `what` (1 vals, 2 U, 4 G, 8 V, 16 hub records, 32 mid-tier
 * records, 64 per-iteration tables) are copied into fresh allocations.  Measurement hook of DESIGN.md 4.1 (where hipMalloc
 * puts these arrays moves an iteration by up to 8 %). */
int gdn_pr_plan_move(gdn_pr_plan *plan, uint32_t what);
/* destination bins of the PB layout = workgroups of the accumulate phase (0 for the CSR layout): a driver that cuts an
 * iteration into row-range parts (gdn_pr_pull_rows_dev) keeps every part at a whole wave of workgroups or more */
int gdn_pr_plan_bins(const gdn_pr_plan *plan, int32_t *n_bins);
int gdn_pr_plan_check(gdn_pr_plan *plan);
/* Option GDN_PR_SUM=reference (read by gdn_pr_plan_create / gdn_pr): behind every pull the rows of at least
 * GDN_PR_SUM_MIN_DEGREE in-edges (default 0: every row) are summed AGAIN in the order of src/pr/omp_base.cc:27-30 -- one fp32
 * addition per in-edge, in CSR order -- and their scores, next contributions and the L1 change rewritten: those rows then carry
 * the reference's bits (with every row selected the whole solve does).  Two passes per pull: the contributions those rows read
 * are staged in row order through LDS slices of the contribution vector (no gather leaves a CU), then every row is summed by
 * scans of parity functions, not by a chain of additions (csrc/gdn_seqsum.hpp).  The plan keeps 10 bytes per
 * selected in-edge.  This call reports what a plan of that mode re-sums: rows, the longest of them, entries (in-edges) and
 * launches per pull; all 0 for a plan without the mode. */
int gdn_pr_plan_refsum_info(const gdn_pr_plan *plan, int32_t *rows, int32_t *longest_row, uint64_t *entries, int32_t *groups);
int gdn_pr_plan_free(gdn_pr_plan *plan);
/* entries of the per-vertex state arrays (scores, contrib) the iteration calls of this plan work on: m_local, or the
 * live vertices of a GDN_LAYOUT_PB_SQUISHED plan */
int gdn_pr_plan_state_size(const gdn_pr_plan *plan, int32_t *m_state);
/* caller's score vector (m entries) -> state (a copy for plans in the caller's vertex space); also records the L1
 * change the vertices outside the state contribute to the FIRST iteration after the import (they move to the base
 * score (1 - damping) / m): gdn_pr_import_diff returns it (blocking; 0 for plans in the caller's vertex space) */
int gdn_pr_import_dev(gdn_pr_plan *plan, const float *d_scores, float *d_state, float damping, void *stream);
int gdn_pr_import_diff(gdn_pr_plan *plan, double *dead_diff);
/* state -> caller's score vector: live vertices from the state, the others get the base score */
int gdn_pr_export_dev(gdn_pr_plan *plan, const float *d_state, float *d_scores, float damping, void *stream);
/* The same relabelling as an object of its own, for drivers that shard the graph (bench.py --gpus N): squish the whole
 * in-CSR once, cut `graph` (relabelled in-CSR, m_state rows, owned by the object) into vertex ranges with
 * gdn_graph_slice_rows, build ordinary plans on the shards (d_degrees: out-degrees in state order) and tell them the
 * original vertex count with gdn_pr_plan_set_base (base score (1 - d) / m).  import: caller's m-entry scores -> state;
 * dead_diff (nullable, makes the call blocking) = L1 change of the vertices outside the state in the first iteration. */
/* gdn_pr_squish_range: the same relabelling for a graph held as vertex ranges (gdn_rmat_build_range).  d_in_degree /
 * d_out_degree: the WHOLE graph's degree vectors (int32[m], all-reduced by the caller); `rows` = the in-CSR rows
 * [v_lo, v_lo + rows) with global column ids, relabelled IN PLACE: its rows become the live vertices of the range (a vertex
 * is live when it has an edge in either direction), its column ids state ids (position among the live vertices, ascending).
 * state_bounds[i] = live vertices below raw_bounds[i], i < n_bounds (raw range cuts -> state range cuts). */
int gdn_pr_squish_range(gdn_graph *rows, int32_t v_lo, const int32_t *d_in_degree, const int32_t *d_out_degree, int32_t m,
                        int32_t n_bounds, const int32_t *raw_bounds, int32_t *state_bounds);
typedef struct gdn_pr_squish gdn_pr_squish;
int gdn_pr_squish_create(const gdn_graph *in_csr, const int32_t *d_out_degree, gdn_pr_squish **sq);
int gdn_pr_squish_info(const gdn_pr_squish *sq, int32_t *m_orig, int32_t *m_state, const gdn_graph **graph,
                       const int32_t **d_degrees);
int gdn_pr_squish_degrees_dev(const gdn_pr_squish *sq, int32_t *d_degrees /* m_state */, void *stream);
int gdn_pr_squish_import_dev(gdn_pr_squish *sq, const float *d_scores, float *d_state, float damping, double *dead_diff,
                             void *stream);
int gdn_pr_squish_export_dev(gdn_pr_squish *sq, const float *d_state, float *d_scores, float damping, void *stream);
int gdn_pr_squish_free(gdn_pr_squish *sq);
int gdn_pr_plan_set_base(gdn_pr_plan *plan, int32_t m_base);
/* contrib[row_base+v] = scores[v]/out_degree[v]  (src/pr/base.cu:14 contrib) */
int gdn_pr_contrib_dev(gdn_pr_plan *plan, const float *d_scores, float *d_contrib, void *stream);
/* one fused pull iteration (src/pr/base.cu:19 pull_step + :37 l1norm + next :14 contrib):
 *   sum = SUM contrib_in[col]; new = base + damping*sum; diff += |new - scores[v]|;
 *   scores[v] = new; contrib_out[row_base+v] = new/out_degree[v]
 * contrib_in/out are m_global-sized and must be different buffers; d_diff receives the L1
 * change of the local rows (double, deterministic reduction order). */
int gdn_pr_pull_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                    double *d_diff, float damping, void *stream);
/* The same iteration cut into parts by LOCAL row range, so that a multi-GPU driver can send the finished rows of
 * part j (RCCL all-gather on another stream) while part j+1 is computed.  Parts are issued in ascending row order
 * on one stream; flags: GDN_PR_PART_FIRST on the first part (it also runs the source-side phase of the whole
 * iteration), GDN_PR_PART_LAST on the last (it covers every remaining row and reduces the L1 change into d_diff).
 * After part j has run, scores[v] and contrib_out[row_base+v] are final for every v < row_end of that part (a part
 * may finish rows beyond its row_end early).  gdn_pr_pull_dev == one part with both flags. */
#define GDN_PR_PART_FIRST 1
#define GDN_PR_PART_LAST 2
int gdn_pr_pull_rows_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                         double *d_diff, float damping, int32_t row_begin, int32_t row_end, int32_t flags,
                         void *stream);
/* The same iteration with its rows becoming final PART BY PART inside ONE launch per phase (no launch tails between the
 * parts: four row-range launches cost RMAT-27's accumulate phase 2.53 -> 3.13 ms on one rank).  row_end[j] (ascending, local
 * rows) ends part j; the last part covers every remaining row.  The accumulate launch walks the bins of part 0 first
 * (largest first inside a part); a workgroup that has finished a bin publishes its rows device-wide and adds a ticket to
 * its part.  gdn_pr_wait_part_dev queues, on ANOTHER stream, a one-wave kernel that ends once part `part` of the pull
 * queued LAST on this plan is final: scores[v] and contrib_out[row_base+v] for every v < row_end[part] -- whatever is queued
 * behind it on that stream (the all-gather of those rows) overlaps the accumulation of the later parts.  A waiter gives up
 * after ~4 s of device time and gdn_pr_plan_check then reports it.  n_parts <= 8.  Not available inside gdn_pr's own loop
 * or under GDN_PR_SUM=reference.  (The reference has no counterpart: its multi-GPU vestige is include/graph_gpu.h:145-165.) */
int gdn_pr_pull_parts_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                          double *d_diff, float damping, int32_t n_parts, const int32_t *row_end, void *stream);
int gdn_pr_wait_part_dev(gdn_pr_plan *plan, int32_t part, void *stream);
/* Per-launch HIP-event timing of the iteration's kernels on the launch stream.  reset != 0 arms
 * it for up to max_launches launches; reset == 0 waits for the events and reports summed
 * durations in total_ms[2] (CSR: [0] = merge-path tile kernel; PB: [0] = expand, [1] =
 * accumulate) and the launch count (bench.py roofline.achieved). */
int gdn_pr_plan_kernel_time(gdn_pr_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
                            int32_t *launches);
/* algorithmic bytes of one pull iteration on this plan:
 * 8(m+1) + 4 nnz [colidx] + 4 nnz [contrib gather] + 16 m */
uint64_t gdn_pr_iter_bytes(const gdn_pr_plan *plan);

/* _dev API -- SpMV */
typedef struct gdn_spmv_plan gdn_spmv_plan;
/* layout as for PageRank.  GDN_LAYOUT_PB keeps Ax inside the plan in tile order (pass d_Ax here; the
 * d_Ax argument of gdn_spmv_dev is then ignored) and accumulates in signed fixed point whose
 * power-of-two scale is derived per call from max|Ax|*max|x|*max row length on the device: a product is converted with
 * an absolute error below one unit = that bound * 2^-61, so every product within 2^-37 of the bound is exact; a row that
 * received a smaller one and whose sum stays below 2^32 units is recomputed from the CSR in fp32 like the reference's
 * loop (csr and d_Ax therefore have to outlive the plan / be passed to gdn_spmv_dev); for every other row the relative
 * error is at most (its inexact products) * 2^-32.  Sums are bitwise reproducible.
 * GDN_LAYOUT_PB with d_Ax == NULL builds the plan of the PATTERN matrix (every nonzero 1): no value stream at all,
 * 8 instead of 12 bytes per nonzero (what delta PageRank's pull multiplies with). */
int gdn_spmv_plan_create(const gdn_graph *csr, const float *d_Ax /*nullable for CSR*/, int32_t layout,
                         gdn_spmv_plan **plan);
/* the same for a ROW SHARD of a larger matrix: csr holds rows [lo,hi) with global column ids, x has n_cols entries
 * (the vertex-range shard one GPU multiplies in the multi-GPU path; y has csr->m entries) */
int gdn_spmv_plan_create_cols(const gdn_graph *csr, const float *d_Ax /*nullable for CSR*/, int32_t n_cols,
                              int32_t layout, gdn_spmv_plan **plan);
int gdn_spmv_plan_check(gdn_spmv_plan *plan);
/* record tiers of the PB layout (hub and mid-tier columns whose nonzeros phase B reads directly as (record, Ax) pairs):
 * hub columns, number of mid tiers, nonzeros in all tiers (0 / 0 / 0 when the plan has none) */
int gdn_spmv_plan_tiers(const gdn_spmv_plan *plan, int32_t *n_hubs, int32_t *n_mid_tiers, uint64_t *tier_edges);
int gdn_spmv_plan_free(gdn_spmv_plan *plan);
/* y[v] += SUM Ax[k]*x[Aj[k]]   (src/spmv/base.cu:13, warp.cu:26, vector.cu:27 superseded) */
int gdn_spmv_dev(gdn_spmv_plan *plan, const float *d_Ax, const float *d_x, float *d_y, void *stream);
int gdn_spmv_plan_kernel_time(gdn_spmv_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
                              int32_t *launches);
uint64_t gdn_spmv_bytes(const gdn_spmv_plan *plan);

/* _dev API -- BFS / SSSP / CC / TC on resident graphs; label arrays are device pointers and
 * are (re)initialised by the call.  Synchronous (they read frontier sizes every level like
 * src/bfs/linear_base.cu:73). */
int gdn_bfs_dev(const gdn_graph *out_csr, const gdn_graph *in_csr /*nullable*/, int32_t source,
                int32_t *d_dist, gdn_stats *stats);
/* resident graph, device score vector (in/out) */
int gdn_bc_dev(const gdn_graph *g, int32_t source, float *d_scores, gdn_stats *stats);
/* Resident plan for many sources on one graph: depths from the BFS plan, and the heavy levels of both phases as
 * propagation-blocked sweeps (forward: exact integer path counts on a layout of the in-CSR; backward: the dependencies in
 * 2^-62 fixed point on a layout of the out-CSR -- deterministic, inside the reference verifier's tolerance, not bit-equal to
 * its sequential fp32 sums on those levels).  gin: the in-CSR (nullable: transposed here).  d_scores in/out like gdn_bc_dev. */
typedef struct gdn_bc_plan gdn_bc_plan;
int gdn_bc_plan_create(const gdn_graph *g, const gdn_graph *gin, gdn_bc_plan **plan);
int gdn_bc_run(gdn_bc_plan *plan, int32_t source, float *d_scores, gdn_stats *stats);
int gdn_bc_plan_free(gdn_bc_plan *plan);
/* Delta PageRank (gdn_pr_delta above) on resident graphs.  layout: GDN_LAYOUT_AUTO / _CSR / _PB of the pull's SpMV plan
 * (pattern matrix of the in-CSR).  d_scores: m floats in (1/m) / out.  gdn_pr_delta_trace copies what the last run did per
 * iteration -- L1 norm, frontier size after it, mode (0 pull, 1 push with atomics, 3 push semantics executed as a pull of the
 * frontier's terms only: heavy frontiers) -- into arrays of `capacity` entries (nullable) and
 * sets *n to the number of iterations. */
typedef struct gdn_pr_delta_plan gdn_pr_delta_plan;
int gdn_pr_delta_plan_create(const gdn_graph *in_csr, const gdn_graph *out_csr, int32_t layout, gdn_pr_delta_plan **plan);
int gdn_pr_delta_run(gdn_pr_delta_plan *plan, float *d_scores, float damping, double epsilon, float epsilon2,
                     int32_t max_iter, int32_t push_div, gdn_stats *stats);
int gdn_pr_delta_trace(const gdn_pr_delta_plan *plan, int32_t capacity, int32_t *n, double *diff, int32_t *items,
                       int32_t *mode);
int gdn_pr_delta_plan_free(gdn_pr_delta_plan *plan);
/* Reusable BFS state for many searches on one resident graph.  dense != 0 (needs in_csr) also
 * builds the propagation-blocked layout of the in-CSR once, so that heavy levels run as one
 * streaming sweep over all in-edges instead of a bottom-up step (built outside the timed search,
 * like the reverse graph the reference builds in its Graph constructor, csr_graph.h:236-240). */
typedef struct gdn_bfs_plan gdn_bfs_plan;
int gdn_bfs_plan_create(const gdn_graph *out_csr, const gdn_graph *in_csr /*nullable*/, int32_t dense,
                        gdn_bfs_plan **plan);
int gdn_bfs_plan_free(gdn_bfs_plan *plan);
int gdn_bfs_run(gdn_bfs_plan *plan, int32_t source, int32_t *d_dist, gdn_stats *stats);
int gdn_sssp_dev(const gdn_graph *csr, const int32_t *d_weight, int32_t source, int32_t delta,
                 int32_t *d_dist, gdn_stats *stats);
/* Reusable SSSP state (graph + weights resident).  dense != 0 also builds the propagation-blocked
 * layout of the out-CSR with the weights in tile order: while the frontier is heavy the solver runs
 * Bellman-Ford sweeps over all edges (LDS min-reduction per destination bin) instead of worklist
 * passes, then finishes with a worklist.  Distances are the exact shortest distances either way. */
/* Round 5: when ALL weights are equal (>= 1; the reference main's own input, src/sssp/main.cc:26 fills 1) and the graph has
 * 2^22 edges or more, a dense plan solves through the direction-optimising BFS plan (gdn_bfs_plan_*) on the transpose it builds
 * once, and multiplies the depths by the weight: the same distances, no relaxation repeated (stats.reserved = 1 says so;
 * GDN_SSSP_UNIT_BFS=0 keeps the sweeps, =1 takes the route at any size). */
typedef struct gdn_sssp_plan gdn_sssp_plan;
int gdn_sssp_plan_create(const gdn_graph *csr, const int32_t *d_weight, int32_t dense, gdn_sssp_plan **plan);
int gdn_sssp_plan_free(gdn_sssp_plan *plan);
int gdn_sssp_run(gdn_sssp_plan *plan, int32_t source, int32_t delta, int32_t *d_dist, gdn_stats *stats);
/* in_csr == NULL (a directed graph, out-edges only): Afforest without the giant-component skip (one closing pass over every
 * out-edge).  Option GDN_CC_REVERSE=build: the reverse graph is built inside the call (gdn_graph_transpose, charged to prep_ms)
 * and the solve is the one with it (stats.reserved = 3); GDN_CC_SV=1 | fused: the reference's Shiloach-Vishkin rounds. */
int gdn_cc_dev(const gdn_graph *csr, const gdn_graph *in_csr /*nullable*/, int32_t *d_comp,
               gdn_stats *stats);
int gdn_tc_dev(const gdn_graph *csr, int32_t oriented, uint64_t *total, gdn_stats *stats);
/* Reusable triangle-count state: what TCSolver's caller does ONCE while loading (`Graph g(prefix, USE_DAG)`,
 * src/tc/main.cc:12 -> src/common/graph.cc:67-113) kept apart from the count the reference's Timer brackets
 * (src/tc/gpu_base.cu:52-58).  gdn_tc_plan_create prepares the formulation the count will run (stats of a count:
 * `reserved` 3 = the FORWARD count -- vertices relabelled by degree rank, the DAG = edges to higher ranks, its transpose
 * and, per DAG edge u -> v, where the walk of N+(u) starts (behind v: a member of N+(v) outranks v), which halves the
 * look-ups; taken from 2^22 DAG edges on -- 0 / 1 = the hash-set count on the reference's orientation, u- / v-centric;
 * 2 = the wave-per-edge binary-search intersect; GDN_TC_FORM = f | a | u | v | bs forces one.  Bits 8.. of `reserved`, forward
 * count only: the ranks of its CORE -- from 2^21 vertices on the look-ups whose middle vertex is among the top 16384 ranks
 * are counted on a bit matrix of those ranks, beside the hash-set kernel; GDN_TC_CORE = 0 | 4096 | 8192 | 12288 | 16384).
 * `csr` is a symmetric graph
 * (oriented == 0) or ANY acyclic orientation of one (oriented != 0); it is not referenced after the call.
 * gdn_tc_dev == create + count + free, the preparation in stats.prep_ms. */
typedef struct gdn_tc_plan gdn_tc_plan;
int gdn_tc_plan_create(const gdn_graph *csr, int32_t oriented, gdn_tc_plan **plan);
int gdn_tc_plan_count(gdn_tc_plan *plan, uint64_t *total, gdn_stats *stats);
int gdn_tc_plan_free(gdn_tc_plan *plan);
/* List elements the forward count of this plan walks against its hash sets: the look-ups around the middle vertices BELOW the
 * core's ranks (all of them without a core: (SUM d+(u)^2 - nnz) / 2); 0 for the other formulations.  4 bytes each are what
 * tc_count_kernel requests from the lists -- bench.py's kernel_list_read_frac.  Blocking. */
int gdn_tc_plan_walked_elements(const gdn_tc_plan *plan, uint64_t *elements);
/* The DAG orientation alone (src/common/graph.cc:67-113, what `Graph g(prefix, USE_DAG)` hands to TCSolver), and the count
 * over the source rows [row_lo, row_hi) of an ORIENTED graph: the shard of a multi-GPU count -- every rank holds the DAG,
 * the ranges partition its rows (by DAG-edge count), the partial counts add up (gardenia_amd.sharded.ShardedTC).
 * stats.edges_traversed = DAG edges of the range. */
int gdn_graph_orient(const gdn_graph *csr, gdn_graph **dag);
/* algorithmic bytes of one count on an oriented graph, SURVEY 8d's merge-equivalent model (the roofline denominator of
 * TC): 4 * SUM over DAG edges (u,v) of (d+(u) + d+(v)) + 4 nnz [source list] + 4 nnz [column ids] + 8 (m + 1) */
int gdn_tc_model_bytes(const gdn_graph *dag, uint64_t *bytes);
/* the list elements a count on `dag` walks: probes[0] the u-centric form (SUM over edges of d+(v); the reference's loop,
 * src/tc/omp_base.cc:16-22), probes[1] the v-centric one (SUM of d+(u)).  gdn_tc_dev runs the cheaper one
 * on small graphs, the forward count (half of probes[1] - nnz) on large ones; stats.reserved says which ran, see
 * gdn_tc_plan_create).  4 B x probes = what the kernel requests from memory. */
int gdn_tc_probe_counts(const gdn_graph *dag, uint64_t *probes);
int gdn_tc_rows_dev(const gdn_graph *dag, int32_t row_lo, int32_t row_hi, uint64_t *total, gdn_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* GARDENIA_HIP_H_ */
