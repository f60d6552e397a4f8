/*
 * oa_icp.h -- C-ABI of liboa_icp.so: the MI355X (gfx950) ICP hot path.
 *
 * This is the drop-in boundary for the one hot path of patmo141/object_alignment
 * (SURVEY.md section 8b): what a maintainer's ctypes binding would call instead of
 *
 *   functions/general.py:257   make_pairs(align_obj, base_obj, base_bvh, vlist, thresh, sample, calc_stats)
 *   functions/general.py:105   affine_matrix_from_points(v0, v1, shear=False, scale, usesvd=True)
 *   operators/icp_align.py:96  the `while n < iters and not converged` loop of
 *                              OBJECT_OT_icp_align.execute
 *
 * Conventions
 *   - plain C, no torch / C++ types; all matrices are 4x4 ROW-major;
 *     `float` matrices carry Blender's float32 matrix_world values.
 *   - host buffers are caller-owned and only touched during the call; device
 *     buffers are library-owned and released by oa_destroy.
 *   - every function returns OA_OK (0) or a negative error code; the text of the
 *     last error on the calling thread is oa_last_error().  No C++ exception
 *     crosses this boundary.
 *   - one context = one host thread at a time (not internally locked).
 *     Multi-GPU, one process: oa_create_multi(devices, n) -- the context shards the
 *     source over the listed GPUs, replicates the target, and oa_run / oa_iterate join
 *     the devices' 24 sums inside the library every iteration.
 *     Multi-GPU, one process per GPU (torchrun, MPI): one oa_create context per rank
 *     and the split-phase loop oa_iter_partial -> all-reduce(sum) -> oa_iter_finish.
 *   - there is NO CPU fallback: every compute entry point fails with
 *     OA_E_NO_DEVICE / OA_E_HIP when no gfx950 device is usable.
 *
 * Correspondence rule: nearest target VERTEX (fp32, d2 = fma(dz,dz,fma(dy,dy,dx*dx)),
 * lowest index wins ties) -- see docs/HISTORY.md 5.3 "D2".
 */
#ifndef OA_ICP_H
#define OA_ICP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OA_OK                 0
#define OA_E_BAD_ARG         -1
#define OA_E_HIP             -2
#define OA_E_TOO_FEW_PAIRS   -3   /* K < 3: the reference raises ValueError (functions/general.py:150-157) */
#define OA_E_SINGULAR        -4   /* matrix_world not invertible (functions/general.py:265-266) */
#define OA_E_NO_DEVICE       -5
#define OA_E_STATE           -6   /* source / target / matrices not set */
#define OA_E_BAD_THRESH      -7   /* thresh <= 0: the reference returns None (functions/general.py:277) */
#define OA_E_CAPACITY        -8
#define OA_E_RCCL            -9   /* multi-device exchange failed: librccl missing / ncclCommInitAll / ncclAllReduce error,
                                     or a device's sums did not arrive within OA_EXCHANGE_TIMEOUT_S (default 30 s) -- mailbox:
                                     the waiting kernels give up by themselves; RCCL: the host's watchdog sees no device
                                     finish an iteration for that long and aborts the communicators (ncclCommAbort) */

#define OA_NSUMS 24               /* doubles exchanged per iteration (see oa_iter_partial) */

typedef struct oa_ctx oa_ctx;

/* Loop parameters: lib/preferences.py:31-72 as read at operators/icp_align.py:82-89 */
typedef struct oa_settings {
    int32_t iters;            /* icp_iterations (50) */
    int32_t use_target;       /* use_target (1): calc d_stats and run the convergence test */
    int32_t with_scale;       /* align_meth == '1' (ROT_LOC_SCALE) */
    int32_t early_exit;       /* 1 = stop when converged (reference behaviour); 0 = always run `iters` (timing runs) */
    double  thresh;           /* min_start (0.5) */
    double  target_d;         /* target_d (0.01) */
} oa_settings;

typedef struct oa_report {
    int32_t iters_done;
    int32_t converged;
    int32_t status;               /* OA_OK or the error that stopped the loop */
    int32_t reserved;
    int64_t last_K;               /* pairs used by the last iteration (all shards) */
    double  last_translation;     /* |new_mat.to_translation()| of the last iteration */
    double  mean_dist, std_dist;  /* d_stats of the last iteration (NaN when use_target = 0) */
    double  mean_rot_angle;       /* mean |rotation angle| over the last <=5 iterations (radians) */
    double  nn_ms_total;          /* hipEvent time summed over the correspondence kernels */
    double  loop_ms;              /* hipEvent time of the whole device-resident loop */
} oa_report;

/* ---- lifetime ------------------------------------------------------------------ */
int         oa_device_count(void);
int         oa_create(oa_ctx **out, int device);
/* One context over n_dev GPUs of this process (1 <= n_dev <= 64; SURVEY.md 8b's oa_create(&ctx, devices, n_dev)).
 * Every upload goes to all of them (target replicated; source: the selection is Morton-sorted ONCE, on the first
 * device, and device i receives range i of n_dev of it, see oa_set_source; the devices build their search structures
 * concurrently, one persistent host thread per GPU), oa_run hands every GPU's host thread the whole loop of its
 * device -- one stream per device, no host round trip per iteration, enqueue cost independent of n_dev -- and every
 * iteration the devices exchange their OA_NSUMS partial sums and perform the identical solve
 * (operators/icp_align.py:96-151 is still ONE call).  A device may be listed more than once (its shards then share
 * one stream and one host thread); that is how the path is tested on a single GPU.  oa_make_pairs and oa_nn_search
 * work on such a context too: every device answers for its shard and the library merges the answers back into the
 * caller's (vlist) order.  Device pointers (on_device != 0) may live on any GPU of the process: the library stages
 * them to the device that needs them.  Not available on it: oa_set_stream and the split-phase calls. */
int         oa_create_multi(oa_ctx **out, const int *devices, int n_dev);
int         oa_num_devices(oa_ctx *ctx);
/* How the devices of an oa_create_multi context join their sums (env OA_EXCHANGE=rccl / mailbox; default AUTO):
 *   OA_EXCHANGE_AUTO     RCCL when the listed devices are distinct and librccl can be brought up, else the mailbox;
 *                        decided before the first loop (OA_STAT_EXCHANGE reports the outcome)
 *   OA_EXCHANGE_RCCL     ncclAllReduce(OA_NSUMS doubles, sum) over xGMI on every device's stream, single-process
 *                        communicator (ncclCommInitAll); librccl.so.1 is loaded on first use; needs distinct devices.
 *                        New communicators first pass one all-reduce of a known vector (bounded wait) or are given up.
 *                        Every host thread enqueues the same number of collectives whatever its device reports when
 *                        (OA_STAT_ENQUEUED_MIN == _MAX); no wait for the devices is unbounded: when no device finishes an
 *                        iteration for OA_EXCHANGE_TIMEOUT_S, or RCCL reports an asynchronous error, the communicators
 *                        are aborted and the call returns OA_E_RCCL.  After that AUTO stays on the mailbox;
 *                        oa_set_exchange(OA_EXCHANGE_RCCL) builds new communicators.
 *   OA_EXCHANGE_MAILBOX  all-gather through mailboxes: every device's post is written into every device's inbox --
 *                        fine-grained device memory, peer-mapped, i.e. 200-byte remote writes over xGMI -- and the solve
 *                        kernel of each device waits for the world's posts in its own inbox and adds them in rank order
 *                        (bitwise identical everywhere); nothing is added to the streams.  Without peer access between
 *                        the devices the inboxes collapse into one box in pinned host memory (PCIe; env OA_MAILBOX=host
 *                        forces that, OA_MAILBOX=device refuses the fallback).  A device whose post does not arrive
 *                        within OA_EXCHANGE_TIMEOUT_S (default 30 s) ends the loop with OA_E_RCCL on every device.
 * Returns OA_E_RCCL when RCCL is asked for and cannot be brought up. */
#define OA_EXCHANGE_AUTO   (-1)
#define OA_EXCHANGE_MAILBOX 0
#define OA_EXCHANGE_RCCL    1
int         oa_set_exchange(oa_ctx *ctx, int mode);
void        oa_destroy(oa_ctx *ctx);
/* Released device blocks are kept in a process-wide cache for the next upload of the same size (hipFree costs
 * ~135 us a call, hipMalloc of 100 MB more): at most 256 MiB or what the library has had allocated at once, whichever is
 * larger (so that a re-upload of the same geometry allocates nothing; env OA_DEV_CACHE_MB sets a fixed cap instead,
 * OA_DEV_CACHE=0 switches the cache off), and nothing at all once the last context has been destroyed.  This gives
 * the cached blocks back immediately. */
void        oa_release_cached_memory(void);
const char *oa_last_error(void);
const char *oa_version(void);
/* stream: the hipStream_t every later call enqueues on, used exactly as given -- NULL is HIP's legacy default
 * stream (torch.cuda.current_stream() unless the caller changed it); OA_STREAM_OWN = the context's private
 * non-blocking stream (the default after oa_create). */
#define OA_STREAM_OWN ((void *)(intptr_t)-1)
int         oa_set_stream(oa_ctx *ctx, void *stream);

/* Correspondence-search strategy.  Every mode returns the same (index, d2) per source point, bit for bit:
 *   OA_SEARCH_BRUTE  k_nn_search_sorted: LDS-tiled brute force over all target vertices (the north-star kernel); its images of
 *                    the target, in the order of the longest axis, are built with the upload when this mode is set, else by
 *                    the call that sets it (a target uploaded for the other modes does not pay for them)
 *   OA_SEARCH_GRID   k_nn_search_grid: exact search through a uniform grid; points it cannot settle within a few
 *                    rings (far from the target: partial overlaps, holes) are finished by the tree search
 *   OA_SEARCH_BVH    k_bvh_search: every query through the 64-ary bounding-box tree, one wavefront per query
 *   OA_SEARCH_AUTO   the tree for small shards (up to ~1.2e4 .. 4e4 points, by target size and kind), the grid with
 *                    the tree behind it for larger ones; shards of up to ~1.4e4 .. 3.9e5 points take the tree while the
 *                    pose still moves and the grid once it has settled (decided on the device, per iteration);
 *                    brute force only for targets with non-finite coordinates
 * The same modes apply to surface targets (oa_set_target_mesh) with triangles in place of vertices.
 * Default: OA_SEARCH_AUTO (env OA_NN_GRID = 0 / 1 / 2 overrides at oa_create). */
#define OA_SEARCH_AUTO  (-1)
#define OA_SEARCH_BRUTE 0
#define OA_SEARCH_GRID  1
#define OA_SEARCH_BVH   2
int oa_set_search_mode(oa_ctx *ctx, int mode);

/* ---- one-time uploads (replaces BVHTree.FromObject, operators/icp_align.py:53, and the vlist
 *      walk over align_obj.data.vertices, functions/general.py:280-284) ------------------------ */
/* target (base object) vertices, base-LOCAL, n x 3 float32.  on_device != 0: xyz is a device pointer. */
int oa_set_target(oa_ctx *ctx, const float *xyz, int64_t n, int on_device);
/* SURFACE mode (the BVHTree.find_nearest semantics of functions/general.py:297): the base object's vertices plus
 * its triangles (n_tris x 3 vertex indices, host int32; quads/ngons triangulated by the caller, e.g. Blender's
 * loop_triangles).  The correspondence of a source point is then the closest point on the nearest triangle
 * (Ericson's closest-point-on-triangle in float32, lowest triangle index on ties) instead of the nearest vertex;
 * everything else (threshold, pair mapping, solve, loop) is unchanged.  oa_set_target() returns to vertex mode. */
int oa_set_target_mesh(oa_ctx *ctx, const float *xyz, int64_t n_verts, int on_device,
                       const int32_t *tris, int64_t n_tris);
/* source (align object) vertices, align-LOCAL, n_verts x 3 float32.
 * vlist (host, may be NULL = all vertices) is the operator's vertex list; stride > 1 applies
 * vlist[0::stride] (functions/general.py:274-275).  The selected list is then cut into shard_count shards of
 * ceil(n_selected / shard_count) points and this context keeps shard shard_index.  With shard_count > 1 the
 * shards are equal ranges of the selection in MORTON order (every rank passes the same arrays and derives the
 * same partition), so that a shard is a compact region of space; for non-finite coordinates they are contiguous
 * ranges in the caller's order.  Per-point outputs of a shard (oa_make_pairs, oa_nn_search) list its points in the
 * caller's order.  The sums the loop all-reduces do not depend on the partition.
 * On an oa_create_multi context pass shard 0 of 1: the context deals the shards to its devices itself.
 * Limits: a shard holds at most 65 535 x 256 x R selected points for the brute-force search (R = 4 points per
 * thread for shards >= 65 536 points: ~6.7e7; OA_NN_R=8 doubles it) -- beyond that oa_run fails with OA_E_BAD_ARG and
 * the job needs more shards; the grid / tree searches take any shard that fits an int32 (< 2^31 - 2^20 points).
 * Targets: < 2^31 - 2^16 vertices, < 7.1e8 triangles. */
int oa_set_source(oa_ctx *ctx, const float *xyz, int64_t n_verts, int on_device,
                  const int64_t *vlist, int64_t n_vlist, int32_t stride,
                  int32_t shard_index, int32_t shard_count);
/* EXTENSION (no counterpart in the reference): reject a pair when the angle between the world-space
 * normals of the source vertex and of its correspondence exceeds max_angle_deg.  src_normals: n_verts x 3 (align-
 * local, host); tgt_normals: nt x 3 per target vertex (vertex mode; ignored in surface mode, where the geometric
 * normal of the nearest triangle is used).  Call after oa_set_source / oa_set_target; passing src_normals == NULL
 * or an angle outside (0, 180) switches the test off; a new source or target upload switches it off too. */
int oa_set_normals(oa_ctx *ctx, const float *src_normals, int64_t n_verts, const float *tgt_normals, int64_t nt,
                   double max_angle_deg);
/* matrix_world of the align and base objects (functions/general.py:262-263) */
int oa_set_matrices(oa_ctx *ctx, const float mx_align[16], const float mx_base[16]);
int oa_get_matrix_world(oa_ctx *ctx, float mx_align[16]);
/* Forget the correspondences of earlier searches (every search seeds itself with the previous answer of its slot: any
 * seed gives the same result, a good one gives it sooner).  A new upload does this by itself; oa_set_matrices does
 * not -- a host-driven make_pairs loop sets the matrices before every call and keeps its seeds.  Timing runs call it to
 * start cold. */
int oa_reset_seeds(oa_ctx *ctx);
/* Introspection (sizes of the search structures, for roofline arithmetic and tests). */
#define OA_STAT_GRID_CELLS        1   /* cells of the vertex grid (0 = none built) */
#define OA_STAT_TRI_GRID_CELLS    2   /* cells of the triangle grid */
#define OA_STAT_TRI_GRID_ENTRIES  3   /* (triangle, cell) entries of the triangle grid's cell lists */
#define OA_STAT_N_TRIS            4
#define OA_STAT_SURFACE           5   /* 1 = surface mode (oa_set_target_mesh) */
#define OA_STAT_CACHE_BYTES       6   /* device bytes currently held by the process-wide allocation cache */
#define OA_STAT_BRUTE_KERNEL      7   /* what OA_SEARCH_BRUTE launches for the current shard: 0 = k_nn_search (exact only), 1 =
                                       * k_nn_search_filtered (rounds 1-4; OA_NN_SORT=0), 2 = k_nn_search_mfma (experiment, env OA_NN_MFMA=1),
                                       * 3 = k_nn_search_sorted (default: the filtered search over the target sorted along its longest axis) */
#define OA_STAT_EXCHANGE          8   /* multi-device context: what its loops exchange through (resolves AUTO): 0 = mailbox in pinned
                                       * host memory, 1 = RCCL, 2 = mailboxes in peer-mapped device memory; -1 = not a multi context */
#define OA_STAT_RCCL_RANKS        9   /* ranks of the RCCL communicator the loops use (0 = RCCL not in use) */
#define OA_STAT_ENQUEUE_US       10   /* multi-device context: host time per iteration spent enqueuing ONE device's work in the
                                       * last oa_run (mean over iterations, max over devices), microseconds */
#define OA_STAT_HOST_THREADS     11   /* multi-device context: host threads that drive its devices (1 = the caller alone) */
#define OA_STAT_FAST_ITERATIONS  12   /* iterations of the last / current loop in which the grid search finished its own leftovers and
                                       * accumulated in its epilogue (the adaptive choice of docs/HISTORY.md 4.4; first device) */
#define OA_STAT_HANDOVER_ENTRIES 13   /* what the last grid search handed to the tree: queries ... */
#define OA_STAT_HANDOVER_WAVE_MAX 14  /* ... and the most any ONE wavefront handed over (what the adaptive choice looks at) */
#define OA_STAT_ENQUEUED_MIN      15   /* multi-device context: iterations the host enqueued for its children in the last oa_run, the */
#define OA_STAT_ENQUEUED_MAX      16   /* least and the most over the children.  Equal by construction (docs/HISTORY.md 4.7, "the invariant"):
                                       * in RCCL mode every enqueued iteration holds a collective every rank has to enter */
#define OA_STAT_WATCHDOG_ABORTS   17   /* times this context's RCCL communicators were aborted (watchdog / asynchronous error) */
#define OA_STAT_NN_MS_MIN         18   /* multi-device context: search time of the last oa_run (sum over its iterations, ms) on the */
#define OA_STAT_NN_MS_MAX         19   /* fastest / the slowest device: how evenly the shards load the GPUs */
#define OA_STAT_SAFE_RADII        20   /* 1 = the vertex grid's safe radii are built for the current target (a seed inside its own settles the
                                       * query without a scan or a descent, docs/HISTORY.md 4.4; built once the target has seen 8 loop iterations; first device) */
#define OA_STAT_TRI_RING          21   /* 1 = the triangle neighbour lists are built for the current mesh (a query within its seed triangle's accept
                                       * radius is settled by the seed and the triangles that touch it, docs/HISTORY.md 4.5; built once the mesh has
                                       * seen 4 loop searches, OA_TRI_RING=2: with the grid; first device) */
#define OA_STAT_TRI_RING_ACCEPTS  22   /* diagnostic (one extra launch + a wait): source points of this shard that the neighbour lists would settle at
                                       * the current pose with the current seeds; multi-device context: the sum over the shards */
#define OA_STAT_EXCHANGE_US       23   /* multi-GPU: mean time per iteration of the last loop a device spent between "its sums are ready" and "the
                                       * world's sums are in hand" (GPU-side stamps: the all-reduce, or the gather's wait), slowest device */
#define OA_STAT_RCCL_FALLBACKS    24   /* loops that AUTO began on RCCL and finished through the mailboxes (the watchdog aborted the communicators) */
#define OA_STAT_RCCL_RANKS_LAST   25   /* ranks of the last RCCL communicator that came up and passed its handshake -- still reported after a
                                       * fallback, when OA_STAT_RCCL_RANKS is back to 0 */
#define OA_STAT_SEARCH_CLOCK_MHZ   26   /* shader clock during the last loop's k_nn_search_sorted launch (cycle counter / constant-rate counter of one
                                       * workgroup dispatched mid-launch); 0 when that kernel did not run.  Multi-device context: its first device */
#define OA_STAT_BRUTE_QUEUE_WGS     27   /* workgroups of the last k_nn_search_sorted launch that took their (split, block) items off the work
                                       * queue (long launches: as many as the chip holds); 0 = one workgroup per item, in launch order */
#define OA_STAT_ENQUEUED_CHILD  1000   /* + i: the same count for child i alone */
int oa_get_stat(oa_ctx *ctx, int what, double *value);
/* why the exchange is what it is (AUTO's reason for not taking RCCL, librccl's error, "RCCL was aborted: ..."), or "" */
const char *oa_exchange_note(oa_ctx *ctx);
int64_t oa_num_selected(oa_ctx *ctx);     /* selected source points held by this context (its shard) */

/* ---- contract 1: make_pairs (functions/general.py:257-329) ------------------------------------ */
/* A, B: caller-allocated 3 x cap row-major doubles (row = axis); pairs come out in vlist order.
 * dstats = [mean, population std] of the world-space pair distances when calc_stats. */
int oa_make_pairs(oa_ctx *ctx, double thresh, int calc_stats,
                  double *A, double *B, int64_t cap, int64_t *K, double dstats[2]);
/* the correspondence search alone: nearest target vertex index (surface mode: nearest triangle index) and fp32
 * squared distance (base-local) for every selected source point of this shard.  idx/d2 may be NULL (timing). */
int oa_nn_search(oa_ctx *ctx, int64_t *idx, float *d2, double *kernel_ms);

/* ---- contract 2: affine_matrix_from_points(v0=A, v1=B, shear=False, scale, usesvd=True)
 *      (functions/general.py:105-217); alias calc_target_matrix in the Python host ------------ */
/* A, B: 3 x K row-major doubles with leading dimension ld (host).  with_scale: bit 0 = uniform scale (scale=True); bit 1 = the
 * rotation through Horn's quaternion (usesvd=False, :191-206) instead of the SVD of the covariance (:179-190). */
int oa_kabsch(oa_ctx *ctx, const double *A, const double *B, int64_t K, int64_t ld,
              int with_scale, double M[16]);
/* The reference's full signature (functions/general.py:105): v0, v1 are ndims x K row-major doubles with leading
 * dimension ld (host), 2 <= ndims <= 64 (the reference takes any; 9 and more run through a device workspace); shear != 0: the affine (Hartley & Zisserman) branch (:168-178, the signature's
 * default), else rigid / similarity through the SVD of the covariance (:179-190, :208-212).  M: (ndims+1)^2 doubles,
 * row-major.  K < ndims or ndims < 2: OA_E_TOO_FEW_PAIRS, the reference's ValueError (:150-157). */
int oa_affine_from_points(oa_ctx *ctx, const double *v0, const double *v1, int ndims, int64_t K, int64_t ld,
                          int shear, int with_scale, double *M);
/* same solve from the OA_NSUMS accumulated sums (host array; layout: S_A .. S_DD in csrc/oa_kernels.hpp).  The sums are taken
 * relative to `pivot` (a' = a - pivot, b' = b - pivot); pivot == NULL means the origin. */
int oa_kabsch_from_sums(oa_ctx *ctx, const double sums[OA_NSUMS], const double pivot[3], int with_scale,
                        double M[16]);
/* the pivot this context subtracts before accumulating (the first selected source vertex) */
int oa_get_pivot(oa_ctx *ctx, double pivot[3]);

/* ---- fused fast path: the operator loop (operators/icp_align.py:91-151) ----------------------- */
/* one iteration, synchronous (the modal operator's per-tick step, icp_align_feedback.py:250-288):
 * M_step = this iteration's 4x4 (float64); stats = [K, mean_dist, std_dist, |translation|, rot_angle, converged].
 * A sequence of calls with the same thresh / target_d / use_target / with_scale is one loop (n counts up, the 5-slot
 * convergence ring fills).  Different settings, or any call in between that re-stages the device state
 * (oa_set_matrices, oa_make_pairs, oa_nn_search, oa_run, a new upload), start a new sequence -- n = 0, fresh ring --
 * from the current matrix_world.  History: the last 64 iterations of the sequence. */
int oa_iterate(oa_ctx *ctx, const oa_settings *st, double M_step[16], double stats[6]);
/* the whole loop, device resident (no host round trip per iteration) */
int oa_run(oa_ctx *ctx, const oa_settings *st, oa_report *rep);
/* per-iteration history of the last oa_run (its first max_n iterations) / oa_iterate sequence (its last max_n,
 * oldest first) -- each array may be NULL:
 * step_M n x 16 doubles, step_new n x 16 floats (new_mat, operators/icp_align.py:116-119),
 * step_K n, step_stats n x 2, step_trans n.  Returns the number of iterations recorded. */
int oa_get_history(oa_ctx *ctx, int32_t max_n, double *step_M, float *step_new,
                   int64_t *step_K, double *step_stats, double *step_trans);

/* search time (ms) of every iteration of the last oa_run / oa_run_end -- what oa_report::nn_ms_total sums; returns how many
 * were written (<= max_n).  Multi-device context: its first device.  (bench.py: per-launch min / median / max) */
int oa_get_search_ms(oa_ctx *ctx, int32_t max_n, double *ms);
/* Diagnostic for bench.py's roofline (no counterpart in the reference): what the vector ALUs issue right now -- ~target_ms of
 * independent v_add_f32 / v_min3_f32 chains on every SIMD, 8 waves each.  out[0] = T lane-ops/s of v_add_f32 (two register
 * sources: the issue rate itself), out[1] = shader clock in MHz during that burn, out[2] = duration in ms, out[3] = T lane-ops/s
 * of v_min3_f32 (the half-rate class: v_min_f32, v_min3_f32, v_cmp_*_f32).  The brute-force search is bound by exactly these
 * rates, weighted by its instruction mix. */
int oa_measure_valu_ceiling(oa_ctx *ctx, double target_ms, double out[4]);

/* ---- split-phase loop for one-process-per-GPU sharding ---------------------------------------- */
/* oa_run_begin resets the loop state (ring buffer, counters) on the device. */
int oa_run_begin(oa_ctx *ctx, const oa_settings *st);
/* enqueue correspondence search + pair accumulation for this shard; the OA_NSUMS partial sums
 * land in d_sums (DEVICE pointer, e.g. a torch tensor) ready for an all-reduce(sum). */
int oa_iter_partial(oa_ctx *ctx, double *d_sums);
/* enqueue solve + matrix_world update + convergence ring from the (all-reduced) sums. */
int oa_iter_finish(oa_ctx *ctx, const double *d_sums);
/* synchronise and fetch the report. */
int oa_run_end(oa_ctx *ctx, oa_report *rep);

#ifdef __cplusplus
}
#endif
#endif /* OA_ICP_H */
