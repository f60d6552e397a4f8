// oa_icp.hip -- C-ABI (include/oa_icp.h) over the gfx950 kernels in oa_kernels.hpp.
//
// Host side of the drop-in boundary: context lifetime, uploads/packing, launch geometry, the device-resident
// iterate loop and its split-phase form for one-process-per-GPU sharding.  No CPU fallback lives here: every
// compute entry point needs a usable HIP device and fails loudly otherwise.
#include <cstring>
#include <dlfcn.h>
// librccl is loaded with dlopen on first use (liboa_icp.so has no link dependency on it); a ROCm install without the
// RCCL development headers still builds this file from the handful of declarations the exchange needs
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include "oa_all.hpp"                // every kernel header + oa_families.hpp: the heavy templates are `extern` here, compiled in oa_fam_*.hip
#include "oa_sort.hpp"
#include "../../include/oa_icp.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#define OA_EXPORT extern "C" __attribute__((visibility("default")))
// a multi-device parent (oa_create_multi) routes the call to every child / to its first child
#define OA_ROUTE_ALL(c, call)                                                                  \
    if ((c) && !(c)->subs.empty()) {                                                           \
        for (oa_ctx *sub : (c)->subs) { const int rc_ = (call); if (rc_) return rc_; }         \
        return OA_OK;                                                                          \
    }
// uploads and index builds: the children work at the same time, one host thread each, when they sit on different GPUs
#define OA_ROUTE_ALL_PAR(c, call)                                                              \
    if ((c) && !(c)->subs.empty()) return route_all_parallel((c), [&](oa_ctx *sub) -> int { return (call); });
#define OA_ROUTE_FIRST(c, call)                                                                \
    if ((c) && !(c)->subs.empty()) { oa_ctx *sub = (c)->subs[0]; return (call); }
#define OA_NOT_MULTI(c, what)                                                                  \
    if ((c) && !(c)->subs.empty())                                                             \
        return fail(OA_E_STATE, what ": not available on a multi-device context (use a single-device one)");

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(OA_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Device allocations go through a small process-wide cache keyed by (device, size): hipFree costs ~135 us a call on
// this stack (it synchronises the device), and an upload frees and reallocates ~25 buffers of unchanged size -- more
// than half of oa_set_target + oa_set_source at 1M points.  A released block is kept for the next request of the same
// size; blocks that sit unused for CACHE_MAX_AGE releases, or push the cache past its cap, are really freed.
// The library is a guest in somebody else's process (Blender): the cap is 256 MiB or the library's own peak of live bytes,
// whichever is larger (OA_DEV_CACHE_MB sets a fixed cap, OA_DEV_CACHE=0 switches the cache off), and when the last context is
// destroyed everything is freed.
// Releases are STREAM ORDERED, not device-synchronising: a block goes back tagged with the stream of the context that
// used it (tl_stream, set by use_device).  The next owner on the same stream needs no wait at all (stream order); an
// owner on another stream first waits for the releasing stream (rare: two contexts trading blocks).  Only a real hipFree
// synchronises (the runtime's own rule).  No events: recording one per release cost more than the device synchronisation
// it replaced on an idle device (a whole alignment call of 2562 points: 0.46 -> 0.64 ms).
thread_local hipStream_t tl_stream = nullptr;
thread_local bool tl_stream_known = false;

struct DevCache {
    struct Block { void *p; size_t bytes; int device; unsigned long long stamp; hipStream_t stream; bool settled; };
    std::mutex mu;
    std::vector<Block> free_blocks;
    std::unordered_map<void *, std::pair<size_t, int>> live;    // pointer -> (bytes, device)
    size_t cached_bytes = 0;
    unsigned long long clock = 0;
    int contexts = 0;                                           // live oa_ctx objects (single-device ones)
    static constexpr unsigned long long CACHE_MAX_AGE = 512;

    const bool enabled = !(getenv("OA_DEV_CACHE") && atoi(getenv("OA_DEV_CACHE")) == 0);   // OA_DEV_CACHE=0: plain hipMalloc / hipFree
    const bool cap_given = getenv("OA_DEV_CACHE_MB") && *getenv("OA_DEV_CACHE_MB");
    const size_t max_bytes = (size_t)(cap_given ? std::max(0.0, atof(getenv("OA_DEV_CACHE_MB"))) : 256.0) << 20;
    // The cap follows what the process has had allocated AT ONCE (round 6): a 1.96M-triangle mesh is ~0.5 GB of images, lists and
    // trees, and with a fixed 256 MiB every re-upload of it really freed and really allocated most of that -- 2.9 of the 4.8 ms of
    // oa_set_target_mesh.  Idle, the cache never holds more than the library already needed live; an explicit OA_DEV_CACHE_MB wins.
    size_t live_bytes = 0, peak_live = 0;
    size_t cap() const { return cap_given ? max_bytes : std::max(max_bytes, peak_live); }

    hipError_t alloc(void **out, size_t bytes)
    {
        if (bytes == 0) bytes = 1;
        if (!enabled) return hipMalloc(out, bytes);
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::unique_lock<std::mutex> lk(mu);
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].bytes == bytes && free_blocks[i].device == dev) {
                    const Block b = free_blocks[i];
                    *out = b.p;
                    cached_bytes -= bytes;
                    free_blocks[i] = free_blocks.back();
                    free_blocks.pop_back();
                    live[*out] = { bytes, dev };
                    live_bytes += bytes; peak_live = std::max(peak_live, live_bytes);
                    lk.unlock();
                    // same stream: stream order is enough.  Another stream (or none known): whatever the releasing
                    // stream still has in flight must finish first
                    if (!b.settled && !(tl_stream_known && b.stream == tl_stream))
                        if (hipStreamSynchronize(b.stream) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
                    return hipSuccess;
                }
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {                                     // out of memory: drop the cache and retry once
            (void)hipGetLastError();
            trim(0, 0);
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); live[*out] = { bytes, dev }; live_bytes += bytes; peak_live = std::max(peak_live, live_bytes); }
        return e;
    }

    // settled: the caller has already synchronised the device (oa_destroy) -- nothing can still be using the block
    void release(void *p, bool settled = false)
    {
        if (!p) return;
        if (!enabled) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; }
        Block b{ p, it->second.first, it->second.second, ++clock, tl_stream, settled };
        live.erase(it);
        live_bytes -= std::min(live_bytes, b.bytes);
        if (b.bytes > cap()) { (void)hipFree(p); return; }         // would not fit under the cap anyway (hipFree synchronises)
        if (!settled && !tl_stream_known) { (void)hipDeviceSynchronize(); b.settled = true; }   // no stream to order against
        free_blocks.push_back(b);
        cached_bytes += b.bytes;
        trim_locked(cap(), CACHE_MAX_AGE);
    }

    // a stream is about to be destroyed (its device has been synchronised): its blocks have nothing in flight any more
    void stream_gone(hipStream_t s)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (Block &b : free_blocks) if (b.stream == s) { b.settled = true; b.stream = nullptr; }
    }

    void trim(size_t cap, unsigned long long max_age) { std::lock_guard<std::mutex> lk(mu); trim_locked(cap, max_age); }

    void trim_locked(size_t cap, unsigned long long max_age)
    {
        // oldest first, so that what stays is what was released last
        std::sort(free_blocks.begin(), free_blocks.end(), [](const Block &a, const Block &b) { return a.stamp < b.stamp; });
        size_t i = 0;
        while (i < free_blocks.size()) {
            const bool old = clock - free_blocks[i].stamp > max_age;
            if (old || cached_bytes > cap) {
                (void)hipFree(free_blocks[i].p);                   // synchronises the device
                cached_bytes -= free_blocks[i].bytes;
                free_blocks.erase(free_blocks.begin() + (long)i);
            } else ++i;
        }
    }

    void context_created() { std::lock_guard<std::mutex> lk(mu); ++contexts; }
    void context_destroyed()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (--contexts > 0) return;
        contexts = 0;
        trim_locked(0, 0);                                         // last context gone: give everything back
        peak_live = live_bytes;
    }
};

DevCache &dev_cache()
{
    static DevCache *cache = new DevCache();                      // never destroyed: the HIP runtime may already be gone at exit
    return *cache;
}

template <typename T> hipError_t dev_malloc(T **p, size_t bytes) { return dev_cache().alloc((void **)p, bytes); }

template <typename T> void dev_free(T *&p, bool synced = false)
{
    if (p) { dev_cache().release((void *)p, synced); p = nullptr; }
}

// temporary device buffer, released on every exit path
template <typename T> struct DevTmp {
    T *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { reset(); }
    hipError_t alloc(size_t n) { reset(); return dev_malloc(&p, sizeof(T) * (n ? n : 1)); }
    void reset() { dev_free(p); }
    operator T *() const { return p; }
};

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

double env_double(const char *name, double dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atof(v) : dflt;
}

// Host threads of a multi-device context.  One PERSISTENT worker per listed GPU (children that share a GPU share a
// worker and stay sequential): uploads and index builds (20+ ms of copies, sorts and builds per device at 1M) run on all
// GPUs at the same time, and oa_run hands every worker the whole loop of its device -- each GPU is fed by its own
// thread, so the enqueue cost per iteration does not grow with the number of GPUs.  Round 2 spawned fresh std::threads
// per upload call and enqueued every device's iteration from the calling thread (~50 runtime calls per iteration at 8
// GPUs).  OA_MULTI_THREADS: -1 (default) as above, 0 = everything on the calling thread, 1 = one worker per CHILD even
// when children share a GPU (test hook: the threaded paths on a one-GPU box).
struct WorkerPool {
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = true, quit = false;
        int rc = 0;
        std::string err;
    };
    std::vector<std::unique_ptr<Worker>> workers;

    static void loop(Worker *w)
    {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [w] { return w->has_job || w->quit; });
                if (w->quit) return;
                job = std::move(w->job);
                w->has_job = false;
            }
            g_err.clear();
            const int rc = job();
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->rc = rc;
                w->err = rc ? g_err : std::string();
                w->done = true;
            }
            w->cv.notify_all();
        }
    }

    void start(size_t n)
    {
        while (workers.size() < n) {
            workers.emplace_back(new Worker());
            Worker *w = workers.back().get();
            w->th = std::thread(loop, w);
        }
    }

    // f(k) for k = 0 .. n-1: k = 0 on the calling thread, the others on workers 0 .. n-2.  The first failure (lowest k)
    // wins: its code is returned and its message becomes this thread's oa_last_error().
    template <typename F> int run(size_t n, F f)
    {
        if (n == 0) return OA_OK;
        start(n - 1);
        for (size_t k = 1; k < n; ++k) {
            Worker *w = workers[k - 1].get();
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->job = [f, k]() -> int { return f(k); };
                w->has_job = true;
                w->done = false;
            }
            w->cv.notify_all();
        }
        int rc0 = f(0);
        std::string err0 = rc0 ? g_err : std::string();
        for (size_t k = 1; k < n; ++k) {
            Worker *w = workers[k - 1].get();
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return w->done; });
            if (!rc0 && w->rc) { rc0 = w->rc; err0 = w->err; }
        }
        if (rc0) g_err = err0;
        return rc0;
    }

    ~WorkerPool()
    {
        for (auto &w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
    }
};

template <typename F> int route_all_parallel(oa_ctx *c, F f);

}  // namespace

struct oa_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // target
    int nt = 0, n_groups_pad = 0;
    float *d_tgt_xyz = nullptr;
    float4 *d_tg = nullptr;
    float4 *d_tf = nullptr;          // filter image, level 1: 3 float4 per group (k_nn_search_filtered)
    float4 *d_tf3 = nullptr;         // filter image, level 2: 2 float4 per group
    // the same images in the order of the coordinate along the cloud's longest axis (k_nn_search_sorted, round 5)
    float4 *d_tfs = nullptr, *d_tf3s = nullptr, *d_tgs = nullptr;   // 3 / 2 / 3 float4 per group of 4 sorted positions
    int4 *d_tidx = nullptr;          // original index of every sorted position
    int sax[3] = { 0, 1, 2 };        // its axes: longest (sorted along), second, thinnest
    bool nn_sort = true;             // OA_NN_SORT=0 (A/B): the brute-force search stays k_nn_search_filtered (rounds 1-4)
    int fax[3] = { 0, 1, 2 };
    double bb_lo[3] = { 0, 0, 0 }, bb_hi[3] = { 0, 0, 0 };
    // uniform grid (k_nn_search_grid)
    bool grid_ok = false;
    int grid_mode = -1;              // OA_NN_GRID: -1 auto, 0 never, 1 whenever possible
    oa::GridParams gp;
    int n_cells = 0;
    int *d_cell_start = nullptr;
    float4 *d_sorted = nullptr;
    int *d_todo_list = nullptr, *d_todo_count = nullptr;   // d_todo_count: {entries, most per wave} of the hand-over list; from TODO_WORDS on: the two counter sets of d_ulist
    int *d_ulist = nullptr;          // surface mode: the queries the front search (k_tri_settle) did not settle -> k_tri_search_grid (oa_tri.hpp: ulist_*)
    int u_slot = 0;                  // which of the two counter sets the next front launch fills (it zeroes the other one)
    // surface mode (closest point on triangle, oa_tri.hpp)
    bool surface = false, tri_grid_ok = false;
    int n_tris = 0;
    float4 *d_tri9 = nullptr;
    oa::GridParams tgp;
    int *d_tcell_start = nullptr;
    float4 *d_tcell_rec = nullptr;   // two float4 per cell-list entry: {disc centre, radius} {unit normal, triangle index}
    long long n_tri_entries = 0;     // entries of the triangle grid's cell lists
    double tri_mean_diag = 0.0;      // mean bounding-box diagonal of the triangles (the cell sizes are multiples of it)
    // the settled-pose search (oa_tri_fine.hpp): a sparse fine grid with inflated lists of whole triangles
    int tri_fine = 0;                // OA_TRI_FINE (EXPERIMENT, off): 0 = never, 1 = built with the mesh when it has >= tri_fine_min_tris triangles, 2 = always
    int tri_fine_min_tris = 200000;  // OA_TRI_FINE_MIN_TRIS (smaller meshes: short calls would pay ~80 us of build for a search they never reach)
    bool tri_fine_ok = false;
    oa::FineParams tfp;
    uint4 *d_tfine_table = nullptr;
    float4 *d_tfine_rec = nullptr;
    long long n_fine_entries = 0;
    // seed + neighbours settle a query (oa_tri_ring.hpp): per triangle its neighbours' indices; the accept radius lives in d_tri9
    int *d_tri_ring = nullptr;
    bool tri_ring_ok = false;        // built for the current mesh
    int tri_ring = 0;                // OA_TRI_RING (EXPERIMENT, off: exact, measured, not faster -- docs/HISTORY.md 4.5): 0 = never; 1 = built once a mesh has seen TRI_RING_LAZY_ITERS searches of a loop; 2 = built with the grid
    double tri_ring_cap = 0.25;      // OA_TRI_RING_CAP: clearances are looked for up to this fraction of a cell edge
    bool tri_wave_wgs = true;        // OA_TRI_WAVE_WGS=0 (A/B): the plain surface search in workgroups of 256 queries also for long launches, as until round 6
    bool tri_split_lanes = true;     // OA_TRI_SPLIT_LANES=0 (A/B): the list is always searched with the shard's own lanes per query
    bool tri_split = true;           // OA_TRI_SPLIT=0 (A/B): the seed + neighbours test stays in the grid search's prologue (no k_tri_accept launch)
    long long tri_iters = 0;         // loop searches enqueued since the mesh was set
    // bounding-box trees (oa_bvh.hpp): over the vertices, and over the triangles in surface mode
    bool bvh_ok = false, tbvh_ok = false;
    oa::BvhParams bvh, tbvh;
    float4 *d_bvh_box = nullptr, *d_bvh_prims = nullptr, *d_tbvh_box = nullptr, *d_tbvh_prims = nullptr;
    bool filter_ok = false;
    float tc[3] = { 0, 0, 0 };
    double qmax = 0.0;
    int *d_members = nullptr;        // shard of a spatially sharded selection: position in the whole selection of every caller-order slot (ascending); nullptr = shard_begin + slot
    long long shard_begin = 0;
    int *d_pos = nullptr;            // oa_make_pairs on a shard: whole-selection position of every pair
    std::vector<int> h_members;      // host copy of d_members (children of a multi-device context)
    int mfma_wps = 4;                // OA_MFMA_WPS: waves per SIMD k_nn_search_mfma is built for (4: 128 registers, 3: 168, 2: 256)
    int nn_mfma = 0;                 // OA_NN_MFMA=1 (experiment): first filter level of the brute-force search on the matrix cores (oa_mfma.hpp)
    oa::half8 *d_tfm = nullptr;      // its image of the target
    double mfma_sigma = 1.0;
    // source (this shard)
    int ns = 0, ns_pad = 0, R = 4;
    float4 *d_src4 = nullptr;
    unsigned long long *d_keys = nullptr;
    int grid_lanes = 0;              // OA_GRID_LANES: lanes per query of k_nn_search_grid (0 = by shard size)
    double turn_frac = 0.1;          // OA_TURN_FRAC: the tree keeps its turn while the pose moves by more than this part of a cell
    bool debug = false;              // OA_DEBUG (read at oa_create)
    bool grid_stats = false;         // OA_GRID_STATS: instrumented triangle-grid launches print what the queries did
    bool tri_share = true;           // OA_TRI_SHARE=0 (A/B): every lane of the triangle-grid search walks its own records (rounds 2-3)
    int list_blocks_per_cu = 16;     // OA_LIST_BLOCKS_PER_CU: workgroups (of four waves) per CU of the tree search over the hand-over list
    bool tri_canon = true;           // OA_TRI_CANON=0 (A/B): surface loops accumulate through the grid-stride k_pair_accumulate (rounds 1-3)
    bool tri_acc = true;             // OA_TRI_ACC=0 (A/B): the triangle grid search never accumulates in its epilogue
    int turns_on = 1;                // OA_SEARCH_TURNS: tree while the pose moves, grid afterwards (mid-size shards, AUTO)
    bool seeded = false;             // a search has run since the last set_source / set_target (seeds exist)
    int *d_prev = nullptr;           // nearest index of the previous search (seed), -1 = none
    float4 *d_win = nullptr;         // vertex mode: per slot, the last winner's coordinates + index in .w (-1 = none)
    uint2 *d_wsafe = nullptr;        // vertex grid: per slot {index, safe2 bits} of a winner the grid scan found (k_grid_safe_radius); 0xFF.. = none
    float *d_safe_by_idx = nullptr;  // vertex grid: safe2 per target vertex (original index); allocated with the grid, filled by build_safe_radii
    bool safe_ok = false;            // ... and filled
    int grid_safe = 1;               // OA_GRID_SAFE: 0 = the vertex searches (grid, whole-shard tree) never take a seed on its safe radius (A/B);
                                     // 1 = the radii are built once a target has seen SAFE_LAZY_ITERS accumulating searches (a 5-iteration
                                     // call at 1M vertices would pay 50-80 us to save 10); 2 = built with the grid
    long long target_iters = 0;      // accumulating grid searches enqueued since the target was set
    int *d_sel = nullptr;            // vertex index held by each source slot
    float4 *d_src4o = nullptr;       // the same points in the caller's (vlist) order -- only oa_make_pairs needs it
    int *d_perm = nullptr;           // sorted slot -> caller-order slot (nullptr: not sorted)
    unsigned short *d_worder = nullptr;   // k_sorted_wave_order: per wave of k_nn_search_sorted, its slots in the order of u (OA_NN_WAVE_ORDER=0: off)
    int *d_homes = nullptr, *d_qcnt = nullptr;   // k_nn_search_sorted's work queue (oa_kernels.hpp): per block of source points its own split; the queues' counters
    int nn_persist = 4;              // OA_NN_PERSIST: workgroups per CU that work the queue off (0: one workgroup per item, in launch order -- as until round 6)
    int nn_queue_min = -1;           // OA_NN_QUEUE_MIN_ITEMS: launches of at least this many items go through the queue (-1: four per workgroup)
    int last_queue_wgs = 0;          // OA_STAT_BRUTE_QUEUE_WGS
    bool nn_wave_order = true;
    long long src_n_verts = 0;
    // normal-angle rejection (extension)
    float *d_src_n = nullptr, *d_tgt_n = nullptr;
    double cos_min = -2.0;
    bool normals_on = false;
    double pivot[3] = { 0, 0, 0 };
    // launch geometry for k_nn_search
    int n_splits = 1, acc_blocks = 1;
    int n_splits_seeded = 1;         // k_nn_search_sorted once the winner records hold seeds: more, shorter splits (plan_geometry)
    bool nn_home_pass = true;        // the first search of a loop: k_nn_seed_sorted + a seeded launch (OA_NN_HOME_PASS=0: ONE unseeded launch)
    bool win_seeds = false;          // an accumulation has written winner records since the last upload / oa_reset_seeds
    int tile_groups = oa::FTILE_GROUPS;   // LDS tile of k_nn_search_filtered: 256 groups, 64 for small targets
    int R_env = 0;                      // OA_NN_R override (0 = choose from the shard size)
    bool use_filter = true;
    double d_pivot0 = 0.0;           // initial distance pivot for the next loop / one-shot (see DevState::d_pivot)
    // device state
    oa::DevState h_state;
    oa::DevState *d_state = nullptr;
    bool have_mats = false, loop_active = false;
    oa::StepRecord *d_hist = nullptr;       // device view of h_hist_map: the solve kernel writes the records straight into host memory
    oa::StepRecord *h_hist_map = nullptr;   // pinned, device-mapped history (read after a stream sync, no copy)
    oa::DevState *h_state_pin = nullptr;    // pinned staging of DevState (pageable copies cost ~30 us each way)
    char *h_scratch = nullptr;              // pinned scratch for small read-backs (read_small)
    char *h_result = nullptr;               // pinned, device-mapped: kernels store small results here (result_buffer)
    void *d_result_view = nullptr;
    double last_exchange_us = 0.0;        // mean exchange wait per iteration of the last loop (fill_report; OA_STAT_EXCHANGE_US)
    std::vector<double> h_search_ms;      // search time of every iteration of the last loop (fill_report; oa_get_search_ms)
    std::vector<oa::StepRecord> h_hist;   // host copy of the executed iterations' records, filled by fill_report
    bool h_hist_valid = false;
    int max_records = 0;
    double *d_partials = nullptr, *d_sums = nullptr, *d_solve = nullptr;
    int fused_acc = 1;                      // OA_FUSED_ACC: grid / tree searches of the loop accumulate in their epilogue
    int tree_acc_max = 4096;                // OA_TREE_ACC_MAX: largest shard whose whole-shard tree search also accumulates
    int acc_threads = 0;                    // OA_ACC_THREADS: 256 / 512 threads per workgroup of the accumulating grid search and k_pair_accumulate_canon (0 = by shard size)
    int grid_path = 0;                      // OA_GRID_PATH: 0 = adaptive (see grid_fast_now), 1 = always the fused path, 2 = never
    int iter_enq = 0;                       // iterations enqueued since the loop began
    int fast_iters = 0;                     // ... of which the grid search finished its own leftovers and accumulated (OA_STAT_FAST_ITERATIONS)
    bool fast_prev = false;                 // what the last iteration's grid search did
    int last_todo_wave_max = -1;            // most queries one wave handed over in the last iteration the host heard of (-1: unknown)
    // make_pairs scratch (sized to ns)
    int emit_cap = 0;
    unsigned char *d_valid = nullptr;
    float *d_b = nullptr;
    double *d_dist = nullptr;
    int *d_counts = nullptr;
    long long *d_offsets = nullptr;
    double *d_A = nullptr, *d_B = nullptr;
    // timing
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    hipEvent_t ev_loop0 = nullptr, ev_loop1 = nullptr;
    bool poll_mapped = false;           // the device can write h_poll (DevState::host_halt is set): else the kernels never report their progress there
    int32_t *h_poll = nullptr;          // pinned, device-mapped {halt, n, hand-over entries, most per wave, -, iteration whose exchange the stream has reached (k_reduce_partials), -, -}: the kernels mirror them here for the enqueuing host
    bool time_events = true;            // hipEvent pair around every search (brute force); else GPU-side stamps (see iter_fused)
    double wall_clock_khz = 100000.0;   // rate of wall_clock64()
    oa_settings settings;
    bool iterate_mode = false;          // the active loop was opened by oa_iterate (small history ring)
    // multi-device (oa_create_multi).  The parent only routes: uploads go to every child (target replicated, source
    // sharded child i of n), the loop drives all children and the exchange below joins their sums every iteration.
    std::vector<oa_ctx *> subs;         // parent: one child per entry of the device list
    struct oa_exchange *xch = nullptr;  // parent: owns it; child: the parent's
    oa_ctx *parent = nullptr;
    int rank = 0, world = 1;            // child: its place in the device list
    std::vector<std::vector<int>> groups;   // parent: children per host thread in the LOOP (one group per GPU, see WorkerPool)
    std::vector<std::vector<int>> upload_groups;   // ... and in uploads / index builds (OA_MULTI_THREADS=1: one per child)
    WorkerPool *pool = nullptr;         // parent: the persistent host threads of groups 1 .. n-1 (group 0 = the caller)
    double last_nn_ms = 0.0;            // child: search time of the last oa_run on this device (OA_STAT_NN_MS_MIN / _MAX)
    double enq_ns = 0.0;                // child: host time spent enqueuing its iterations in the last loop
    long long enq_iters = 0;
};

namespace {

// d_todo_count: TODO_WORDS ints for the tree's hand-over list, then two sets of the front search's list counters (ulist_*)
constexpr int TODO_WORDS = 32, UCOUNT_WORDS = oa::ULIST_PARTS * oa::ULIST_STRIDE, TODO_COUNT_INTS = TODO_WORDS + 2 * UCOUNT_WORDS;
[[maybe_unused]] inline int *ucount_set(const oa_ctx *c, int slot) { return c->d_todo_count + TODO_WORDS + slot * UCOUNT_WORDS; }

int use_device(oa_ctx *c)
{
    HIPCHK(hipSetDevice(c->device));
    tl_stream = c->stream; tl_stream_known = true;              // releases of device blocks are ordered on this stream
    return OA_OK;
}

// which GPU a device pointer lives on (-1: not a device pointer the runtime knows)
int pointer_device(const void *ptr)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.type == hipMemoryTypeDevice ? at.device : -1;
}

template <typename F> int route_all_parallel(oa_ctx *c, F f)
{
    const size_t ng = c->upload_groups.size();
    if (ng <= 1 || !c->pool) {
        for (oa_ctx *sub : c->subs) { const int rc = f(sub); if (rc) return rc; }
        return OA_OK;
    }
    return c->pool->run(ng, [c, f](size_t g) -> int {
        for (int i : c->upload_groups[g]) { const int rc = f(c->subs[(size_t)i]); if (rc) return rc; }
        return OA_OK;
    });
}

void plan_geometry(oa_ctx *c)
{
    if (c->ns <= 0 || c->nt <= 0) return;
    const int src_blocks = c->ns_pad / (oa::NN_THREADS * c->R);
    // small targets: quarter-size tiles give 4x more (and 4x shorter) workgroups; the unfiltered kernel keeps 256
    c->tile_groups = (c->filter_ok && c->use_filter && c->nt <= 65536 && !env_int("OA_NN_BIGTILE", 0)) ? 64 : oa::FTILE_GROUPS;
    const int tiles_total = c->n_groups_pad / c->tile_groups;
    // workgroups wanted: ~2e5 pairs each (below that the prologue and the atomicMin merge dominate), at most 64 per CU
    const double pairs = (double)c->ns * (double)c->nt;
    const int want_auto = (int)std::min((double)c->n_cu * 64.0, std::max((double)c->n_cu * 2.0, pairs / 2e5));
    const int want = env_int("OA_NN_TARGET_BLOCKS", want_auto);
    int splits = (want + src_blocks - 1) / src_blocks;
    splits = std::max(1, std::min(splits, tiles_total));
    if (splits > 8) splits = std::min(tiles_total, ((splits + 7) / 8) * 8);   // multiple of 8: one XCD per split residue
    const int forced = env_int("OA_NN_SPLITS", 0);
    if (forced > 0) splits = std::min(forced, tiles_total);
    c->n_splits = splits;                                          // the kernels cut the tiles into exactly this many ranges (split_range)
    // k_nn_search_sorted with seeds: a workgroup's points reach levels 1-3 only in the split that holds their own slab, so short
    // splits (>= 3 tiles each, up to 512 workgroups per CU in the grid) even the workgroups out and keep a split's few tiles in
    // its XCD's L2: 1M <-> 1M 37.0 -> 34.5 ms, the 125k shard 5.14 -> 4.74 ms (tools/sweep_sorted_splits.py,
    // profiles/r05y_sorted_splits.txt).  WITHOUT seeds every split has to find a best of its own first, and more splits cost
    // (52 -> 80 ms at 136): the first search of a loop keeps the count above.
    int seeded = std::min(tiles_total / 3, (c->n_cu * 512 + src_blocks - 1) / src_blocks);
    if (seeded > 8) seeded = std::min(tiles_total, ((seeded + 7) / 8) * 8);
    seeded = std::max(splits, seeded);
    const int forced_seeded = env_int("OA_NN_SPLITS_SEEDED", 0);
    if (forced > 0) seeded = splits;
    if (forced_seeded > 0) seeded = std::min(forced_seeded, tiles_total);
    c->n_splits_seeded = seeded;
    const int acc_cap = std::max(1, std::min(oa::ACC_MAX_BLOCKS, env_int("OA_ACC_BLOCKS", 512)));
    c->acc_blocks = std::max(1, std::min(acc_cap, (c->ns + oa::ACC_THREADS - 1) / oa::ACC_THREADS));
}

int ensure_common(oa_ctx *c)
{
    if (!c->d_state) HIPCHK(dev_malloc(&c->d_state, sizeof(oa::DevState)));
    if (!c->d_partials) HIPCHK(dev_malloc(&c->d_partials, sizeof(double) * oa::NSUMS * oa::ACC_MAX_BLOCKS));
    if (!c->d_sums) HIPCHK(dev_malloc(&c->d_sums, sizeof(double) * oa::NSUMS));
    if (!c->d_solve) HIPCHK(dev_malloc(&c->d_solve, sizeof(double) * 32));
    if (!c->h_poll) {
        HIPCHK(hipHostMalloc((void **)&c->h_poll, 8 * sizeof(int32_t), hipHostMallocMapped));
        for (int k = 0; k < 8; ++k) c->h_poll[k] = 0;
    }
    if (!c->h_state_pin) HIPCHK(hipHostMalloc((void **)&c->h_state_pin, sizeof(oa::DevState), hipHostMallocDefault));
    return OA_OK;
}

// small device -> host read-back through a pinned scratch buffer, then a stream sync (a pageable destination costs
// ~30 us per copy; uploads do half a dozen of these)
// Small per-workgroup results (bounding-box partials, block maxima, counts) that the HOST reduces: the kernel stores them
// straight into pinned, device-mapped host memory, the host waits for the stream and reads -- no device buffer, no copy
// operation (an upload is bound by its number of enqueued operations, ~5 us each, and every read-back was two).
// result_ptr: where the kernel writes (device view) / where the host reads (host view); nullptr when `bytes` do not fit.
constexpr size_t RESULT_BYTES = 1 << 16;
int result_buffer(oa_ctx *c, size_t bytes, void **dev_view, const void **host_view)
{
    *dev_view = nullptr; *host_view = nullptr;
    if (bytes > RESULT_BYTES || !env_int("OA_MAPPED_RESULTS", 1)) return OA_OK;
    if (!c->h_result) {
        HIPCHK(hipHostMalloc((void **)&c->h_result, RESULT_BYTES, hipHostMallocMapped));
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->h_result, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(c->h_result); c->h_result = nullptr; return OA_OK; }
        c->d_result_view = dp;
    }
    *dev_view = c->d_result_view; *host_view = c->h_result;
    return OA_OK;
}

int read_small(oa_ctx *c, void *dst, const void *d_src, size_t bytes)
{
    constexpr size_t SCRATCH = 1 << 16;
    if (!c->h_scratch) HIPCHK(hipHostMalloc((void **)&c->h_scratch, SCRATCH, hipHostMallocDefault));
    if (bytes <= SCRATCH) {
        HIPCHK(hipMemcpyAsync(c->h_scratch, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(dst, c->h_scratch, bytes);
    } else {
        HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return OA_OK;
}

int ensure_history(oa_ctx *c, int n)
{
    n = std::max(16, std::min(n, 1 << 16));
    if (n <= c->max_records) return OA_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->h_hist_map) { (void)hipHostFree(c->h_hist_map); c->h_hist_map = nullptr; c->d_hist = nullptr; }
    HIPCHK(hipHostMalloc((void **)&c->h_hist_map, sizeof(oa::StepRecord) * (size_t)n, hipHostMallocMapped));
    void *dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, c->h_hist_map, 0));
    c->d_hist = (oa::StepRecord *)dp;
    c->max_records = n;
    return OA_OK;
}

int ensure_events(oa_ctx *c, int n_pairs)
{
    while ((int)c->ev.size() < 2 * n_pairs) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    if (!c->ev_loop0) { HIPCHK(hipEventCreate(&c->ev_loop0)); HIPCHK(hipEventCreate(&c->ev_loop1)); }
    return OA_OK;
}

int check_ready(oa_ctx *c)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (c->nt <= 0 || !c->d_tg) return fail(OA_E_STATE, "target not set (oa_set_target)");
    if (!c->d_src4) return fail(OA_E_STATE, "source not set (oa_set_source)");
    if (!c->have_mats) return fail(OA_E_STATE, "matrices not set (oa_set_matrices)");
    return OA_OK;
}

bool grid_active(const oa_ctx *c);
// every query through the tree: on request, and in auto mode for shards of up to `auto_max` points -- one wave per
// query has far lower latency than the one-thread-per-query grid kernels until the waves no longer fit the chip
// (measured crossover after the round-2 grid kernels: 3e3 .. 1.4e4 points for vertices -- the grid kernel spreads a
// query over 2 or 4 lanes when the shard is small --, 1.5e4 .. 2e4 for triangles, later for big targets whose grid no
// longer sits in cache; profiles/r02l_search_mode_crossover.txt.  Round 1: 1.2e4 .. 2.4e4 and 2.8e4 .. 4e4.)
inline int vertex_tree_max(const oa_ctx *c) { return c->nt >= 500000 ? 14336 : 3072; }
inline int tri_tree_max(const oa_ctx *c) { return c->n_tris >= 1000000 ? 20480 : (c->n_tris >= 250000 ? 18432 : 15360); }
// ... and while the pose still moves by a good part of a cell per iteration (stale seeds, long reach: the first
// iterations of a run, i.e. ALL of a typical early-exit run) the tree wins up to much larger shards (5-iteration runs
// from a cold start, profiles/r02l_search_mode_crossover_short_runs.txt): shards between the two limits get both
// searches enqueued and DevState::tree_turn decides on the device
inline int vertex_tree_early(const oa_ctx *c) { return c->nt >= 500000 ? 28672 : 8192; }
inline int tri_tree_early(const oa_ctx *c) { return c->n_tris >= 1000000 ? 57344 : (c->n_tris >= 250000 ? 40960 : 22528); }
inline bool bvh_whole(const oa_ctx *c, bool ok, int auto_max)
{
    if (!ok) return false;
    return c->grid_mode == 2 || (c->grid_mode == -1 && c->ns <= auto_max);
}
int build_grid(oa_ctx *c);
int build_safe_radii(oa_ctx *c);
// Stable argsort of 30-bit Morton keys on the context's stream, no wait: v_out = the indices in the keys' stable ascending order
// (v_in holds 0, 1, 2, ... at every call site and k_out, the sorted keys, is read by nobody: both stay in the signature for the
// call sites' sake).  The library's own sorts (oa_sort.hpp): one workgroup up to SORT_SMALL_MAX keys, the three-pass LSD argsort
// above that; rounds 4-5 still fell back on rocprim outside 65k .. 3M keys -- 628 of the library's 749 kernels were its
// instantiations.
int sort_order_bits(oa_ctx *c, const unsigned *keys, int *order, size_t n, int bits)
{
    DevTmp<char> tmp;
    if (n > (size_t)oa::SORT_SMALL_MAX) HIPCHK(tmp.alloc(oa::sort_order_tmp_bytes(n)));
    HIPCHK(oa::sort_order((void *)tmp.p, keys, order, n, bits, c->stream));
    return OA_OK;
}
int sort_pairs30(oa_ctx *c, const unsigned *k_in, unsigned *, const int *, int *v_out, size_t n)
{
    return sort_order_bits(c, k_in, v_out, n, 30);
}
// keys[0 .. n) (non-negative ints below 2^bits) -> sorted[] ascending and order[] = where each came from (stable)
int sort_ints(oa_ctx *c, const int *keys, int *sorted, int *order, size_t n, int bits)
{
    int rc = sort_order_bits(c, (const unsigned *)keys, order, n, bits);
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(oa::k_gather_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, keys, (const int *)order, (int)n, sorted);
    HIPCHK(hipGetLastError());
    return OA_OK;
}
inline int bits_for(long long n_values) { int b = 1; while (b < 32 && (1ll << b) < n_values) ++b; return b; }
constexpr long long SAFE_LAZY_ITERS = 8;   // OA_GRID_SAFE=1: loop iterations a target has to see before its safe radii are built
// OA_GRID_SAFE=1: the radii are built once the target has seen SAFE_LAZY_ITERS iterations of a loop (counted once per iteration,
// launch_search_accumulate).  The array is allocated with the grid; the build is ONE launch on the context's stream and no
// allocation -- so it may sit between two iterations of a running loop, on the caller's stream or a multi-device child's
// (round 4 allocated here: inside a group's loop an allocation can wait for another child's stream, whose gather kernel waits
// for the post this thread has not enqueued yet -- children therefore built between loops only).
int safe_radii_lazy(oa_ctx *c)
{
    if (c->grid_safe != 1 || c->safe_ok || !c->d_safe_by_idx) return OA_OK;
    return c->target_iters > SAFE_LAZY_ITERS ? build_safe_radii(c) : OA_OK;
}
int build_sorted_images(oa_ctx *c);
int build_tri_grid(oa_ctx *c, const double *diag_sum_known = nullptr);
int build_tri_ring(oa_ctx *c);
int build_tri_fine(oa_ctx *c);
// OA_TRI_RING=1: the neighbour lists are built once the mesh has served TRI_RING_LAZY_ITERS searches of a loop -- while the pose
// still moves by a good part of a triangle no query is within its seed's accept radius, and a short early-exit call should not
// pay for the build.  The buffer is allocated with the grid (never inside a loop: an allocation can wait for another stream,
// see safe_radii_lazy); the build itself is one launch on the context's stream, the searches queue behind it.
constexpr long long TRI_RING_LAZY_ITERS = 4;
int tri_ring_lazy(oa_ctx *c, bool counting)
{
    if (c->tri_ring != 1 || c->tri_ring_ok || !c->d_tri_ring) return OA_OK;
    if (counting) ++c->tri_iters;
    return c->tri_iters > TRI_RING_LAZY_ITERS ? build_tri_ring(c) : OA_OK;
}
int build_bvh(oa_ctx *c, bool tri);
int scan_counts(oa_ctx *c, const int *d_counts, int n, long long *d_off, DevTmp<char> &tmp);
int launch_tri_search(oa_ctx *c, bool acc = false);

// one wave per query: 4 queries per workgroup, workgroups loop when there are more queries than that
inline unsigned bvh_blocks(const oa_ctx *c, bool listed, bool acc = false)
{
    const int items = listed ? std::min(c->ns, 1 << 17) : c->ns;
    if (acc) return (unsigned)std::max(1, std::min((items + 15) / 16, c->n_cu * 4));   // workgroups of 16 waves
    return (unsigned)std::max(1, std::min((items + 3) / 4, c->n_cu * (listed ? c->list_blocks_per_cu : 16)));   // 16 waves per SIMD: enough to fill the chip, cheap to dispatch when the list is empty
}

oa::NormalTest normal_test(const oa_ctx *c)
{
    oa::NormalTest nrm{};
    if (c->normals_on) { nrm.src_n = c->d_src_n; nrm.tgt_n = c->surface ? nullptr : c->d_tgt_n; nrm.cos_min = c->cos_min; }
    return nrm;
}

// acc: a whole-shard search of the loop that also takes the pair test and the sums (k_bvh_search<TRI, true>)
template <bool TRI>
int launch_bvh(oa_ctx *c, const int *list, const int *list_count, int turn = -1, bool acc = false)
{
    const unsigned blocks = bvh_blocks(c, list != nullptr, acc);
    // whole-shard vertex searches take seeds on their safe radii (docs/HISTORY.md 4.4 / 4.8): both arrays or neither
    const float *safe_by_idx = nullptr;
    uint2 *wsafe = nullptr;
    if (!TRI && !list && c->grid_safe) {
        if (acc) { const int rcs = safe_radii_lazy(c); if (rcs) return rcs; }
        if (c->safe_ok && c->d_safe_by_idx && c->d_wsafe) { safe_by_idx = c->d_safe_by_idx; wsafe = c->d_wsafe; }
    }
#define OA_BVH_ARGS c->d_state, c->d_src4, c->ns, TRI ? c->tbvh : c->bvh, TRI ? c->d_tbvh_box : c->d_bvh_box, TRI ? c->d_tbvh_prims : c->d_bvh_prims, \
                    c->d_tri9, c->d_prev, TRI ? (float4 *)nullptr : c->d_win, c->d_keys, list, list_count, turn
    if (acc) hipLaunchKernelGGL((oa::k_bvh_search<TRI, true>), dim3(blocks), dim3(1024), 0, c->stream, OA_BVH_ARGS, normal_test(c), c->d_partials, safe_by_idx, wsafe);
    else hipLaunchKernelGGL((oa::k_bvh_search<TRI, false>), dim3(blocks), dim3(256), 0, c->stream, OA_BVH_ARGS, oa::NormalTest{}, (double *)nullptr, safe_by_idx, wsafe);
#undef OA_BVH_ARGS
    HIPCHK(hipGetLastError());
    return OA_OK;
}

// lanes per query of the vertex grid search: shards too small to fill the chip's wave slots split every query's rows
// between 2 or 4 lanes.  Measured on 256 CUs with the round-2 kernel (profiles/r02n_grid_lanes_sweep.txt; round 1:
// r01h_grid_lanes_sweep.txt): against a target that sits in cache 4 lanes win up to ~20k queries and 2 up to ~350k;
// against >= 500k vertices every round trip is longer and the lanes pay for longer: 4 up to ~100k queries, 2 up to
// ~700k; 1M queries lose 7 % with 2
inline int grid_lanes_for(const oa_ctx *c)
{
    int lanes = c->grid_lanes;
    if (lanes != 1 && lanes != 2 && lanes != 4) {
        const bool big = c->nt >= 500000;
        lanes = (c->ns <= (big ? 400 : 80) * c->n_cu) ? 4 : ((c->ns <= (big ? 2800 : 1400) * c->n_cu) ? 2 : 1);
    }
    return lanes;
}

// lanes per query of the triangle grid search, as for the vertex grid (measured on 256 CUs with the round-2 kernel,
// profiles/r02n_grid_lanes_sweep.txt: small meshes 4 lanes up to ~32k queries, 2 up to ~180k; >= 250k triangles
// 4 up to ~100k, 2 up to ~800k)
inline int tri_lanes_for(const oa_ctx *c)
{
    int lanes = c->grid_lanes;
    if (lanes != 1 && lanes != 2 && lanes != 4) {
        const bool big = c->n_tris >= 250000;
        lanes = (c->ns <= (big ? 400 : 128) * c->n_cu) ? 4 : ((c->ns <= (big ? 3200 : 700) * c->n_cu) ? 2 : 1);
    }
    return lanes;
}
// does the loop's surface search go through the triangle grid (and not through the tree alone, or brute force)?
inline bool tri_grid_active(const oa_ctx *c)
{
    return c->surface && c->tri_grid_ok && c->tbvh_ok && c->grid_mode != 0 && !bvh_whole(c, c->tbvh_ok, tri_tree_max(c));
}

// Workgroups of the canonical accumulation (k_pair_accumulate_canon, and the epilogue of k_nn_search_grid<L, true>): one
// thread per (slot, lane of the query).  0 = the shard is too large for it (> ACC_MAX_BLOCKS rows before combining) or the
// target is a surface: the grid-stride k_pair_accumulate is used instead.
// (512 threads per workgroup from 262k (query, lane) pairs on: half the rows for the reduction behind it; below that the
//  finer workgroups balance better)
// (surface targets, round 4: the triangle grid search accumulates in its epilogue too -- k_tri_search_grid<L, .., ACC> --, always
//  in workgroups of 256 threads, with ITS lanes per query)
inline int canon_lanes(const oa_ctx *c) { return c->surface ? tri_lanes_for(c) : grid_lanes_for(c); }
inline int canon_threads(const oa_ctx *c)
{
    if (c->surface) return 256;
    if (c->acc_threads == 256 || c->acc_threads == 512) return c->acc_threads;   // OA_ACC_THREADS (A/B)
    return (long long)c->ns * grid_lanes_for(c) >= 262144 ? 512 : 256;
}
inline int canon_blocks(const oa_ctx *c)
{
    if (c->ns <= 0 || (c->surface && !c->tri_canon)) return 0;
    const int t = canon_threads(c);
    const long long b = ((long long)c->ns * canon_lanes(c) + t - 1) / t;
    return b <= oa::ACC_MAX_BLOCKS ? (int)b : 0;
}

// What the loop's next search launch looks like.
//   TREE   every query through the tree, which also accumulates (k_bvh_search<.., true>)
//   DUAL   tree and grid both enqueued, DevState::tree_turn picks on the device; both accumulate; the grid search finishes
//          its leftovers itself (it only has the turn once the pose has settled)
//   GRID   the grid search.  FAST: it finishes its leftovers itself and accumulates -- two launches per iteration.  SAFE:
//          grid search -> tree search of the hand-over list -> k_pair_accumulate_canon -- four launches, and the right
//          thing when many queries are handed over (one wave per query over the whole chip instead of the owner wave
//          walking its own leftovers one after the other: a cold start far from the target is 3-4x slower the FAST way).
//          Both leave bitwise the same rows, so the choice -- taken per iteration from what the host last heard about the
//          hand-over list, i.e. timing dependent -- never shows in a result.
//   PLAIN  search, then accumulate (brute force; surface grid; shards too large for the canonical rows)
enum SearchPlan { PLAN_PLAIN, PLAN_TREE, PLAN_DUAL, PLAN_GRID };
SearchPlan search_plan(const oa_ctx *c)
{
    if (!c->fused_acc || !c->loop_active || c->ns <= 0) return PLAN_PLAIN;
    // (the accumulating tree search needs twice the registers of the plain one: worth it while the shard is small enough
    //  that occupancy does not matter -- 12k queries against 1M vertices: 58 us fused, 47 us search + accumulate)
    const bool small = c->ns <= c->tree_acc_max;
    if (c->surface) {
        if (bvh_whole(c, c->tbvh_ok, tri_tree_max(c))) return small ? PLAN_TREE : PLAN_PLAIN;
        // the triangle grid search with the accumulating epilogue: shards above the zone where tree and grid take turns
        const bool dual = c->grid_mode == -1 && c->turns_on && c->ns <= tri_tree_early(c);
        // (with the neighbour lists: k_tri_accept + the search of what it leaves, which has no accumulating form)
        if (c->tri_ring_ok && c->tri_split && c->seeded) return PLAN_PLAIN;
        // (the settled-pose search in front, oa_tri_fine.hpp: likewise k_tri_settle + the search of what it leaves)
        if (c->tri_fine_ok && c->seeded && !dual && tri_grid_active(c)) return PLAN_PLAIN;
        return (tri_grid_active(c) && !dual && canon_blocks(c) > 0 && c->tri_acc) ? PLAN_GRID : PLAN_PLAIN;
    }
    if (bvh_whole(c, c->bvh_ok, vertex_tree_max(c))) return small ? PLAN_TREE : PLAN_PLAIN;
    if (!grid_active(c) || canon_blocks(c) == 0) return PLAN_PLAIN;
    return (c->grid_mode == -1 && c->turns_on && c->ns <= vertex_tree_early(c)) ? PLAN_DUAL : PLAN_GRID;
}

// PLAN_GRID: may this iteration's grid search finish its own leftovers?  Yes while the most any one wave handed over in
// the last iteration the host heard of is small (a wave walks its leftovers one after the other, ~5-10 us each).  The
// host stays within a few iterations of the device (oa_run throttles itself), so "last heard of" is recent; with no
// news from this loop yet the previous decision stands, and a loop starts SAFE unless it continues on warm seeds.
constexpr int FAST_WAVE_MAX = 3;
bool grid_fast_now(oa_ctx *c)
{
    if (c->grid_path == 1) return true;
    if (c->grid_path == 2) return false;
    bool fast = c->fast_prev;
    const volatile int32_t *poll = c->h_poll;
    const int n_done = poll ? poll[1] : 0;
    if (n_done > 0 && n_done >= c->iter_enq - 4) fast = poll[3] <= FAST_WAVE_MAX;
    else if (c->iter_enq == 0) fast = c->seeded && c->last_todo_wave_max >= 0 && c->last_todo_wave_max <= FAST_WAVE_MAX;
    else if (n_done <= 0 && c->iter_enq > 4) fast = false;          // the host ran far ahead of the device: no news, no risk
    if (c->debug) fprintf(stderr, "[oa] grid path: iteration %d enqueued, device at %d, last hand-over %d entries / %d per wave (previous loop: %d) -> %s\n", c->iter_enq, n_done, poll ? poll[2] : -1, poll ? poll[3] : -1, c->last_todo_wave_max, fast ? "fast" : "safe");
    c->fast_prev = fast;
    return fast;
}

int launch_nn_impl(oa_ctx *c, bool acc);
// acc: the search also accumulates (launch_search_accumulate decided so: no accumulation launch follows)
int launch_nn(oa_ctx *c, bool acc = false)
{
    const int rc = launch_nn_impl(c, acc);
    if (rc == OA_OK) { c->seeded = true; if (acc && !c->surface) c->win_seeds = true; }
    return rc;
}
int launch_nn_impl(oa_ctx *c, bool acc)
{
    if (c->ns <= 0) return OA_OK;
    if (c->surface) return launch_tri_search(c, acc);
    dim3 grid(c->n_splits, c->ns_pad / (oa::NN_THREADS * c->R));
    dim3 block(oa::NN_THREADS);
    if (bvh_whole(c, c->bvh_ok, vertex_tree_max(c))) return launch_bvh<false>(c, nullptr, nullptr, -1, acc);
    if (grid_active(c)) {
        // the grid search settles the queries near the target; the rest (far away, or in crowded cells) go through the
        // tree: in the loop by the wave that owns them (acc), in one-shot calls through a list that k_bvh_search finishes
        if (!acc && !c->loop_active) HIPCHK(hipMemsetAsync(c->d_todo_count, 0, 2 * sizeof(int), c->stream));
        const bool dual = c->grid_mode == -1 && c->turns_on && c->ns <= vertex_tree_early(c);
        if (dual) { int rcb = launch_bvh<false>(c, nullptr, nullptr, 1, acc); if (rcb) return rcb; }   // runs when DevState::tree_turn
        const int turn = dual ? 0 : -1;
        const int lanes = grid_lanes_for(c);
#define OA_GRID_ARGS c->d_state, c->d_src4, c->ns, c->gp, c->d_cell_start, c->d_sorted, c->d_win, c->d_keys, c->d_todo_list, c->d_todo_count, turn
#define OA_GRID_ACC_ARGS OA_GRID_ARGS, c->bvh, (const float4 *)c->d_bvh_box, (const float4 *)c->d_bvh_prims, normal_test(c), c->d_partials
        // {index, safe2} per slot beside the winner records: both or neither (OA_GRID_SAFE=0)
        if (acc) { const int rcs = safe_radii_lazy(c); if (rcs) return rcs; }
        const float *safe_by_idx = (c->grid_safe && c->safe_ok) ? c->d_safe_by_idx : nullptr;
        uint2 *wsafe = safe_by_idx ? c->d_wsafe : nullptr;
        if (!wsafe) safe_by_idx = nullptr;
#define OA_GRID_SAFE_ARGS safe_by_idx, wsafe
        const dim3 gblocks((unsigned)(((long long)c->ns * lanes + 255) / 256));
        if (acc) {
            const dim3 ablocks((unsigned)canon_blocks(c));
            if (canon_threads(c) == 512) {
                if (lanes == 4) hipLaunchKernelGGL((oa::k_nn_search_grid<4, true, 512>), ablocks, dim3(512), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
                else if (lanes == 2) hipLaunchKernelGGL((oa::k_nn_search_grid<2, true, 512>), ablocks, dim3(512), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
#if defined(OA_EXPERIMENTS)
                else if (c->grid_stats) {                                // OA_GRID_STATS=1: instrumented launch, phase shares to stderr (synchronises)
                    DevTmp<unsigned long long> d_stats;
                    const size_t n_waves = (size_t)ablocks.x * 8;
                    HIPCHK(d_stats.alloc(oa::GRID_STAT_N * n_waves));
                    HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * oa::GRID_STAT_N * n_waves, c->stream));
                    hipLaunchKernelGGL((oa::k_nn_search_grid<1, true, 512, true>), ablocks, dim3(512), 0, c->stream, OA_GRID_ACC_ARGS, d_stats.p, OA_GRID_SAFE_ARGS);
                    HIPCHK(hipGetLastError());
                    std::vector<unsigned long long> rows(oa::GRID_STAT_N * n_waves);
                    { int rcr = read_small(c, rows.data(), d_stats, sizeof(unsigned long long) * rows.size()); if (rcr) return rcr; }
                    unsigned long long h[oa::GRID_STAT_N] = { 0 };
                    for (size_t w = 0; w < n_waves; ++w) for (int k = 0; k < oa::GRID_STAT_N; ++k) h[k] += rows[w * oa::GRID_STAT_N + (size_t)k];
                    const double nw = (double)std::max(1ull, h[oa::GRID_STAT_WAVES]), ct = (double)std::max(1ull, h[oa::GRID_STAT_CYC_TOTAL]);
                    fprintf(stderr, "[oa] vertex grid phases: %llu of %d queries settled by the seed's safe radius | per wave %.0f shader cycles, %.2f loop trips, %.2f scan trips, %.1f candidates per query (longest lane of a wave %.1f) | "
                                    "prologue %.1f%% listing %.1f%% scan %.1f%% bookkeeping %.1f%% finish %.1f%% epilogue %.1f%% (loads + pair test %.1f%%, wave reduction %.1f%%, barrier %.1f%%)\n",
                            h[oa::GRID_STAT_ACCEPTED], c->ns, ct / nw, h[oa::GRID_STAT_LOOP_TRIPS] / nw, h[oa::GRID_STAT_SCAN_TRIPS] / nw, h[oa::GRID_STAT_CANDIDATES] / (64.0 * nw),
                            h[oa::GRID_STAT_MAX_LANE_CANDIDATES] / nw, 100.0 * h[oa::GRID_STAT_CYC_PROLOGUE] / ct, 100.0 * h[oa::GRID_STAT_CYC_LIST] / ct,
                            100.0 * h[oa::GRID_STAT_CYC_SCAN] / ct, 100.0 * h[oa::GRID_STAT_CYC_BOOK] / ct, 100.0 * h[oa::GRID_STAT_CYC_FINISH] / ct,
                            100.0 * h[oa::GRID_STAT_CYC_EPILOGUE] / ct, 100.0 * h[oa::GRID_STAT_CYC_EPI_PAIR] / ct, 100.0 * h[oa::GRID_STAT_CYC_EPI_REDUCE] / ct,
                            100.0 * h[oa::GRID_STAT_CYC_EPI_BARRIER] / ct);
                }
#endif
                else hipLaunchKernelGGL((oa::k_nn_search_grid<1, true, 512>), ablocks, dim3(512), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
            } else {
                if (lanes == 4) hipLaunchKernelGGL((oa::k_nn_search_grid<4, true, 256>), ablocks, dim3(256), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
                else if (lanes == 2) hipLaunchKernelGGL((oa::k_nn_search_grid<2, true, 256>), ablocks, dim3(256), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
                else hipLaunchKernelGGL((oa::k_nn_search_grid<1, true, 256>), ablocks, dim3(256), 0, c->stream, OA_GRID_ACC_ARGS, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
            }
            HIPCHK(hipGetLastError());
            return OA_OK;
        }
        if (lanes == 4) hipLaunchKernelGGL((oa::k_nn_search_grid<4, false>), gblocks, dim3(256), 0, c->stream, OA_GRID_ARGS, oa::BvhParams{}, (const float4 *)nullptr, (const float4 *)nullptr, oa::NormalTest{}, (double *)nullptr, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
        else if (lanes == 2) hipLaunchKernelGGL((oa::k_nn_search_grid<2, false>), gblocks, dim3(256), 0, c->stream, OA_GRID_ARGS, oa::BvhParams{}, (const float4 *)nullptr, (const float4 *)nullptr, oa::NormalTest{}, (double *)nullptr, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
        else hipLaunchKernelGGL((oa::k_nn_search_grid<1, false>), gblocks, dim3(256), 0, c->stream, OA_GRID_ARGS, oa::BvhParams{}, (const float4 *)nullptr, (const float4 *)nullptr, oa::NormalTest{}, (double *)nullptr, (unsigned long long *)nullptr, OA_GRID_SAFE_ARGS);
#undef OA_GRID_SAFE_ARGS
#undef OA_GRID_ACC_ARGS
#undef OA_GRID_ARGS
        HIPCHK(hipGetLastError());
        return launch_bvh<false>(c, c->d_todo_list, c->d_todo_count);
    }
    if (c->ns_pad / (oa::NN_THREADS * c->R) > 65535)               // only the brute-force launch has this limit (grid.y)
        return fail(OA_E_BAD_ARG, "shard of %d points exceeds the brute-force launch grid (use more shards)", c->ns);
#define OA_NN_ARGS c->d_state, c->d_src4, c->d_tg, c->n_groups_pad, c->d_keys
#define OA_NNF_ARGS c->d_state, c->d_src4, c->d_tg, c->d_tf, c->d_tf3, (const float4 *)c->d_win, c->n_groups_pad, c->d_keys
#if defined(OA_EXPERIMENTS)
    if (c->filter_ok && c->use_filter && c->nn_mfma && c->d_tfm && c->R == 4 && c->tile_groups == oa::FTILE_GROUPS) {
#define OA_MFMA_ARGS c->d_state, c->d_src4, c->d_tg, (const oa::half8 *)c->d_tfm, (const float4 *)c->d_win, c->n_groups_pad, c->mfma_sigma, c->d_keys
        if (c->mfma_wps == 2) hipLaunchKernelGGL(oa::k_nn_search_mfma<2>, grid, block, 0, c->stream, OA_MFMA_ARGS);
        else if (c->mfma_wps == 3) hipLaunchKernelGGL(oa::k_nn_search_mfma<3>, grid, block, 0, c->stream, OA_MFMA_ARGS);
        else hipLaunchKernelGGL(oa::k_nn_search_mfma<4>, grid, block, 0, c->stream, OA_MFMA_ARGS);
#undef OA_MFMA_ARGS
    } else
#endif
    if (c->filter_ok && c->use_filter && c->nn_sort && c->d_tfs) {
        const bool small = (c->tile_groups == 64);
#define OA_NNS_ARGS c->d_state, c->d_src4, (const float4 *)c->d_tgs, (const float4 *)c->d_tfs, (const float4 *)c->d_tf3s, (const int4 *)c->d_tidx, \
                    (const float4 *)c->d_win, c->n_groups_pad, c->sax[0], c->sax[1], c->d_keys
        // with seeds: one launch over n_splits_seeded splits.  Without (the first search of a loop): k_nn_seed_sorted -- every point
        // against the tile that holds its own slab --, then the whole search seeded from what that left in keys (OA_NN_HOME_PASS=0:
        // one unseeded launch over n_splits splits, as until r05z)
        const bool two = !c->win_seeds && c->nn_home_pass && c->n_splits_seeded > 1;
        dim3 sgrid((unsigned)((c->win_seeds || two) ? c->n_splits_seeded : c->n_splits), grid.y);
        int pass = 0;
#define OA_LAUNCH_S(RR)                                                                                              \
        do {                                                                                                         \
            if (small) hipLaunchKernelGGL((oa::k_nn_search_sorted<RR, 64>), sgrid, block, 0, c->stream, OA_NNS_ARGS, pass, worder, q_splits, q_blocks, homes, qcnt);  \
            else hipLaunchKernelGGL((oa::k_nn_search_sorted<RR, oa::FTILE_GROUPS>), sgrid, block, 0, c->stream, OA_NNS_ARGS, pass, worder, q_splits, q_blocks, homes, qcnt); \
        } while (0)
#if defined(OA_EXPERIMENTS)
#define OA_LAUNCH_S_CASE8 case 8: OA_LAUNCH_S(8); break;
#else
#define OA_LAUNCH_S_CASE8
#endif
#define OA_LAUNCH_S_R()                     \
        switch (c->R) {                     \
        case 1: OA_LAUNCH_S(1); break;      \
        case 2: OA_LAUNCH_S(2); break;      \
        OA_LAUNCH_S_CASE8                   \
        default: OA_LAUNCH_S(4); break;     \
        }
        // the waves' slots in the order of u at this pose (k_sorted_wave_order: ~10 us in front of a 30 ms search)
        const unsigned short *worder = nullptr;
        if (c->d_worder && c->R >= 2) {
            const dim3 ob((unsigned)(c->ns_pad / (64 * c->R)));
            switch (c->R) {
            case 2: hipLaunchKernelGGL(oa::k_sorted_wave_order<2>, ob, dim3(64), 0, c->stream, (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->sax[0], c->d_worder); break;
            case 8: hipLaunchKernelGGL(oa::k_sorted_wave_order<8>, ob, dim3(64), 0, c->stream, (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->sax[0], c->d_worder); break;
            default: hipLaunchKernelGGL(oa::k_sorted_wave_order<4>, ob, dim3(64), 0, c->stream, (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->sax[0], c->d_worder); break;
            }
            HIPCHK(hipGetLastError());
            worder = c->d_worder;
        }
        // long launches go through the work queue (oa_kernels.hpp): as many workgroups as the chip holds, the long items first
        const int q_splits = (int)sgrid.x, q_blocks = (int)sgrid.y;
        const long long q_wgs = (long long)c->n_cu * std::min(c->nn_persist, c->R <= 4 ? 4 : 2);
        const bool queued = q_wgs > 0 && q_splits > 1 && (long long)q_splits * q_blocks >= (c->nn_queue_min >= 0 ? (long long)c->nn_queue_min : 4 * q_wgs);
        c->last_queue_wgs = queued ? (int)q_wgs : 0;
        if (queued) {
            if (!c->d_homes) HIPCHK(dev_malloc(&c->d_homes, sizeof(int) * (size_t)q_blocks));
            if (!c->d_qcnt) HIPCHK(dev_malloc(&c->d_qcnt, sizeof(int) * (size_t)(oa::SORTED_QUEUES * oa::SORTED_QUEUE_STRIDE)));
            hipLaunchKernelGGL(oa::k_sorted_block_homes, dim3((unsigned)((std::max(q_blocks, oa::SORTED_QUEUES) + 63) / 64)), dim3(64), 0, c->stream,
                               (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, q_blocks, oa::NN_THREADS * c->R, (const float4 *)c->d_tfs,
                               c->n_groups_pad, c->tile_groups, c->sax[0], q_splits, c->d_homes, c->d_qcnt);
            HIPCHK(hipGetLastError());
            sgrid = dim3((unsigned)q_wgs);
        }
        const int *homes = queued ? c->d_homes : nullptr;
        int *qcnt = queued ? c->d_qcnt : nullptr;
        if (two) {
            const dim3 sb((unsigned)((c->ns_pad + 255) / 256));
            if (small) hipLaunchKernelGGL((oa::k_nn_seed_sorted<64>), sb, dim3(256), 0, c->stream, (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->ns_pad,
                                          (const float4 *)c->d_tgs, (const float4 *)c->d_tfs, (const int4 *)c->d_tidx, c->n_groups_pad, c->sax[0], c->d_keys);
            else hipLaunchKernelGGL((oa::k_nn_seed_sorted<oa::FTILE_GROUPS>), sb, dim3(256), 0, c->stream, (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->ns_pad,
                                    (const float4 *)c->d_tgs, (const float4 *)c->d_tfs, (const int4 *)c->d_tidx, c->n_groups_pad, c->sax[0], c->d_keys);
            HIPCHK(hipGetLastError());
            pass = 2;
        }
        OA_LAUNCH_S_R();
#undef OA_LAUNCH_S_R
#undef OA_LAUNCH_S_CASE8
#undef OA_LAUNCH_S
#undef OA_NNS_ARGS
    }
#if defined(OA_EXPERIMENTS)
    else if (c->filter_ok && c->use_filter) {
        const bool small = (c->tile_groups == 64);
#define OA_LAUNCH_F(RR)                                                                                              \
        do {                                                                                                         \
            if (small) hipLaunchKernelGGL((oa::k_nn_search_filtered<RR, 64>), grid, block, 0, c->stream, OA_NNF_ARGS); \
            else hipLaunchKernelGGL((oa::k_nn_search_filtered<RR, oa::FTILE_GROUPS>), grid, block, 0, c->stream, OA_NNF_ARGS); \
        } while (0)
        switch (c->R) {
        case 1: OA_LAUNCH_F(1); break;
        case 2: OA_LAUNCH_F(2); break;
        case 8: OA_LAUNCH_F(8); break;
        default: OA_LAUNCH_F(4); break;
        }
#undef OA_LAUNCH_F
    }
#endif
    else {
        switch (c->R) {
        case 1: hipLaunchKernelGGL(oa::k_nn_search<1>, grid, block, 0, c->stream, OA_NN_ARGS); break;
        case 2: hipLaunchKernelGGL(oa::k_nn_search<2>, grid, block, 0, c->stream, OA_NN_ARGS); break;
        case 8: hipLaunchKernelGGL(oa::k_nn_search<8>, grid, block, 0, c->stream, OA_NN_ARGS); break;
        default: hipLaunchKernelGGL(oa::k_nn_search<4>, grid, block, 0, c->stream, OA_NN_ARGS); break;
        }
    }
#undef OA_NN_ARGS
#undef OA_NNF_ARGS
    HIPCHK(hipGetLastError());
    return OA_OK;
}

int launch_accumulate(oa_ctx *c, bool emit, int *nn_idx, float *nn_d2)
{
    if (!c->surface) c->win_seeds = true;                          // (the winner records of this pass seed the next search)
    oa::PairOut po{};
    const oa::NormalTest nrm = normal_test(c);
    if (emit) {
        po.valid = c->d_valid; po.b = c->d_b; po.dist = c->d_dist; po.nn_idx = nn_idx; po.nn_d2 = nn_d2; po.perm = c->d_perm;
        hipLaunchKernelGGL(oa::k_pair_accumulate<true>, dim3(c->acc_blocks), dim3(oa::ACC_THREADS), 0, c->stream,
                           c->d_state, c->d_src4, c->ns, c->d_tgt_xyz, c->d_keys, c->d_prev, c->surface ? (float4 *)nullptr : c->d_win,
                           c->surface ? (const float4 *)c->d_tri9 : (const float4 *)nullptr, nrm, c->d_partials, po,
                           (unsigned long long *)nullptr);
    } else if (canon_blocks(c) > 0) {
        hipLaunchKernelGGL(oa::k_pair_accumulate_canon, dim3((unsigned)canon_blocks(c)), dim3((unsigned)canon_threads(c)), 0, c->stream, (const oa::DevState *)c->d_state,
                           (const float4 *)c->d_src4, c->ns, canon_lanes(c), (const float *)c->d_tgt_xyz, c->d_keys, c->d_prev, c->surface ? (float4 *)nullptr : c->d_win,
                           c->surface ? (const float4 *)c->d_tri9 : (const float4 *)nullptr, nrm, c->d_partials,
                           c->loop_active ? &c->d_state->t_acc_start : (unsigned long long *)nullptr);
    } else {
        hipLaunchKernelGGL(oa::k_pair_accumulate<false>, dim3(c->acc_blocks), dim3(oa::ACC_THREADS), 0, c->stream,
                           c->d_state, c->d_src4, c->ns, c->d_tgt_xyz, c->d_keys, c->d_prev, c->surface ? (float4 *)nullptr : c->d_win,
                           c->surface ? (const float4 *)c->d_tri9 : (const float4 *)nullptr, nrm, c->d_partials, po,
                           c->loop_active ? &c->d_state->t_acc_start : (unsigned long long *)nullptr);
    }
    HIPCHK(hipGetLastError());
    return OA_OK;
}

// rows the (non-emitting) accumulation of this context leaves for the reduce launch when it is its own kernel
inline oa::RowSel plain_rows(const oa_ctx *c)
{
    const int cb = canon_blocks(c);
    return oa::RowSel{ cb > 0 ? cb : c->acc_blocks, 0, 0 };
}

// rows of partials -> d_sums.  stamp: the launch marks the end of the search + accumulate part (fused searches)
int launch_reduce(oa_ctx *c, double *d_sums, const oa::RowSel &sel, bool stamp)
{
    hipLaunchKernelGGL(oa::k_reduce_partials, dim3(1), dim3(oa::RED_THREADS), 0, c->stream, (const oa::DevState *)c->d_state,
                       (const double *)c->d_partials, sel, d_sums, stamp ? &c->d_state->t_acc_start : (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    return OA_OK;
}

// search + accumulate of one iteration, by the plan above; sel = the rows the reduce launch finds; fused = no separate
// accumulation launch ran (the reduce launch then stamps the end of the search part)
int launch_search_accumulate(oa_ctx *c, bool timed, oa::RowSel &sel, bool &fused)
{
    int rc;
    // hipEvent pairs around the search cost ~7 us of stream time per iteration (two barrier packets): fine for the
    // brute-force kernel (50 ms per launch), not for the grid / tree searches, whose launches are that short
    // themselves -- those are timed on the GPU instead (DevState::t_prev_end / t_acc_start, StepRecord::search_ticks)
    timed = timed && c->time_events;
    const SearchPlan plan = search_plan(c);
    fused = plan == PLAN_TREE || plan == PLAN_DUAL || (plan == PLAN_GRID && grid_fast_now(c));
    c->iter_enq++;
    // the lazy safe radii: counted once per iteration, whatever the plan launches (tree and grid in turns count once; shards too
    // large for the fused path -- BASELINE config 5 on one GPU -- count too: their plain grid search takes seeds on their radii
    // all the same); brute force never reads them
    if (!c->surface && c->grid_ok && c->grid_mode != 0) { ++c->target_iters; if ((rc = safe_radii_lazy(c))) return rc; }
    if (plan == PLAN_GRID && fused) c->fast_iters++;
    if (timed) {
        if ((rc = ensure_events(c, c->ev_used + 1))) return rc;
        HIPCHK(hipEventRecord(c->ev[2 * c->ev_used], c->stream));
    }
    if ((rc = launch_nn(c, fused))) return rc;
    if (timed) {
        HIPCHK(hipEventRecord(c->ev[2 * c->ev_used + 1], c->stream));
        c->ev_used++;
    }
    if (!fused) {
        if ((rc = launch_accumulate(c, false, nullptr, nullptr))) return rc;
        sel = plain_rows(c);
        return OA_OK;
    }
    const int tree_rows = (int)bvh_blocks(c, false, true);
    if (plan == PLAN_TREE) sel = oa::RowSel{ tree_rows, 0, 0 };
    else sel = oa::RowSel{ canon_blocks(c), plan == PLAN_DUAL ? tree_rows : 0, plan == PLAN_DUAL ? 1 : 0 };
    return OA_OK;
}

// smallest singular value of the upper-left 3x3 of a row-major 4x4 (cyclic Jacobi on M^T M, double)
double min_singular_3x3(const float *M)
{
    double S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += (double)M[4 * k + i] * (double)M[4 * k + j];
            S[i][j] = a;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (!(off > 1e-300)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (S[p][q] == 0.0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) { const double a = S[k][p], b = S[k][q]; S[k][p] = cs * a - sn * b; S[k][q] = sn * a + cs * b; }
                for (int k = 0; k < 3; ++k) { const double a = S[p][k], b = S[q][k]; S[p][k] = cs * a - sn * b; S[q][k] = sn * a + cs * b; }
            }
    }
    double m = std::min(S[0][0], std::min(S[1][1], S[2][2]));
    if (!(m > 0.0)) return 0.0;
    return sqrt(m);
}

void init_loop_state(oa_ctx *c, const oa_settings *st, int iters, bool cutoff = true)
{
    oa::DevState &s = c->h_state;
    // search radius (DevState::cut_a / cut_b): only the grid / tree searches use it
    s.cut_a = INFINITY; s.cut_b = 0.0;
    s.local_per_world = 0.0;
    s.host_halt = nullptr;
    c->poll_mapped = false;
    s.t_prev_end = 0; s.t_acc_start = 0;
    s.t_xchg_start = 0;                                             // (a stale stamp would leak into the first exchange_ticks of this loop)
    s.search_clk[0] = 0; s.search_clk[1] = 0;                       // OA_STAT_SEARCH_CLOCK_MHZ: 0 unless THIS loop's search stamps it
    {   // first search of a loop: the tree (cold or stale seeds); a one-shot call on warm seeds: the grid
        const bool grid_up = c->surface ? c->tri_grid_ok : c->grid_ok;
        const oa::GridParams &g = c->surface ? c->tgp : c->gp;
        s.turn_limit = grid_up ? c->turn_frac * g.h : 0.0;
        s.turn_scale = grid_up ? g.scale : 0.0;
        s.tree_turn = (iters > 1 || !c->seeded) ? 1 : 0;
        static const float eye16[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        s.mx2_identity = (memcmp(s.mx2, eye16, sizeof eye16) == 0 && !env_int("OA_NO_IDENTITY_PATH", 0)) ? 1 : 0;
    }
    if (c->h_poll) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->h_poll, 0) == hipSuccess) { s.host_halt = (int32_t *)dp; c->poll_mapped = true; }
        else (void)hipGetLastError();
    }
    if (cutoff && c->filter_ok && c->grid_mode != 0 && env_int("OA_NN_CUTOFF", 1)) {
        const double smin = min_singular_3x3(s.mx2) * (1.0 - 1e-9);
        if (smin > 0.0) s.local_per_world = 1.0 / smin;
        double m2norm = 0.0, tmax = 0.0, tscale = 0.0;
        for (int i = 0; i < 3; ++i) {
            m2norm = std::max(m2norm, fabs((double)s.mx2[4 * i]) + fabs((double)s.mx2[4 * i + 1]) + fabs((double)s.mx2[4 * i + 2]));
            tmax = std::max(tmax, fabs((double)s.mx2[4 * i + 3]));
            tscale = std::max(tscale, std::max(fabs(c->bb_lo[i]), fabs(c->bb_hi[i])));
        }
        const double u64 = 64.0 * 5.9604644775390625e-08;
        if (smin > 0.0 && st->thresh > 0.0 && st->thresh < 1e300) {
            s.cut_a = (st->thresh + u64 * (m2norm * 3.0 * tscale + tmax)) / smin * (1.0 + 1e-5);
            s.cut_b = u64 * m2norm / smin * (1.0 + 1e-5);
            if (!(s.cut_a < 1e300) || !(s.cut_b < 1e300)) { s.cut_a = INFINITY; s.cut_b = 0.0; }
        }
    }
    for (int k = 0; k < 3; ++k) s.pivot[k] = c->pivot[k];
    s.thresh = st->thresh;
    s.target_d = st->target_d;
    for (int k = 0; k < 5; ++k) { s.ring_t[k] = st->target_d * 2.0; s.ring_r[k] = 0.0; }   // icp_align.py:93-94
    s.iters = iters;
    s.use_target = st->use_target ? 1 : 0;
    s.with_scale = st->with_scale ? 1 : 0;
    s.early_exit = st->early_exit ? 1 : 0;
    s.n = 0; s.converged = 0; s.status = 0; s.halt = (iters <= 0) ? 1 : 0;
    s.max_records = c->max_records; s.pad0 = 0;
    for (int k = 0; k < 3; ++k) { s.tc[k] = c->tc[k]; s.fax[k] = c->fax[k]; }
    s.pad1 = 0.f; s.pad2 = 0;
    s.qmax = c->qmax;
    s.d_pivot = c->d_pivot0;
    s.jac_valid = 0; s.pad4 = 0;                                    // every loop starts its Jacobi from the identity
    for (int k = 0; k < 9; ++k) s.jac_v[k] = 0.0;
}

// oa_iterate opens a loop without an end; its history is a ring of the last ITERATE_RING iterations
constexpr int ITERATE_OPEN = 0x7FFFFFFF, ITERATE_RING = 64;

int begin_loop(oa_ctx *c, const oa_settings *st, int iters)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (!st) return fail(OA_E_BAD_ARG, "null settings");
    if (!(st->thresh > 0.0)) return fail(OA_E_BAD_THRESH, "thresh must be > 0 (the reference's make_pairs returns None)");
    if ((rc = use_device(c))) return rc;
    if ((rc = ensure_common(c))) return rc;
    if ((rc = ensure_history(c, iters == ITERATE_OPEN ? ITERATE_RING : iters))) return rc;   // oa_iterate: a small ring
    c->settings = *st;
    c->iterate_mode = false;
    HIPCHK(hipStreamSynchronize(c->stream));                    // nothing of an earlier loop may still write the host flag
    if (c->h_poll) { for (int k = 0; k < 8; ++k) c->h_poll[k] = 0; }
    init_loop_state(c, st, iters);
    *c->h_state_pin = c->h_state;                               // pinned staging copy (the stream is idle, see above)
    HIPCHK(hipMemcpyAsync(c->d_state, c->h_state_pin, sizeof(oa::DevState), hipMemcpyHostToDevice, c->stream));
    if (c->d_todo_count) HIPCHK(hipMemsetAsync(c->d_todo_count, 0, TODO_COUNT_INTS * sizeof(int), c->stream));   // kept at zero by k_solve_update (the front search's counters: by its launches)
    c->u_slot = 0;
    if (!c->surface && (rc = safe_radii_lazy(c))) return rc;
    hipLaunchKernelGGL(oa::k_stamp_start, dim3(1), dim3(64), 0, c->stream, c->d_state);
    HIPCHK(hipGetLastError());
    c->ev_used = 0;
    c->iter_enq = 0;
    c->fast_iters = 0;
    c->h_hist_valid = false;
    {
        const bool brute = !c->surface ? !(bvh_whole(c, c->bvh_ok, vertex_tree_max(c)) || grid_active(c))
                                       : (c->grid_mode == 0 || !c->tbvh_ok);
        c->time_events = env_int("OA_TIME_EVENTS", brute ? 1 : 0) != 0;
    }
    c->loop_active = true;
    return OA_OK;
}

int iter_partial(oa_ctx *c, double *d_sums, bool timed)
{
    oa::RowSel sel;
    bool fused;
    const int rc = launch_search_accumulate(c, timed, sel, fused);
    if (rc) return rc;
    return launch_reduce(c, d_sums, sel, fused);
}

int iter_finish(oa_ctx *c, const double *d_sums)
{
    hipLaunchKernelGGL(oa::k_solve_update, dim3(1), dim3(128), 0, c->stream, c->d_state, d_sums, c->d_hist, c->d_todo_count);
    HIPCHK(hipGetLastError());
    return OA_OK;
}

// single-GPU iteration: search (+ accumulate), then reduce + solve in one launch
int iter_fused(oa_ctx *c, bool timed)
{
    oa::RowSel sel;
    bool fused;
    const int rc = launch_search_accumulate(c, timed, sel, fused);
    if (rc) return rc;
    hipLaunchKernelGGL(oa::k_reduce_solve_update, dim3(1), dim3(oa::RED_THREADS), 0, c->stream, c->d_state,
                       (const double *)c->d_partials, sel, c->d_sums, c->d_hist, c->d_todo_count, fused ? 1 : 0);
    HIPCHK(hipGetLastError());
    return OA_OK;
}

int fetch_state(oa_ctx *c)
{
    HIPCHK(hipMemcpyAsync(c->h_state_pin, c->d_state, sizeof(oa::DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->h_state = *c->h_state_pin;
    return OA_OK;
}

// host mirror -> device through the pinned staging copy (the stream must be idle: the staging copy is reused)
int push_state(oa_ctx *c)
{
    *c->h_state_pin = c->h_state;
    HIPCHK(hipMemcpyAsync(c->d_state, c->h_state_pin, sizeof(oa::DevState), hipMemcpyHostToDevice, c->stream));
    return OA_OK;
}

int fill_report(oa_ctx *c, oa_report *rep)
{
    int rc = fetch_state(c);
    if (rc) return rc;
    const oa::DevState &s = c->h_state;
    memset(rep, 0, sizeof *rep);
    if (c->h_poll && s.n > 0) c->last_todo_wave_max = c->h_poll[3];   // where the next loop's first grid search starts from (grid_fast_now)
    rep->iters_done = s.n;
    rep->converged = s.converged;
    rep->status = s.status;
    rep->mean_dist = rep->std_dist = NAN;
    c->h_hist_valid = false;
    const int m = (s.n > 0 && c->d_hist) ? std::min(s.n, c->max_records) : 0;
    if (m > 0) {                                                  // one copy serves the report, the timing and oa_get_history
        c->h_hist.resize((size_t)m);
        memcpy(c->h_hist.data(), c->h_hist_map, sizeof(oa::StepRecord) * (size_t)m);   // the stream is idle: fetch_state synchronised
        c->h_hist_valid = true;
        const oa::StepRecord &r = c->h_hist[(size_t)((s.n - 1) % c->max_records)];
        rep->last_K = (int64_t)r.K;
        rep->last_translation = r.trans;
        if (s.use_target) { rep->mean_dist = r.mean_d; rep->std_dist = r.std_d; }
        double a = 0.0;
        const int mm = std::min(s.n, 5);
        if (s.use_target) { for (int k = 0; k < mm; ++k) a += s.ring_r[k]; rep->mean_rot_angle = a / mm; }
        else rep->mean_rot_angle = r.angle;
    }
    // search time: hipEvent pairs around every launch (brute force), else the GPU-side stamps the loop left in the
    // step records (end of the previous iteration -> start of k_pair_accumulate)
    double nn_ms = 0.0;
    c->h_search_ms.clear();                                       // per launch, for oa_get_search_ms
    if (c->time_events) {
        for (int k = 0; k < c->ev_used; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]) == hipSuccess) { nn_ms += ms; c->h_search_ms.push_back((double)ms); }
        }
    } else if (m > 0) {
        double ticks = 0.0;
        for (int k = 0; k < m; ++k) { ticks += c->h_hist[(size_t)k].search_ticks; c->h_search_ms.push_back(c->h_hist[(size_t)k].search_ticks / c->wall_clock_khz); }
        nn_ms = ticks / c->wall_clock_khz * ((double)s.n / (double)m);
    }
    rep->nn_ms_total = nn_ms;
    // multi-GPU: how long the device waited for the world's sums, per iteration (StepRecord::exchange_ticks)
    c->last_exchange_us = 0.0;
    if (m > 0) {
        double ticks = 0.0;
        for (int k = 0; k < m; ++k) ticks += c->h_hist[(size_t)k].exchange_ticks;
        c->last_exchange_us = ticks / c->wall_clock_khz * 1e3 / (double)m;
    }
    return OA_OK;
}


// ------------------------------------------------------------------------------------------------
// multi-device exchange: how the children of an oa_create_multi context join their sums
// ------------------------------------------------------------------------------------------------
//   OA_EXCHANGE_RCCL     ncclAllReduce(24 doubles, sum) over xGMI on every device's stream through a single-process
//                        ncclCommInitAll communicator (librccl is loaded on first use: liboa_icp.so does not link it).
//                        What BASELINE.json's north-star names; the DEFAULT whenever the listed devices are distinct and
//                        librccl can be brought up (OA_EXCHANGE_AUTO).
//   OA_EXCHANGE_MAILBOX  one-shot all-gather through mailboxes: k_reduce_post writes a rank's 24 sums + a sequence word
//                        into every rank's inbox, k_gather_solve_update waits for the world's posts in its own inbox and
//                        adds them in rank order (bitwise identical on every device).  No host involvement, no extra
//                        launches: an iteration stays search -> accumulate -> post -> solve.  The inboxes live in device
//                        memory (fine-grained, peer-mapped: the 200 B travel over xGMI and every rank polls its own HBM);
//                        when peer access is not available, one box in pinned host memory every device maps (PCIe).
//                        The fallback of AUTO: duplicate devices (one-GPU test boxes), no librccl.
}  // namespace
struct oa_exchange {
    int world = 0;
    int requested = OA_EXCHANGE_AUTO;              // what the caller asked for (env OA_EXCHANGE / oa_set_exchange)
    int mode = OA_EXCHANGE_MAILBOX;                // what the loops use once `resolved`
    bool resolved = false;
    std::string auto_note;                         // why AUTO did not take RCCL
    // mailbox
    bool device_box = false;                       // inboxes in device memory (peer-mapped) instead of the pinned host box
    oa::MailSlot *h_box = nullptr;                 // host box [2][world]
    std::vector<oa::MailSlot *> box;               // rank r's mailbox as rank r's device sees it (its inbox / the host box)
    std::vector<oa::MailSlot **> d_dests;          // rank r: device array of the mailboxes r posts into
    int n_dest = 1;
    unsigned long long timeout_ticks = 0;
    unsigned long long seq_next = 0;               // DevState::seq_base of the next loop
    int fault_skip_rank = -1;                      // OA_FAULT_SKIP_POST_RANK: that rank's posts never arrive (test hook)
    double timeout_s = 30.0;                       // OA_EXCHANGE_TIMEOUT_S: the mailbox kernels' wait (timeout_ticks) and the host's watchdog in RCCL mode
    // How many iterations the host threads of one oa_run enqueue: agreed, not observed.  Every thread stops on what ITS device
    // reports (DevState::host_halt), and the threads hear of a halt at different times; an iteration enqueued in RCCL mode
    // contains a collective, and a collective one rank enqueues and another does not never completes.  So: a thread COMMITS
    // to iteration `it` under the lock before it enqueues it (agree_committed = most iterations any thread has committed to);
    // the first thread that hears of the halt -- or fails -- freezes agree_stop_at = agree_committed; from then on every
    // thread enqueues exactly agree_stop_at iterations, no more (it stops there) and no fewer (it tops up to it: iterations
    // after the halt are empty but for the collective).  The count is a function of the lock order alone, never of what a
    // thread happened to read from its device afterwards.  (multi_group_loop; docs/HISTORY.md 4.7 "the invariant")
    std::mutex agree_mu;
    int agree_committed = 0;
    std::atomic<int> agree_stop_at{ -1 };          // -1: not frozen yet
    bool agree_on = true;                          // OA_MULTI_AGREE=0 (test hook, ignored in RCCL mode): round 3's every-thread-for-itself stop
    int fault_lag_group = -1, fault_lag_us = 0;    // OA_FAULT_LAG_GROUP / OA_FAULT_LAG_US: that host thread sleeps before every look at its halt flag (test hook)
    // (atomics: every host thread looks at these, the one they name clears them once they have fired -- a plain int was a
    //  data race, found by the ThreadSanitizer pass of round 4, profiles/r04d_sanitizers.txt)
    std::atomic<int> fault_fail_group{ -1 };           // OA_FAULT_FAIL_GROUP / OA_FAULT_FAIL_ITER: that host thread's enqueue fails at that iteration (test hook)
    int fault_fail_iter = -1;
    std::atomic<int> fault_stall_rank{ -1 };           // OA_FAULT_STALL_RANK / _ITER: that rank's stream stops ahead of its collective, as if its peers never arrived (test hook)
    int fault_stall_iter = 2;
    int32_t *h_release = nullptr;                  // pinned word the stalled stream watches: exchange_abort_rccl releases it
    // RCCL
    void *lib = nullptr;
    std::vector<ncclComm_t> comms;
    int rccl_ranks_last = 0;                       // ranks of the last communicator that came up and passed its handshake (it may have been aborted since)
    int rccl_fallbacks = 0;                        // loops that AUTO started on RCCL and finished through the mailboxes (multi_run)
    bool rccl_aborted = false;                     // a loop ran into the watchdog and the communicators were aborted: AUTO stays on the mailbox from here on
    long long watchdog_aborts = 0;                 // OA_STAT_WATCHDOG_ABORTS
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
namespace {
using Exchange = oa_exchange;

#define RCCLCHK(x, expr)                                                                                         \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return fail(OA_E_RCCL, "%s failed: %s", #expr, (x)->GetErrorString ? (x)->GetErrorString(r_) : "?"); \
    } while (0)

bool devices_distinct(const oa_ctx *p)
{
    for (size_t i = 0; i < p->subs.size(); ++i)
        for (size_t j = 0; j < i; ++j)
            if (p->subs[i]->device == p->subs[j]->device) return false;
    return true;
}

// RCCL's communicators are given up: every collective kernel still waiting for a peer returns, the streams drain.
// (ncclCommAbort frees the communicator; the next explicit oa_set_exchange(RCCL) builds new ones.)
void exchange_abort_rccl(oa_ctx *p, const char *why)
{
    Exchange *x = p->xch;
    if (x->h_release) { __atomic_store_n(x->h_release, 1, __ATOMIC_RELEASE); }   // (test hook: the stalled stream goes on)
    if (x->comms.empty()) return;
    if (p->subs.empty() ? false : p->subs[0]->debug) fprintf(stderr, "[oa] RCCL communicators aborted: %s\n", why);
    for (size_t i = 0; i < x->comms.size(); ++i)
        if (x->comms[i] && x->CommAbort) { if (i < p->subs.size()) (void)hipSetDevice(p->subs[i]->device); (void)x->CommAbort(x->comms[i]); }
    x->comms.clear();
    x->rccl_aborted = true;
    x->watchdog_aborts++;
    x->resolved = false;                             // AUTO resolves to the mailbox from here on (exchange_resolve)
    x->auto_note = std::string("RCCL was aborted: ") + why;
}

// Wait for every child's stream, never for ever.  Mailbox: the gather kernels bound their own wait (timeout_ticks), so a
// plain synchronise returns.  RCCL: a collective whose peers never enter it spins until its communicator is aborted, so
// the host watches the streams (hipStreamQuery) and the devices' progress words; when no device has finished an
// iteration for OA_EXCHANGE_TIMEOUT_S -- or RCCL reports an asynchronous error -- it aborts the communicators
// (ncclCommAbort), gives the streams the same time again to drain, and the call fails with OA_E_RCCL.
// abort_now: the caller knows the ranks' collective counts differ (a host thread failed mid-loop): no point in waiting.
int multi_wait(oa_ctx *p, bool abort_now = false, bool in_loop = true)
{
    Exchange *x = p->xch;
    const size_t n = p->subs.size();
    const bool rccl = x && x->mode == OA_EXCHANGE_RCCL && !x->comms.empty();
    if (!rccl) {
        int rc = OA_OK;
        for (oa_ctx *c : p->subs) {
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess && !rc) rc = fail(OA_E_HIP, "device %d: %s", c->device, hipGetErrorString(e));
        }
        return rc;
    }
    using clk = std::chrono::steady_clock;
    const auto limit = std::chrono::duration<double>(x->timeout_s);
    std::vector<char> done(n, 0);
    std::vector<int> seen(n, -1);
    bool aborted = false;
    std::string why;
    if (abort_now) { exchange_abort_rccl(p, "a host thread failed while the loop was being enqueued"); aborted = true; why = "a host thread failed mid-loop"; }
    auto t_last = clk::now();
    for (;;) {
        // The clock runs while SOME stream sits in its exchange (h_poll[5] > h_poll[1]: k_reduce_partials has run, the solve behind
        // the collective has not) and nothing moves; a device that is still searching -- a 10M-point brute-force shard takes
        // seconds per iteration -- is not a stalled collective.  in_loop = false: the handshake, which is nothing but a collective.
        bool all = true, moved = false, waiting = !in_loop || aborted;
        for (size_t i = 0; i < n; ++i) {
            if (done[i]) continue;
            oa_ctx *c = p->subs[i];
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipStreamQuery(c->stream);
            if (e == hipSuccess) { done[i] = 1; moved = true; continue; }
            if (e != hipErrorNotReady) { (void)hipGetLastError(); return fail(OA_E_HIP, "device %d: %s", c->device, hipGetErrorString(e)); }
            (void)hipGetLastError();
            all = false;
            const int at = c->h_poll ? ((volatile int32_t *)c->h_poll)[1] : 0;
            if (at != seen[i]) { seen[i] = at; moved = true; }
            // (a child whose progress words are not mapped can never show "in its exchange": it counts as waiting, so that a
            //  hung collective is still aborted -- the clock then also runs through its searches, as before round 5)
            if (!c->h_poll || !c->poll_mapped || ((volatile int32_t *)c->h_poll)[5] > at) waiting = true;
            if (!aborted && x->CommGetAsyncError && i < x->comms.size() && x->comms[i]) {
                ncclResult_t ar = ncclSuccess;
                if (x->CommGetAsyncError(x->comms[i], &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    why = std::string("RCCL reported an asynchronous error on device ") + std::to_string(c->device) + ": " + (x->GetErrorString ? x->GetErrorString(ar) : "?");
                    exchange_abort_rccl(p, why.c_str());
                    aborted = true; t_last = clk::now();
                }
            }
        }
        if (all) break;
        if (moved || !waiting) t_last = clk::now();
        else if (clk::now() - t_last > limit) {
            if (aborted) return fail(OA_E_RCCL, "multi-device exchange: %s, and the streams did not drain within %.1f s of the abort", why.c_str(), x->timeout_s);
            why = "no device finished an iteration for " + std::to_string(x->timeout_s) + " s (OA_EXCHANGE_TIMEOUT_S)";
            exchange_abort_rccl(p, why.c_str());
            aborted = true; t_last = clk::now();
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (aborted) return fail(OA_E_RCCL, "multi-device exchange: %s; the RCCL communicators were aborted", why.c_str());
    return OA_OK;
}

// first use of new communicators: one all-reduce of a known value on every device, waited for with the watchdog.  A
// communicator that cannot do that (a link down, a peer mapping refused) is aborted here -- AUTO then takes the mailbox --
// instead of in the middle of somebody's alignment.
int exchange_handshake_rccl(oa_ctx *p)
{
    Exchange *x = p->xch;
    const size_t n = p->subs.size();
    std::vector<double *> bufs(n, nullptr);
    int rc = OA_OK;
    const int keep_mode = x->mode;
    for (size_t i = 0; i < n && !rc; ++i) {
        oa_ctx *c = p->subs[i];
        if ((rc = use_device(c))) break;
        double h[oa::NSUMS];
        for (int k = 0; k < oa::NSUMS; ++k) h[k] = (double)(i + 1) * (k + 1);
        hipError_t e = hipMalloc((void **)&bufs[i], sizeof h);
        if (e == hipSuccess) e = hipMemcpy(bufs[i], h, sizeof h, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(OA_E_HIP, "RCCL handshake: %s", hipGetErrorString(e));
    }
    if (!rc) {
        ncclResult_t r = x->GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) {
            oa_ctx *c = p->subs[i];
            if (hipSetDevice(c->device) != hipSuccess) { r = (ncclResult_t)1; break; }
            r = x->AllReduce(bufs[i], bufs[i], oa::NSUMS, ncclDouble, ncclSum, x->comms[i], c->stream);
        }
        const ncclResult_t r2 = x->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) rc = fail(OA_E_RCCL, "RCCL handshake: ncclAllReduce failed: %s", x->GetErrorString ? x->GetErrorString(r) : "?");
    }
    if (!rc) {
        x->mode = OA_EXCHANGE_RCCL;                                     // (multi_wait's watchdog is the RCCL one)
        rc = multi_wait(p, false, false);
        x->mode = keep_mode;
    }
    for (size_t i = 0; i < n && !rc; ++i) {
        double h[oa::NSUMS];
        if (hipSetDevice(p->subs[i]->device) != hipSuccess || hipMemcpy(h, bufs[i], sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(OA_E_HIP, "RCCL handshake: read-back failed"); break; }
        for (int k = 0; k < oa::NSUMS; ++k)
            if (h[k] != (double)(n * (n + 1) / 2) * (k + 1)) { rc = fail(OA_E_RCCL, "RCCL handshake: device %d holds a wrong sum", p->subs[i]->device); break; }
    }
    for (size_t i = 0; i < n; ++i) if (bufs[i]) { (void)hipSetDevice(p->subs[i]->device); (void)hipFree(bufs[i]); }
    if (rc && !x->comms.empty()) { const std::string keep = g_err; exchange_abort_rccl(p, "handshake failed"); g_err = keep; }
    if (!rc) { x->rccl_aborted = false; x->rccl_ranks_last = (int)n; }
    return rc;
}

int exchange_init_rccl(oa_ctx *p)
{
    Exchange *x = p->xch;
    if (!x->comms.empty()) return OA_OK;
    for (size_t i = 0; i < p->subs.size(); ++i)
        for (size_t j = 0; j < i; ++j)
            if (p->subs[i]->device == p->subs[j]->device)
                return fail(OA_E_RCCL, "OA_EXCHANGE_RCCL needs distinct devices (device %d is listed twice); use OA_EXCHANGE_MAILBOX", p->subs[i]->device);
    if (!x->lib) {
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *nm : names) if ((x->lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!x->lib) return fail(OA_E_RCCL, "librccl not found (%s)", dlerror());
        x->CommInitAll = (decltype(x->CommInitAll))dlsym(x->lib, "ncclCommInitAll");
        x->CommDestroy = (decltype(x->CommDestroy))dlsym(x->lib, "ncclCommDestroy");
        x->CommCount = (decltype(x->CommCount))dlsym(x->lib, "ncclCommCount");
        x->AllReduce = (decltype(x->AllReduce))dlsym(x->lib, "ncclAllReduce");
        x->GroupStart = (decltype(x->GroupStart))dlsym(x->lib, "ncclGroupStart");
        x->GroupEnd = (decltype(x->GroupEnd))dlsym(x->lib, "ncclGroupEnd");
        x->GetErrorString = (decltype(x->GetErrorString))dlsym(x->lib, "ncclGetErrorString");
        x->CommAbort = (decltype(x->CommAbort))dlsym(x->lib, "ncclCommAbort");
        x->CommGetAsyncError = (decltype(x->CommGetAsyncError))dlsym(x->lib, "ncclCommGetAsyncError");
        if (!x->CommInitAll || !x->CommDestroy || !x->AllReduce || !x->GroupStart || !x->GroupEnd || !x->CommAbort)
            return fail(OA_E_RCCL, "librccl lacks the expected entry points");
    }
    std::vector<int> devs;
    for (oa_ctx *c : p->subs) devs.push_back(c->device);
    x->comms.assign(devs.size(), nullptr);
    ncclResult_t r = x->CommInitAll(x->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        x->comms.clear();
        return fail(OA_E_RCCL, "ncclCommInitAll failed: %s", x->GetErrorString ? x->GetErrorString(r) : "?");
    }
    return exchange_handshake_rccl(p);
}

// how many ranks RCCL's communicator spans (0 = RCCL is not what this context exchanges through)
int exchange_rccl_ranks(const Exchange *x)
{
    if (!x || x->mode != OA_EXCHANGE_RCCL || x->comms.empty()) return 0;
    int n = 0;
    if (x->CommCount && x->CommCount(x->comms[0], &n) == ncclSuccess) return n;
    return (int)x->comms.size();
}

// AUTO -> RCCL when the devices are distinct and RCCL comes up, else the mailbox.  Decided once, before the first loop
// (ncclCommInitAll takes ~0.1-1 s: not something to pay inside oa_create_multi for contexts that only upload and pair).
int exchange_resolve(oa_ctx *p)
{
    Exchange *x = p->xch;
    if (x->resolved) return OA_OK;
    if (x->requested == OA_EXCHANGE_RCCL) {
        const int rc = exchange_init_rccl(p);
        if (rc) return rc;
        x->mode = OA_EXCHANGE_RCCL;
    } else if (x->requested == OA_EXCHANGE_MAILBOX) {
        x->mode = OA_EXCHANGE_MAILBOX;
    } else {
        x->mode = OA_EXCHANGE_MAILBOX;
        const bool any = env_int("OA_AUTO_RCCL_ANY", 0) != 0;       // test hook: AUTO takes RCCL for a world of one too (the fallback of multi_run on a one-GPU box)
        if (p->subs.size() < 2 && !any) x->auto_note = "one device: nothing to exchange";
        else if (!devices_distinct(p)) x->auto_note = "a device is listed more than once";
        else if (x->rccl_aborted) { /* auto_note says why: a communicator that hung once is not trusted again unasked */ }
        else {
            const std::string keep = g_err;
            if (exchange_init_rccl(p) == OA_OK) x->mode = OA_EXCHANGE_RCCL;
            else { x->auto_note = g_err; g_err = keep; }
        }
    }
    if (x->mode == OA_EXCHANGE_RCCL) x->agree_on = true;             // the test hook that switches the agreement off is for the mailbox only
    x->resolved = true;
    return OA_OK;
}

void exchange_destroy(oa_ctx *p)
{
    Exchange *x = p->xch;
    if (!x) return;
    for (ncclComm_t cm : x->comms) if (cm && x->CommDestroy) (void)x->CommDestroy(cm);
    for (size_t r = 0; r < p->subs.size(); ++r) {
        if (hipSetDevice(p->subs[r]->device) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (r < x->d_dests.size() && x->d_dests[r]) (void)hipFree(x->d_dests[r]);
        if (x->device_box && r < x->box.size() && x->box[r]) (void)hipFree(x->box[r]);
    }
    if (x->h_box) (void)hipHostFree(x->h_box);
    if (x->h_release) (void)hipHostFree(x->h_release);
    // the library handle stays open: unloading RCCL under a live HIP runtime is not worth the risk
    delete x;
    p->xch = nullptr;
}

// Mailboxes of a new multi-device context.  Device placement needs every pair of distinct devices to map each other's
// memory (hipDeviceEnablePeerAccess) and a fine-grained allocation per rank; anything less falls back to the pinned host
// box.  OA_MAILBOX=host / device forces one (device: fails instead of falling back).
int exchange_setup_mailboxes(oa_ctx *p)
{
    Exchange *x = p->xch;
    const int world = x->world;
    const size_t bytes = sizeof(oa::MailSlot) * 2 * (size_t)world;
    const char *want = getenv("OA_MAILBOX");
    const bool force_host = want && !strcmp(want, "host"), force_dev = want && !strcmp(want, "device");
    bool dev_ok = !force_host;
    std::string why;
    if (dev_ok) {
        for (int i = 0; i < world && dev_ok; ++i)
            for (int j = 0; j < world && dev_ok; ++j) {
                const int di = p->subs[(size_t)i]->device, dj = p->subs[(size_t)j]->device;
                if (di == dj) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) { (void)hipGetLastError(); dev_ok = false; why = "no peer access between the listed devices"; break; }
                if (hipSetDevice(di) != hipSuccess) { (void)hipGetLastError(); dev_ok = false; why = "hipSetDevice failed"; break; }
                const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { dev_ok = false; why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
                (void)hipGetLastError();
            }
    }
    x->box.assign((size_t)world, nullptr);
    x->d_dests.assign((size_t)world, nullptr);
    if (dev_ok) {
        for (int r = 0; r < world && dev_ok; ++r) {
            void *ptr = nullptr;
            hipError_t e = hipSetDevice(p->subs[(size_t)r]->device);
            if (e == hipSuccess) e = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained);
            if (e == hipSuccess) e = hipMemset(ptr, 0, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); dev_ok = false; why = std::string("fine-grained device allocation: ") + hipGetErrorString(e); if (ptr) (void)hipFree(ptr); break; }
            x->box[(size_t)r] = (oa::MailSlot *)ptr;
        }
        if (!dev_ok) for (int r = 0; r < world; ++r) if (x->box[(size_t)r]) { (void)hipSetDevice(p->subs[(size_t)r]->device); (void)hipFree(x->box[(size_t)r]); x->box[(size_t)r] = nullptr; }
    }
    if (!dev_ok && force_dev) return fail(OA_E_HIP, "OA_MAILBOX=device: %s", why.c_str());
    x->device_box = dev_ok;
    if (!dev_ok) {
        hipError_t e = hipHostMalloc((void **)&x->h_box, bytes, hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) return fail(OA_E_HIP, "oa_create_multi: mailbox allocation failed: %s", hipGetErrorString(e));
        memset(x->h_box, 0, bytes);
        for (int r = 0; r < world; ++r) {
            void *dp = nullptr;
            e = hipSetDevice(p->subs[(size_t)r]->device);
            if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, x->h_box, 0);
            if (e != hipSuccess) return fail(OA_E_HIP, "oa_create_multi: device %d cannot map the mailbox: %s", p->subs[(size_t)r]->device, hipGetErrorString(e));
            x->box[(size_t)r] = (oa::MailSlot *)dp;
        }
    }
    // where rank r posts: every rank's inbox (device placement) or the one host box as r's device sees it
    x->n_dest = dev_ok ? world : 1;
    for (int r = 0; r < world; ++r) {
        std::vector<oa::MailSlot *> dst;
        if (dev_ok) dst = x->box; else dst.push_back(x->box[(size_t)r]);
        void *dp = nullptr;
        hipError_t e = hipSetDevice(p->subs[(size_t)r]->device);
        if (e == hipSuccess) e = hipMalloc(&dp, sizeof(oa::MailSlot *) * dst.size());
        if (e == hipSuccess) e = hipMemcpy(dp, dst.data(), sizeof(oa::MailSlot *) * dst.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(OA_E_HIP, "oa_create_multi: mailbox table: %s", hipGetErrorString(e));
        x->d_dests[(size_t)r] = (oa::MailSlot **)dp;
    }
    return OA_OK;
}

// One iteration on the children of one host thread's group: pass 1 search + accumulate + post / reduce on each child,
// [all-reduce], pass 2 gather + solve on each.  (Two passes, not child by child: children that share a GPU may share a
// hardware queue, and every post of an iteration has to be submitted ahead of the gathers that wait for it.)
int multi_iteration_group(oa_ctx *p, const std::vector<int> &group, bool timed)
{
    Exchange *x = p->xch;
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i : group) {
        oa_ctx *c = p->subs[(size_t)i];
        if ((rc = use_device(c))) return rc;
        oa::RowSel sel;
        bool fused;
        if ((rc = launch_search_accumulate(c, timed, sel, fused))) return rc;
        if (x->mode == OA_EXCHANGE_RCCL) rc = launch_reduce(c, c->d_sums, sel, fused);
        else {
            hipLaunchKernelGGL(oa::k_reduce_post, dim3(1), dim3(oa::RED_THREADS), 0, c->stream, (const oa::DevState *)c->d_state,
                               (const double *)c->d_partials, sel,
                               (oa::MailSlot *const *)x->d_dests[(size_t)c->rank], x->n_dest, c->rank, x->world,
                               x->fault_skip_rank == c->rank ? 1 : 0, fused ? &c->d_state->t_acc_start : (unsigned long long *)nullptr);
            HIPCHK(hipGetLastError());
        }
        if (rc) return rc;
    }
    if (x->mode == OA_EXCHANGE_RCCL) {
        if (x->comms.empty()) return fail(OA_E_RCCL, "multi-device exchange: the RCCL communicators were aborted");
        // (test hook: this rank's stream stops here, ahead of its collective, until the watchdog releases it -- what a rank
        //  whose peers never enter the collective looks like to the host; bounded, so that a test can never hang the GPU)
        for (int i : group) {
            oa_ctx *c = p->subs[(size_t)i];
            if (c->rank == x->fault_stall_rank.load(std::memory_order_relaxed) && c->enq_iters == x->fault_stall_iter && x->h_release) {
                if ((rc = use_device(c))) return rc;
                void *dp = nullptr;
                HIPCHK(hipHostGetDevicePointer(&dp, x->h_release, 0));
                hipLaunchKernelGGL(oa::k_fault_stall, dim3(1), dim3(64), 0, c->stream, (const int32_t *)dp, (unsigned long long)(20.0 * c->wall_clock_khz * 1e3));
                HIPCHK(hipGetLastError());
                x->fault_stall_rank.store(-1, std::memory_order_relaxed);   // once
            }
        }
        // one thread, several devices: group semantics; one thread per device: plain calls (each rank's kernel waits
        // for its peers on the device, the host call only enqueues)
        const bool grouped = group.size() > 1;
        if (grouped) RCCLCHK(x, x->GroupStart());
        for (int i : group) {
            oa_ctx *c = p->subs[(size_t)i];
            if ((rc = use_device(c))) { if (grouped) (void)x->GroupEnd(); return rc; }
            ncclResult_t r = x->AllReduce(c->d_sums, c->d_sums, oa::NSUMS, ncclDouble, ncclSum, x->comms[(size_t)i], c->stream);
            if (r != ncclSuccess) { if (grouped) (void)x->GroupEnd(); return fail(OA_E_RCCL, "ncclAllReduce failed: %s", x->GetErrorString ? x->GetErrorString(r) : "?"); }
        }
        if (grouped) RCCLCHK(x, x->GroupEnd());
    }
    for (int i : group) {
        oa_ctx *c = p->subs[(size_t)i];
        if ((rc = use_device(c))) return rc;
        if (x->mode == OA_EXCHANGE_RCCL) rc = iter_finish(c, c->d_sums);
        else {
            hipLaunchKernelGGL(oa::k_gather_solve_update, dim3(1), dim3(128), 0, c->stream, c->d_state, x->box[(size_t)c->rank],
                               x->world, c->d_sums, c->d_hist, c->d_todo_count, x->timeout_ticks);
            HIPCHK(hipGetLastError());
        }
        if (rc) return rc;
    }
    const double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    for (int i : group) { p->subs[(size_t)i]->enq_ns += ns / (double)group.size(); p->subs[(size_t)i]->enq_iters++; }
    return OA_OK;
}

// f(group index) on every host thread of the context (group 0 on the caller's)
template <typename F> int multi_for_groups(oa_ctx *p, F f)
{
    if (p->groups.size() <= 1 || !p->pool) {
        for (size_t g = 0; g < p->groups.size(); ++g) { const int rc = f(g); if (rc) return rc; }
        return OA_OK;
    }
    return p->pool->run(p->groups.size(), f);
}

// end whatever loop is open: wait for the streams (bounded, see multi_wait), close the loop on every child.
// failed: a host thread gave up in the middle of enqueuing -- in RCCL mode the ranks' collective counts may differ
void multi_abort(oa_ctx *p, bool failed = false)
{
    const std::string keep = g_err;
    (void)multi_wait(p, failed && p->xch && p->xch->mode == OA_EXCHANGE_RCCL && p->loop_active);
    g_err = keep;
    for (oa_ctx *c : p->subs) c->loop_active = false;
    p->loop_active = false;
}

int multi_begin(oa_ctx *p, const oa_settings *st, int iters)
{
    Exchange *x = p->xch;
    int rc;
    if ((rc = exchange_resolve(p))) return rc;
    // sequence numbers of this loop's posts: above everything an earlier loop left in the mailboxes
    const unsigned long long seq_base = x->seq_next;
    x->seq_next += (iters == ITERATE_OPEN || iters < 0) ? (1ull << 32) : (unsigned long long)iters + 2ull;
    for (oa_ctx *c : p->subs) {
        c->h_state.seq_base = seq_base;
        if ((rc = begin_loop(c, st, iters))) { multi_abort(p); return rc; }     // synchronises the child's stream first
        if ((rc = ensure_events(c, (c->time_events && iters != ITERATE_OPEN) ? std::max(1, std::min(iters, 1 << 16)) : 1))) { multi_abort(p); return rc; }
        c->enq_ns = 0.0; c->enq_iters = 0;
    }
    for (oa_ctx *c : p->subs) {
        if ((rc = use_device(c))) { multi_abort(p); return rc; }
        HIPCHK(hipEventRecord(c->ev_loop0, c->stream));
    }
    x->agree_committed = 0;
    x->agree_stop_at.store(-1);
    if (x->h_release) __atomic_store_n(x->h_release, 0, __ATOMIC_RELEASE);
    p->settings = *st;
    p->loop_active = true;
    return OA_OK;
}

int status_error(int status)
{
    if (status == OA_E_TOO_FEW_PAIRS) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");
    if (status == OA_E_SINGULAR) return fail(OA_E_SINGULAR, "align matrix_world became singular");
    if (status == OA_E_RCCL) return fail(OA_E_RCCL, "multi-device exchange: a device's sums did not arrive in time");
    return OA_OK;
}

int multi_end(oa_ctx *p, oa_report *rep)
{
    int rc = OA_OK;
    oa_report agg{};
    for (oa_ctx *c : p->subs) {
        if ((rc = use_device(c))) break;
        if (hipEventRecord(c->ev_loop1, c->stream) != hipSuccess) { rc = fail(OA_E_HIP, "hipEventRecord failed"); break; }
    }
    if (!rc) rc = multi_wait(p);                                   // bounded in RCCL mode: a hung collective becomes OA_E_RCCL
    if (rc) { for (oa_ctx *c : p->subs) c->loop_active = false; p->loop_active = false; return rc; }
    for (size_t i = 0; i < p->subs.size(); ++i) {
        oa_ctx *c = p->subs[i];
        oa_report r{};
        int rci = use_device(c);
        if (!rci) rci = fill_report(c, &r);
        float ms = 0.f;
        if (!rci && hipEventElapsedTime(&ms, c->ev_loop0, c->ev_loop1) == hipSuccess) r.loop_ms = ms;
        c->loop_active = false;
        if (rci) { if (!rc) rc = rci; continue; }
        c->last_nn_ms = r.nn_ms_total;
        if (i == 0) agg = r;
        else {
            agg.nn_ms_total = std::max(agg.nn_ms_total, r.nn_ms_total);      // the slowest device sets the pace
            agg.loop_ms = std::max(agg.loop_ms, r.loop_ms);
            if (agg.status == 0 && r.status != 0) agg.status = r.status;
        }
    }
    p->loop_active = false;
    if (rc) return rc;
    *rep = agg;
    return status_error(agg.status);
}

// May this host thread enqueue iteration `it`?  halt_seen: its own device has reported the halt.  See Exchange::agree_mu.
bool agree_next(Exchange *x, int it, bool halt_seen)
{
    if (!x->agree_on) return !halt_seen;
    std::lock_guard<std::mutex> lk(x->agree_mu);
    int stop = x->agree_stop_at.load(std::memory_order_relaxed);
    if (stop < 0 && halt_seen) { stop = x->agree_committed; x->agree_stop_at.store(stop, std::memory_order_release); }
    if (stop >= 0) return it < stop;
    x->agree_committed = std::max(x->agree_committed, it + 1);
    return true;
}

// a host thread failed: nobody commits to anything new (the failing thread's own count is whatever it got to -- multi_run
// then aborts the communicators instead of waiting for collectives that cannot complete)
void agree_fail(Exchange *x)
{
    std::lock_guard<std::mutex> lk(x->agree_mu);
    if (x->agree_stop_at.load(std::memory_order_relaxed) < 0) x->agree_stop_at.store(x->agree_committed, std::memory_order_release);
}

// the loop of one host thread's group: enqueue its children's iterations, at most `lag` ahead of the GPU, until the
// budget is spent or the threads have agreed on where the loop ends (every device takes the same decision from the same
// sums, but the threads hear of it at different times: agree_next)
int multi_group_loop(oa_ctx *p, size_t g, const oa_settings *st)
{
    const std::vector<int> &group = p->groups[g];
    if (group.empty()) return OA_OK;
    Exchange *x = p->xch;
    oa_ctx *c0 = p->subs[(size_t)group[0]];
    // (the adaptive grid path needs recent news from the device too: then the host stays close even without early exit)
    bool poll = c0->h_poll && env_int("OA_RUN_POLL", 1) && (st->early_exit || (search_plan(c0) == PLAN_GRID && c0->grid_path == 0));
    const int lag = 2;
    volatile int32_t *progress = c0->h_poll;
    for (int it = 0; it < st->iters; ++it) {
        bool halt_seen = false;
        if (poll) {                                                     // see oa_run
            const auto t_wait = std::chrono::steady_clock::now();
            while (!progress[0] && progress[1] < it - lag && x->agree_stop_at.load(std::memory_order_acquire) < 0) {
                if (std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(5)) { poll = false; break; }   // once is enough: enqueue the rest blindly
                std::this_thread::yield();
            }
            if ((int)g == x->fault_lag_group && x->fault_lag_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(x->fault_lag_us));
            halt_seen = progress[0] != 0;
        }
        if (!agree_next(x, it, halt_seen)) break;
        int rc = OA_OK;
        if (it == x->fault_fail_iter) {
            int want = (int)g;
            if (x->fault_fail_group.compare_exchange_strong(want, -1)) rc = fail(OA_E_HIP, "injected enqueue failure (OA_FAULT_FAIL_GROUP)");   // once
        }
        if (!rc) rc = multi_iteration_group(p, group, true);
        if (rc) { agree_fail(x); return rc; }
    }
    return OA_OK;
}

int multi_run(oa_ctx *p, const oa_settings *st, oa_report *rep)
{
    for (int attempt = 0;; ++attempt) {
        int rc = multi_begin(p, st, st->iters);
        if (rc) return rc;
        const bool was_rccl = p->xch && p->xch->mode == OA_EXCHANGE_RCCL;
        rc = multi_for_groups(p, [p, st](size_t g) -> int { return multi_group_loop(p, g, st); });
        const bool enqueued = rc == OA_OK;
        if (enqueued) rc = multi_end(p, rep);                       // (waits: the watchdog's OA_E_RCCL comes out of here)
        if (!rc) return OA_OK;
        const std::string keep = g_err;
        if (!enqueued) multi_abort(p, true);
        // AUTO had picked RCCL and RCCL let the loop down (watchdog, asynchronous error: the communicators are aborted and AUTO
        // resolves to the mailboxes from here on): the caller asked for an alignment, not for RCCL -- the same loop once more,
        // through the mailboxes.  The children's host mirrors still hold the pose the loop started from (a loop's state comes
        // back to the host in multi_end only); what the failed attempt left on the devices -- seeds -- changes no result.
        // An explicit oa_set_exchange(RCCL) gets the error.  (the RCCL path has never run on distinct GPUs.)
        if (rc == OA_E_RCCL && attempt == 0 && was_rccl && p->xch->requested == OA_EXCHANGE_AUTO && p->xch->rccl_aborted) {
            p->xch->rccl_fallbacks++;
            if (!p->subs.empty() && p->subs[0]->debug) fprintf(stderr, "[oa] RCCL failed (%s): the loop runs again through the mailboxes\n", keep.c_str());
            continue;
        }
        g_err = keep;
        return rc;
    }
}
}  // namespace

// ================================================================================================
// lifetime
// ================================================================================================
OA_EXPORT const char *oa_last_error(void) { return g_err.c_str(); }
OA_EXPORT const char *oa_version(void) { return "oa_icp 0.1 (gfx950)"; }

OA_EXPORT int oa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

OA_EXPORT int oa_create(oa_ctx **out, int device)
{
    if (!out) return fail(OA_E_BAD_ARG, "oa_create: null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(OA_E_NO_DEVICE, "no HIP device available: the oa_icp engine has no CPU fallback");
    }
    if (device < 0 || device >= n) return fail(OA_E_BAD_ARG, "device %d out of range (0..%d)", device, n - 1);
    oa_ctx *c = new (std::nothrow) oa_ctx();
    if (!c) return fail(OA_E_HIP, "out of host memory");
    c->device = device;
    memset(&c->h_state, 0, sizeof c->h_state);
    memset(&c->settings, 0, sizeof c->settings);
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) { c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    if (e == hipSuccess) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(OA_E_HIP, "oa_create: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    c->R_env = env_int("OA_NN_R", 0);
    if (c->R_env != 1 && c->R_env != 2 && c->R_env != 4 && c->R_env != 8) c->R_env = 0;
    c->R = c->R_env ? c->R_env : 4;
    c->grid_lanes = env_int("OA_GRID_LANES", 0);
    c->acc_threads = env_int("OA_ACC_THREADS", 0);
    c->turns_on = env_int("OA_SEARCH_TURNS", 1);
    c->debug = getenv("OA_DEBUG") != nullptr;
    c->grid_stats = env_int("OA_GRID_STATS", 0) != 0;
    c->tri_share = env_int("OA_TRI_SHARE", 1) != 0;
    c->grid_safe = std::max(0, std::min(2, env_int("OA_GRID_SAFE", 1)));
    c->tri_ring = std::max(0, std::min(2, env_int("OA_TRI_RING", 0)));
    c->tri_ring_cap = env_double("OA_TRI_RING_CAP", 0.25);
    c->tri_split = env_int("OA_TRI_SPLIT", 1) != 0;
    c->tri_wave_wgs = env_int("OA_TRI_WAVE_WGS", 1) != 0;
    c->tri_fine = std::max(0, std::min(2, env_int("OA_TRI_FINE", 0)));   // EXPERIMENT (liboa_icp_exp.so), off: measured slower than the search it fronts, docs/HISTORY.md
    c->tri_fine_min_tris = std::max(64, env_int("OA_TRI_FINE_MIN_TRIS", 200000));
    c->tri_split_lanes = env_int("OA_TRI_SPLIT_LANES", 1) != 0;
    c->tri_canon = env_int("OA_TRI_CANON", 1) != 0;
    c->list_blocks_per_cu = std::max(1, std::min(64, env_int("OA_LIST_BLOCKS_PER_CU", 16)));
    c->tri_acc = env_int("OA_TRI_ACC", 1) != 0;
    c->turn_frac = env_double("OA_TURN_FRAC", 0.1);
    c->use_filter = env_int("OA_NN_FILTER", 1) != 0;
    c->nn_sort = env_int("OA_NN_SORT", 1) != 0;
    c->nn_home_pass = env_int("OA_NN_HOME_PASS", 1) != 0;
    c->nn_wave_order = env_int("OA_NN_WAVE_ORDER", 1) != 0;
    c->nn_persist = std::max(0, std::min(16, env_int("OA_NN_PERSIST", 4)));
    c->nn_queue_min = env_int("OA_NN_QUEUE_MIN_ITEMS", -1);
    c->nn_mfma = env_int("OA_NN_MFMA", 0);
#if !defined(OA_EXPERIMENTS)
    // the default library does not carry the experiments (oa_families.hpp): their knobs are inert here, liboa_icp_exp.so has them
    c->nn_mfma = 0; c->tri_ring = 0; c->tri_fine = 0; c->nn_sort = true; c->grid_stats = false; c->tri_share = true;
    if (c->R_env == 8) { c->R_env = 4; c->R = 4; }              // (8 points per thread: an OA_EXPERIMENTS instantiation)
#endif
    c->mfma_wps = env_int("OA_MFMA_WPS", 4);
    c->grid_mode = env_int("OA_NN_GRID", -1);
    c->fused_acc = env_int("OA_FUSED_ACC", 1);
    c->tree_acc_max = env_int("OA_TREE_ACC_MAX", 4096);
    {
        const char *gp = getenv("OA_GRID_PATH");
        c->grid_path = (gp && !strcmp(gp, "fast")) ? 1 : ((gp && !strcmp(gp, "safe")) ? 2 : 0);
    }
    dev_cache().context_created();
    *out = c;
    return OA_OK;
}

// oa_create(&ctx, devices, n_dev).  One process, one child context (and stream) per listed device; the
// same device may be listed more than once -- that is how the multi-device path is tested on a single GPU.  Children
// that share a device share one stream by default (the exchange's waits then can never sit in front of the post they
// wait for); OA_MULTI_OWN_STREAMS=1 gives each its own, so that k_gather_solve_update really spins against concurrent
// producers on a one-GPU box (the two-pass enqueue order of multi_iteration_group keeps that deadlock-free).
OA_EXPORT int oa_create_multi(oa_ctx **out, const int *devices, int n_dev)
{
    if (!out) return fail(OA_E_BAD_ARG, "oa_create_multi: null out pointer");
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > 64) return fail(OA_E_BAD_ARG, "oa_create_multi: 1..64 devices, got %d", n_dev);
    oa_ctx *p = new (std::nothrow) oa_ctx();
    if (!p) return fail(OA_E_HIP, "out of host memory");
    memset(&p->h_state, 0, sizeof p->h_state);
    memset(&p->settings, 0, sizeof p->settings);
    p->device = devices[0];
    p->world = n_dev;
    Exchange *x = new (std::nothrow) Exchange();
    if (!x) { delete p; return fail(OA_E_HIP, "out of host memory"); }
    p->xch = x;
    x->world = n_dev;
    const bool own_streams = env_int("OA_MULTI_OWN_STREAMS", 0) != 0;
    int rc = OA_OK;
    for (int i = 0; i < n_dev && !rc; ++i) {
        oa_ctx *c = nullptr;
        rc = oa_create(&c, devices[i]);
        if (rc) break;
        c->parent = p; c->rank = i; c->world = n_dev; c->xch = x;
        if (!own_streams)
            for (oa_ctx *e : p->subs) if (e->device == c->device) { c->stream = e->stream; break; }   // one stream per device
        p->subs.push_back(c);
    }
    if (!rc) rc = exchange_setup_mailboxes(p);
    if (!rc) {
        const double secs = std::max(0.05, env_double("OA_EXCHANGE_TIMEOUT_S", 30.0));
        x->timeout_ticks = (unsigned long long)(secs * p->subs[0]->wall_clock_khz * 1e3);
        x->seq_next = 0;
        x->timeout_s = secs;
        x->fault_skip_rank = env_int("OA_FAULT_SKIP_POST_RANK", -1);
        x->agree_on = env_int("OA_MULTI_AGREE", 1) != 0;
        x->fault_lag_group = env_int("OA_FAULT_LAG_GROUP", -1);
        x->fault_lag_us = env_int("OA_FAULT_LAG_US", 0);
        x->fault_fail_group.store(env_int("OA_FAULT_FAIL_GROUP", -1));
        x->fault_fail_iter = env_int("OA_FAULT_FAIL_ITER", -1);
        x->fault_stall_rank.store(env_int("OA_FAULT_STALL_RANK", -1));
        x->fault_stall_iter = env_int("OA_FAULT_STALL_ITER", 2);
        if (x->fault_stall_rank.load() >= 0) {
            if (hipHostMalloc((void **)&x->h_release, sizeof(int32_t), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); x->h_release = nullptr; }
            else *x->h_release = 0;
        }
        const char *m = getenv("OA_EXCHANGE");
        if (m && (!strcmp(m, "rccl") || !strcmp(m, "RCCL") || !strcmp(m, "1"))) x->requested = OA_EXCHANGE_RCCL;
        else if (m && (!strcmp(m, "mailbox") || !strcmp(m, "MAILBOX") || !strcmp(m, "0"))) x->requested = OA_EXCHANGE_MAILBOX;
        // host threads: one group of children per GPU (OA_MULTI_THREADS=1: per child, 0: a single group).  In the loop
        // children that share a STREAM always share a thread: enqueued from two threads, one child's gather could land
        // in front of the post it waits for.
        const int threads = env_int("OA_MULTI_THREADS", -1);
        auto deal = [&](std::vector<std::vector<int>> &groups, bool per_child) {
            for (int i = 0; i < n_dev; ++i) {
                size_t g = groups.size();
                if (threads == 0) g = 0;
                else if (!per_child)
                    for (size_t k = 0; k < groups.size(); ++k)
                        if (p->subs[(size_t)groups[k][0]]->device == devices[i]) { g = k; break; }
                if (g == groups.size()) groups.emplace_back();
                groups[g].push_back(i);
            }
        };
        // (per-child loop threads only while the children of a GPU fit its hardware queues -- HIP multiplexes streams
        //  over 4 of them -- or two children's packets can end up in one queue in the wrong order; a test-box matter:
        //  on distinct GPUs every child has a device of its own)
        int max_on_one = 0;
        for (int i = 0; i < n_dev; ++i) {
            int same = 0;
            for (int j = 0; j < n_dev; ++j) same += devices[j] == devices[i] ? 1 : 0;
            max_on_one = std::max(max_on_one, same);
        }
        deal(p->upload_groups, threads > 0);
        deal(p->groups, threads > 0 && own_streams && max_on_one <= 4);
        const size_t n_threads = std::max(p->groups.size(), p->upload_groups.size());
        if (n_threads > 1) {
            p->pool = new (std::nothrow) WorkerPool();
            if (!p->pool) rc = fail(OA_E_HIP, "out of host memory");
            else p->pool->start(n_threads - 1);
        }
    }
    if (rc) { const std::string keep = g_err; oa_destroy(p); g_err = keep; return rc; }
    *out = p;
    return OA_OK;
}

OA_EXPORT int oa_set_exchange(oa_ctx *c, int mode)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (c->subs.empty()) return fail(OA_E_STATE, "oa_set_exchange: not a multi-device context (oa_create_multi)");
    if (mode != OA_EXCHANGE_AUTO && mode != OA_EXCHANGE_MAILBOX && mode != OA_EXCHANGE_RCCL) return fail(OA_E_BAD_ARG, "exchange mode %d", mode);
    if (c->loop_active) multi_abort(c);                                 // an open oa_iterate sequence ends here
    c->xch->requested = mode;
    c->xch->resolved = false;
    return exchange_resolve(c);                                         // RCCL fails now rather than in the first iteration
}

// why the exchange is what it is: the reason AUTO did not take (or no longer takes) RCCL -- "a device is listed more than once",
// librccl's load error, "RCCL was aborted: ..." after a watchdog abort -- or "" when there is nothing to say
OA_EXPORT const char *oa_exchange_note(oa_ctx *c)
{
    static thread_local std::string note;
    note = (c && c->xch) ? c->xch->auto_note : std::string();
    return note.c_str();
}

OA_EXPORT int oa_num_devices(oa_ctx *c) { return !c ? 0 : (c->subs.empty() ? 1 : (int)c->subs.size()); }

OA_EXPORT void oa_release_cached_memory(void) { dev_cache().trim(0, 0); }

OA_EXPORT void oa_destroy(oa_ctx *c)
{
    if (!c) return;
    if (!c->subs.empty() || c->xch) {
        if (!c->parent) {                                           // a multi-device parent: workers, exchange, then the children
            for (oa_ctx *sub : c->subs) { if (hipSetDevice(sub->device) == hipSuccess) (void)hipStreamSynchronize(sub->stream); }
            delete c->pool;
            c->pool = nullptr;
            exchange_destroy(c);
            for (oa_ctx *sub : c->subs) { sub->xch = nullptr; sub->parent = nullptr; oa_destroy(sub); }
            delete c;
            return;
        }
    }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    tl_stream_known = false;                                        // the stream below is about to go away
#define OA_FREE(x) dev_free(c->x, true)
    OA_FREE(d_tgt_xyz); OA_FREE(d_tg); OA_FREE(d_tf); OA_FREE(d_tf3); OA_FREE(d_tfs); OA_FREE(d_tf3s); OA_FREE(d_tgs); OA_FREE(d_tidx); OA_FREE(d_tfm); OA_FREE(d_members); OA_FREE(d_pos); OA_FREE(d_prev); OA_FREE(d_win); OA_FREE(d_worder); OA_FREE(d_homes); OA_FREE(d_qcnt); OA_FREE(d_wsafe); OA_FREE(d_safe_by_idx); OA_FREE(d_cell_start);
    OA_FREE(d_sorted); OA_FREE(d_todo_list); OA_FREE(d_todo_count); OA_FREE(d_ulist); OA_FREE(d_src4); OA_FREE(d_keys); OA_FREE(d_state);
    OA_FREE(d_partials); OA_FREE(d_sums); OA_FREE(d_solve);
    OA_FREE(d_valid); OA_FREE(d_b); OA_FREE(d_dist); OA_FREE(d_counts); OA_FREE(d_offsets); OA_FREE(d_A); OA_FREE(d_B);
    OA_FREE(d_bvh_box); OA_FREE(d_bvh_prims); OA_FREE(d_tbvh_box); OA_FREE(d_tbvh_prims);
    OA_FREE(d_tri9); OA_FREE(d_tcell_start); OA_FREE(d_tcell_rec); OA_FREE(d_tri_ring); OA_FREE(d_tfine_table); OA_FREE(d_tfine_rec);
    OA_FREE(d_sel); OA_FREE(d_src_n); OA_FREE(d_tgt_n); OA_FREE(d_src4o); OA_FREE(d_perm);
#undef OA_FREE
    if (c->h_hist_map) (void)hipHostFree(c->h_hist_map);
    if (c->h_state_pin) (void)hipHostFree(c->h_state_pin);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    if (c->h_result) (void)hipHostFree(c->h_result);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_loop0) (void)hipEventDestroy(c->ev_loop0);
    if (c->h_poll) (void)hipHostFree(c->h_poll);
    if (c->ev_loop1) (void)hipEventDestroy(c->ev_loop1);
    if (c->own_stream) { dev_cache().stream_gone(c->own_stream); (void)hipStreamDestroy(c->own_stream); }
    delete c;
    dev_cache().context_destroyed();                                // the last context gives the cached blocks back
}

OA_EXPORT int oa_set_search_mode(oa_ctx *c, int mode)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (mode < -1 || mode > 2) return fail(OA_E_BAD_ARG, "search mode %d (use OA_SEARCH_AUTO/BRUTE/GRID/BVH)", mode);
    OA_ROUTE_ALL(c, oa_set_search_mode(sub, mode));
    const bool rebuild = (c->grid_mode == 0 && mode != 0 && c->nt > 0);
    const bool sorted_now = (mode == 0 && c->nt > 0 && !c->surface && !c->d_tfs);   // (skipped when the target was uploaded)
    c->grid_mode = mode;
    if (sorted_now) {
        int rc = use_device(c);
        if (rc) return rc;
        if ((rc = build_sorted_images(c))) return rc;
    }
    if (mode == 0 && c->nn_wave_order && !c->d_worder && c->ns_pad > 0 && c->d_src4) {   // (a source uploaded for the other modes)
        const int rc = use_device(c);
        if (rc) return rc;
        HIPCHK(dev_malloc(&c->d_worder, sizeof(unsigned short) * (size_t)c->ns_pad));
    }
    if (rebuild) {                                   // the grids were skipped when the target was uploaded
        int rc = use_device(c);
        if (rc) return rc;
        if (!c->grid_ok && (rc = build_grid(c))) return rc;
        if (!c->bvh_ok && (rc = build_bvh(c, false))) return rc;
        if (c->surface && !c->tri_grid_ok && (rc = build_tri_grid(c))) return rc;
        if (c->surface && !c->tbvh_ok && (rc = build_bvh(c, true))) return rc;
    }
    return OA_OK;
}

OA_EXPORT int oa_set_stream(oa_ctx *c, void *stream)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (!c->subs.empty())
        return stream == OA_STREAM_OWN ? OA_OK : fail(OA_E_STATE, "oa_set_stream: a multi-device context runs on its own streams (one per device)");
    if (c->parent && stream != OA_STREAM_OWN) return fail(OA_E_STATE, "oa_set_stream: child of a multi-device context");
    // the handle is used as given: NULL is HIP's legacy default stream (what torch.cuda.current_stream() is
    // unless the caller switched streams); OA_STREAM_OWN selects the context's private non-blocking stream
    c->stream = (stream == OA_STREAM_OWN) ? c->own_stream : (hipStream_t)stream;
    return OA_OK;
}

// ================================================================================================
// uploads
// ================================================================================================
namespace {
// nb x 6 bounding-box partials of a float[3n] array, on the host (the stream has been waited for on return)
int bbox_partials(oa_ctx *c, const float *d_xyz, long long n, int nb, float *out)
{
    void *dv; const void *hv;
    int rc = result_buffer(c, sizeof(float) * 6 * (size_t)nb, &dv, &hv);
    if (rc) return rc;
    if (dv) {
        hipLaunchKernelGGL(oa::k_bbox_partial, dim3(nb), dim3(256), 0, c->stream, d_xyz, (int)n, (float *)dv);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(out, hv, sizeof(float) * 6 * (size_t)nb);
        return OA_OK;
    }
    DevTmp<float> d_bb;
    HIPCHK(d_bb.alloc(6 * (size_t)nb));
    hipLaunchKernelGGL(oa::k_bbox_partial, dim3(nb), dim3(256), 0, c->stream, d_xyz, (int)n, d_bb.p);
    HIPCHK(hipGetLastError());
    return read_small(c, out, d_bb, sizeof(float) * 6 * (size_t)nb);
}

// filter image for k_nn_search_filtered: bbox centre, centred -2q / |q|^2 arrays, max |q - centre|
// k_nn_search_sorted's images: the target in the order of its coordinate along the longest axis (30-bit quantised key through
// the library's stable argsort -- the order is for speed only, any permutation gives the same answers).  Needs build_filter's
// results (bounding box, centre, axes, filter_ok); allocates, so it runs with the upload or from oa_set_search_mode, never
// inside a loop.
int build_sorted_images(oa_ctx *c)
{
    dev_free(c->d_tfs); dev_free(c->d_tf3s); dev_free(c->d_tgs); dev_free(c->d_tidx);
    if (!(c->filter_ok && c->nn_sort && c->nt >= 2)) return OA_OK;
    const double *lo = c->bb_lo, *hi = c->bb_hi;
    const int ad = c->fax[2];
    int su = 0;
    for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[su] - lo[su]) su = a;
    const int sd = ad != su ? ad : (su + 1) % 3;                   // (all extents equal: any other axis)
    c->sax[0] = su; c->sax[2] = sd; c->sax[1] = 3 - su - sd;
    const double ext = hi[su] - lo[su];
    const int blocks = (c->n_groups_pad + 255) / 256;
    DevTmp<unsigned> k_in, k_out;
    DevTmp<int> v_in, v_out;
    HIPCHK(k_in.alloc((size_t)c->nt)); HIPCHK(k_out.alloc((size_t)c->nt)); HIPCHK(v_in.alloc((size_t)c->nt)); HIPCHK(v_out.alloc((size_t)c->nt));
    hipLaunchKernelGGL(oa::k_sort_keys_axis, dim3((c->nt + 255) / 256), dim3(256), 0, c->stream, (const float *)c->d_tgt_xyz, c->nt, su, lo[su],
                       ext > 0.0 ? 1073741823.0 / ext : 0.0, k_in.p, v_in.p);
    HIPCHK(hipGetLastError());
    { const int rcs = sort_pairs30(c, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)c->nt); if (rcs) return rcs; }
    HIPCHK(dev_malloc(&c->d_tfs, sizeof(float4) * 3 * (size_t)c->n_groups_pad));
    HIPCHK(dev_malloc(&c->d_tf3s, sizeof(float4) * 2 * (size_t)c->n_groups_pad));
    HIPCHK(dev_malloc(&c->d_tgs, sizeof(float4) * 3 * (size_t)c->n_groups_pad));
    HIPCHK(dev_malloc(&c->d_tidx, sizeof(int4) * (size_t)c->n_groups_pad));
    hipLaunchKernelGGL(oa::k_pack_sorted, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const float *)c->d_tgt_xyz, c->nt, c->n_groups_pad,
                       (const int *)v_out.p, c->tc[0], c->tc[1], c->tc[2], c->sax[0], c->sax[1], c->sax[2], c->d_tfs, c->d_tf3s, c->d_tgs, c->d_tidx);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));                      // (the temporaries are released on return)
    return OA_OK;
}

int build_filter(oa_ctx *c, bool vertex_target)
{
    c->filter_ok = false;
    const int nb = 256;
    std::vector<float> bb(6 * nb);
    { int rcb = bbox_partials(c, c->d_tgt_xyz, c->nt, nb, bb.data()); if (rcb) return rcb; }
    double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    bool finite = true;
    for (int b = 0; b < nb; ++b)
        for (int a = 0; a < 3; ++a) {
            const float l = bb[6 * b + a], h = bb[6 * b + 3 + a];
            if (l != l || h != h) finite = false;
            if (l < lo[a]) lo[a] = l;
            if (h > hi[a]) hi[a] = h;
        }
    for (int a = 0; a < 3; ++a) if (!(lo[a] <= hi[a]) || !(fabs(lo[a]) < 1e18) || !(fabs(hi[a]) < 1e18)) finite = false;
    if (!finite) return OA_OK;                                   // exact kernel only
    for (int a = 0; a < 3; ++a) { c->tc[a] = (float)(0.5 * (lo[a] + hi[a])); c->bb_lo[a] = lo[a]; c->bb_hi[a] = hi[a]; }
    // the first-level score drops the axis along which the cloud is thinnest (fewest vertices share a projection)
    int ad = 0;
    for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] < hi[ad] - lo[ad]) ad = a;
    c->fax[2] = ad; c->fax[0] = (ad + 1) % 3; c->fax[1] = (ad + 2) % 3;
    if (c->fax[0] > c->fax[1]) std::swap(c->fax[0], c->fax[1]);
    const int blocks = (c->n_groups_pad + 255) / 256;
    DevTmp<double> d_mx;
#if defined(OA_EXPERIMENTS)        // the images in the caller's order are k_nn_search_filtered's (OA_NN_SORT=0); the launch below still finds qmax
    HIPCHK(dev_malloc(&c->d_tf, sizeof(float4) * 3 * (size_t)c->n_groups_pad));
    HIPCHK(dev_malloc(&c->d_tf3, sizeof(float4) * 2 * (size_t)c->n_groups_pad));
#endif
    std::vector<double> mx((size_t)blocks);
    {
        // the workgroups' maxima straight into mapped host memory when they fit (result_buffer), else through a device buffer
        void *dv; const void *hv;
        { int rcv = result_buffer(c, sizeof(double) * (size_t)blocks, &dv, &hv); if (rcv) return rcv; }
        if (!dv) HIPCHK(d_mx.alloc((size_t)blocks));
        hipLaunchKernelGGL(oa::k_pack_filter, dim3(blocks), dim3(256), 0, c->stream, c->d_tgt_xyz, c->nt, c->n_groups_pad,
                           c->tc[0], c->tc[1], c->tc[2], c->fax[0], c->fax[1], c->fax[2], c->d_tf, c->d_tf3, dv ? (double *)dv : d_mx.p);
        HIPCHK(hipGetLastError());
        if (dv) { HIPCHK(hipStreamSynchronize(c->stream)); memcpy(mx.data(), hv, sizeof(double) * (size_t)blocks); }
        else { int rcr = read_small(c, mx.data(), d_mx, sizeof(double) * (size_t)blocks); if (rcr) return rcr; }
    }
    double m = 0.0;
    for (double v : mx) if (v > m) m = v;
    c->qmax = sqrt(m) * (1.0 + 1e-6);
    c->filter_ok = (c->qmax < 1e18);
    dev_free(c->d_tfs); dev_free(c->d_tf3s); dev_free(c->d_tgs); dev_free(c->d_tidx);
    // k_nn_search_sorted's images are built for the search mode that uses them: a target uploaded for the grid / tree searches
    // (OA_SEARCH_AUTO, the default) does not pay for them; oa_set_search_mode(OA_SEARCH_BRUTE) builds them when it finds none
    // (a mesh's surface searches never launch k_nn_search_sorted: no images, no sort -- ~36 B per vertex)
    if (c->grid_mode == 0 && vertex_target) { const int rcs = build_sorted_images(c); if (rcs) return rcs; }
    dev_free(c->d_tfm);
#if defined(OA_EXPERIMENTS)
    if (c->filter_ok && c->nn_mfma) {                            // experiment: the MFMA image (32 B per target)
        int e = 0;
        if (c->qmax > 0.0) { frexp(c->qmax, &e); }               // qmax = f 2^e, f in [0.5, 1)  ->  qmax 2^-e < 1
        c->mfma_sigma = ldexp(1.0, -e);
        const int n_targets_pad = c->n_groups_pad * 4;
        HIPCHK(dev_malloc(&c->d_tfm, sizeof(oa::half8) * 2 * (size_t)n_targets_pad));
        hipLaunchKernelGGL(oa::k_pack_filter_mfma, dim3((unsigned)((n_targets_pad + 255) / 256)), dim3(256), 0, c->stream, c->d_tgt_xyz,
                           c->nt, n_targets_pad, c->tc[0], c->tc[1], c->tc[2], c->fax[0], c->fax[1], c->fax[2], c->mfma_sigma, c->d_tfm);
        HIPCHK(hipGetLastError());
    }
#endif
    return OA_OK;
}

// safe radii per target vertex (k_grid_safe_radius): a seed within its own settles the query without a scan.  One
// launch on the context's stream, no wait: the searches that use the radii are enqueued behind it.
int build_safe_radii(oa_ctx *c)
{
    if (!c->grid_ok || c->safe_ok || !c->d_safe_by_idx || c->nt < 2) return OA_OK;
    hipLaunchKernelGGL(oa::k_grid_safe_radius, dim3((c->nt + 255) / 256), dim3(256), 0, c->stream, (const float4 *)c->d_sorted, c->nt, c->gp,
                       (const int *)c->d_cell_start, c->d_safe_by_idx);
    HIPCHK(hipGetLastError());
    c->safe_ok = true;
    return OA_OK;
}

// uniform grid over the target for k_nn_search_grid (needs the finite bbox build_filter found)
int build_grid(oa_ctx *c)
{
    c->grid_ok = false;
    dev_free(c->d_cell_start); dev_free(c->d_sorted); dev_free(c->d_safe_by_idx);
    c->safe_ok = false;
    if (!c->filter_ok || c->grid_mode == 0 || c->nt < 2) return OA_OK;
    if ((long long)c->nt > oa::GRID_MAX_TARGETS) return OA_OK;       // k_nn_search_grid addresses `sorted` through 32-bit byte offsets; the tree takes over
    double ext[3], vol = 1.0, scale = 0.0;
    int nz = 0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = c->bb_hi[a] - c->bb_lo[a];
        if (ext[a] > 0.0) { vol *= ext[a]; ++nz; }
        scale = std::max(scale, std::max(fabs(c->bb_lo[a]), fabs(c->bb_hi[a])));
    }
    const double ppc = env_double("OA_GRID_PPC", 2.0);               // target vertices per cell
    double h = nz ? pow(vol * ppc / (double)c->nt, 1.0 / nz) : 1.0;
    if (!(h > 0.0) || !(h < INFINITY)) return OA_OK;
    const long long max_cells = 1ll << 24;
    static_assert((1ll << 24) <= oa::GRID_MAX_CELLS, "cell_start is addressed through 32-bit byte offsets");
    DevTmp<int2> d_cell_of;                                         // per vertex {cell, rank inside the cell}
    DevTmp<int> d_counts;
    DevTmp<long long> d_off;
    HIPCHK(d_cell_of.alloc((size_t)c->nt));
    oa::GridParams gp{};
    int n_cells = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        long long total = 1;
        for (int a = 0; a < 3; ++a) {
            long long n = ext[a] > 0.0 ? (long long)floor(ext[a] / h) + 1 : 1;
            n = std::max(1ll, std::min(n, 1024ll));
            gp.n[a] = (int)n;
            total *= n;
            gp.lo[a] = c->bb_lo[a]; gp.hi[a] = c->bb_hi[a];
        }
        if (total > max_cells) { h *= 1.3; n_cells = 0; continue; }
        // a clamped axis (1024 cells) needs a cell edge that still covers the extent
        for (int a = 0; a < 3; ++a) if (ext[a] > 0.0 && ext[a] / h >= gp.n[a]) h = std::max(h, ext[a] / (gp.n[a] - 0.5));
        gp.h = h; gp.inv_h = 1.0 / h;
        gp.r_max = std::min(env_int("OA_GRID_RMAX", 3), 3);
        gp.seeded_start = env_int("OA_GRID_SEEDED_START", 1) ? 1 : 0;
        // (256: 128 was tuned in round 2, when a hand-over was a list entry; surface-like clouds -- C5 -- keep ~1000 queries per
        //  search above 128 once the doubling for a moving pose ends, and every one of them costs the iteration its fast path)
        gp.budget = env_int("OA_GRID_BUDGET", 256);
        gp.budget_moving = (int)(gp.budget * env_double("OA_GRID_BUDGET_MOVING", 2.0));
        gp.moving_h = 0.25 * gp.h;
        gp.slack = 1e-10 * scale + 1e-300;
        gp.scale = scale;
        oa::grid_params_finish(gp);
        n_cells = (int)total;
        // [n_cells]: the scan's closing zero; behind it the occupied-cell counts of k_count_nonzero's workgroups when they do not
        // go to mapped host memory (one memset for all of it: an upload is bound by its number of operations)
        const int nz_blocks = std::min(512, (n_cells + 1023) / 1024);
        HIPCHK(d_counts.alloc((size_t)n_cells + 1 + (size_t)nz_blocks));
        HIPCHK(d_off.alloc((size_t)n_cells + 1));
        void *dv; const void *hv;
        { int rcv = result_buffer(c, sizeof(int) * (size_t)nz_blocks, &dv, &hv); if (rcv) return rcv; }
        int *d_nz = dv ? (int *)dv : d_counts.p + n_cells + 1;
        HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int) * ((size_t)n_cells + 1), c->stream));
        hipLaunchKernelGGL(oa::k_grid_count, dim3((c->nt + 255) / 256), dim3(256), 0, c->stream, c->d_tgt_xyz, c->nt, gp, d_cell_of.p, d_counts.p);
        hipLaunchKernelGGL(oa::k_count_nonzero, dim3((unsigned)nz_blocks), dim3(1024), 0, c->stream, d_counts.p, n_cells, d_nz);
        HIPCHK(hipGetLastError());
        std::vector<int> nz((size_t)nz_blocks);
        if (dv) { HIPCHK(hipStreamSynchronize(c->stream)); memcpy(nz.data(), hv, sizeof(int) * (size_t)nz_blocks); }
        else { int rcr = read_small(c, nz.data(), d_nz, sizeof(int) * (size_t)nz_blocks); if (rcr) return rcr; }
        long long occupied = 0;
        for (int v : nz) occupied += v;
        const double avg = occupied > 0 ? (double)c->nt / occupied : 0.0;
        // surfaces fill few cells: refine until occupied cells hold a handful of vertices each
        if (avg > 6.0 && total * 8 <= max_cells && attempt < 5) { h *= 0.5; continue; }
        break;
    }
    if (n_cells <= 0) return OA_OK;
    HIPCHK(dev_malloc(&c->d_cell_start, sizeof(int) * (size_t)(n_cells + 1)));
    HIPCHK(dev_malloc(&c->d_sorted, sizeof(float4) * (size_t)c->nt));
    DevTmp<char> scan_tmp;                                           // released on return, behind the wait below
    { int rcs = scan_counts(c, d_counts.p, n_cells, d_off.p, scan_tmp); if (rcs) return rcs; }
    hipLaunchKernelGGL(oa::k_grid_starts, dim3((n_cells + 256) / 256), dim3(256), 0, c->stream, d_off.p, n_cells, c->d_cell_start, (int *)nullptr);
    hipLaunchKernelGGL(oa::k_grid_scatter, dim3((c->nt + 255) / 256), dim3(256), 0, c->stream, c->d_tgt_xyz, c->nt, (const int2 *)d_cell_of.p, (const int *)c->d_cell_start, c->d_sorted);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->gp = gp; c->n_cells = n_cells; c->grid_ok = true;
    c->target_iters = 0;
    if (c->grid_safe) HIPCHK(dev_malloc(&c->d_safe_by_idx, sizeof(float) * (size_t)c->nt));
    if (c->grid_safe == 2) return build_safe_radii(c);
    return OA_OK;
}

bool grid_active(const oa_ctx *c)
{
    if (!c->grid_ok || !c->bvh_ok || !c->filter_ok || !c->use_filter || c->grid_mode == 0) return false;
    if (c->grid_mode == 2) return false;
    return true;                                                      // auto: shards of <= 32768 points took the tree already
}
// vertex_index = false: the caller (oa_set_target_mesh) searches triangles; the vertex grid and vertex tree would never
// be used (their build is ~40 % of a mesh upload)
int set_target_common(oa_ctx *c, const float *xyz, int64_t n, int on_device, bool vertex_index)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (n < 0 || n > 0x7FFF0000ll) return fail(OA_E_BAD_ARG, "target vertex count %lld out of range", (long long)n);
    if (n > 0 && !xyz) return fail(OA_E_BAD_ARG, "null target pointer");
    int rc = use_device(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loop_active = false;                                         // an open oa_iterate sequence ends with the old target
    dev_free(c->d_tgt_xyz); dev_free(c->d_tg); dev_free(c->d_tf); dev_free(c->d_tf3); dev_free(c->d_tfm);
    dev_free(c->d_tfs); dev_free(c->d_tf3s); dev_free(c->d_tgs); dev_free(c->d_tidx);
    dev_free(c->d_tri9); dev_free(c->d_tcell_start); dev_free(c->d_tcell_rec); dev_free(c->d_tri_ring);
    dev_free(c->d_tfine_table); dev_free(c->d_tfine_rec);
    c->tri_ring_ok = false; c->tri_fine_ok = false;
    dev_free(c->d_bvh_box); dev_free(c->d_bvh_prims); dev_free(c->d_tbvh_box); dev_free(c->d_tbvh_prims);
    c->surface = false; c->tri_grid_ok = false; c->n_tris = 0; c->bvh_ok = false; c->tbvh_ok = false;
    dev_free(c->d_tgt_n);
    c->normals_on = false;
    c->filter_ok = false;
    c->nt = (int)n;
    c->n_groups_pad = 0;
    c->seeded = false; c->win_seeds = false;
    // {index, safe radius} per slot: only the vertex searches read it (8 bytes per source point)
    if (!vertex_index || !c->grid_safe) dev_free(c->d_wsafe);
    else if (c->d_prev && !c->d_wsafe) HIPCHK(dev_malloc(&c->d_wsafe, sizeof(uint2) * (size_t)c->ns_pad));
    if (c->d_prev) {   // seeds index the old target
        // (the safe radii beside the winner records belong to the old target's indices too)
        hipLaunchKernelGGL(oa::k_init_slots, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, c->d_prev, c->d_win, c->d_wsafe,
                           (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr, c->ns_pad);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (n == 0) return OA_OK;
    HIPCHK(dev_malloc(&c->d_tgt_xyz, sizeof(float) * 3 * (size_t)n));
    {
        // a device pointer may live on another GPU of the process (multi-device contexts replicate the target)
        const int src_dev = on_device ? pointer_device(xyz) : -1;
        if (on_device && src_dev >= 0 && src_dev != c->device)
            HIPCHK(hipMemcpyPeerAsync(c->d_tgt_xyz, c->device, xyz, src_dev, sizeof(float) * 3 * (size_t)n, c->stream));
        else
            HIPCHK(hipMemcpyAsync(c->d_tgt_xyz, xyz, sizeof(float) * 3 * (size_t)n,
                                  on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    const long long groups = (n + 3) / 4;
    const long long tiles = (groups + oa::TILE_GROUPS - 1) / oa::TILE_GROUPS;
    c->n_groups_pad = (int)(tiles * oa::TILE_GROUPS);
    HIPCHK(dev_malloc(&c->d_tg, sizeof(float4) * 3 * (size_t)c->n_groups_pad));
    hipLaunchKernelGGL(oa::k_pack_target, dim3((c->n_groups_pad + 255) / 256), dim3(256), 0, c->stream, c->d_tgt_xyz,
                       c->nt, c->n_groups_pad, c->d_tg);
    HIPCHK(hipGetLastError());
    int rcf = build_filter(c, vertex_index);                                      // (waits for the stream when it reads the bounding box: the copy above is done by then)
    if (rcf) return rcf;
    if (vertex_index) {
        if ((rcf = build_grid(c))) return rcf;
        if ((rcf = build_bvh(c, false))) return rcf;
    }
    plan_geometry(c);
    return OA_OK;
}
}  // namespace
OA_EXPORT int oa_set_target(oa_ctx *c, const float *xyz, int64_t n, int on_device)
{
    if (c && !c->subs.empty() && c->loop_active) multi_abort(c);            // a new target ends an open oa_iterate sequence
    OA_ROUTE_ALL_PAR(c, oa_set_target(sub, xyz, n, on_device));             // replicated on every device
    return set_target_common(c, xyz, n, on_device, true);
}

namespace {
// offsets[0..n] = exclusive prefix sums of counts[0..n-1] (offsets[n] = total).
// Enqueued, not waited for: `tmp` (the scan's temporary storage) belongs to the caller, who keeps it until the stream has
// been synchronised (the wait in here was a 20-50 us hole in every grid build).
int scan_counts(oa_ctx *c, const int *d_counts, int n, long long *d_off, DevTmp<char> &tmp)
{
    HIPCHK(tmp.alloc(oa::scan_tmp_bytes((size_t)n)));
    HIPCHK(oa::scan_counts_ll((void *)tmp.p, d_counts, (size_t)n, d_off, c->stream));
    return OA_OK;
}

// 64-ary bounding-box tree over the Morton-sorted vertices (tri = false) or triangles (tri = true), oa_bvh.hpp
int build_bvh(oa_ctx *c, bool tri)
{
    bool &ok = tri ? c->tbvh_ok : c->bvh_ok;
    float4 *&d_box = tri ? c->d_tbvh_box : c->d_bvh_box;
    float4 *&d_prims = tri ? c->d_tbvh_prims : c->d_bvh_prims;
    oa::BvhParams &bp = tri ? c->tbvh : c->bvh;
    ok = false;
    dev_free(d_box); dev_free(d_prims);
    const int n = tri ? c->n_tris : c->nt;
    if (!c->filter_ok || c->grid_mode == 0 || n < 1) return OA_OK;
    bp = oa::BvhParams{};
    bp.n_prims = n;
    long long total = 0;
    int level = 0;
    for (long long cnt = ((long long)n + oa::BVH_W - 1) / oa::BVH_W;; cnt = (cnt + oa::BVH_W - 1) / oa::BVH_W) {
        ++level;
        if (level > oa::BVH_MAX_LEVELS) return OA_OK;
        bp.cnt[level] = (int)cnt;
        bp.off[level] = (int)total;
        total += ((cnt + oa::BVH_W - 1) / oa::BVH_W) * oa::BVH_W;
        if (cnt <= oa::BVH_W) break;
    }
    bp.levels = level;
    double scale = 0.0;
    float lo[3], sc[3];
    for (int a = 0; a < 3; ++a) {
        scale = std::max(scale, std::max(fabs(c->bb_lo[a]), fabs(c->bb_hi[a])));
        lo[a] = (float)c->bb_lo[a];
        const double ext = c->bb_hi[a] - c->bb_lo[a];
        sc[a] = ext > 0.0 ? (float)(1023.0 / ext) : 0.f;
    }
    bp.scale = scale;
    bp.slack = 1e-10 * scale + 1e-300;
    const int n_pad = bp.cnt[1] * oa::BVH_W;
    DevTmp<unsigned> k_in, k_out;
    DevTmp<int> v_in, v_out;
    HIPCHK(k_in.alloc((size_t)n)); HIPCHK(k_out.alloc((size_t)n)); HIPCHK(v_in.alloc((size_t)n)); HIPCHK(v_out.alloc((size_t)n));
    const dim3 blk(256), grd((unsigned)((n + 255) / 256));
    if (tri) hipLaunchKernelGGL(oa::k_bvh_keys<true>, grd, blk, 0, c->stream, (const float *)c->d_tgt_xyz, (const float4 *)c->d_tri9,
                                n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], k_in.p, v_in.p);
    else hipLaunchKernelGGL(oa::k_bvh_keys<false>, grd, blk, 0, c->stream, (const float *)c->d_tgt_xyz, (const float4 *)nullptr,
                            n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], k_in.p, v_in.p);
    HIPCHK(hipGetLastError());
    { const int rcs = sort_pairs30(c, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)n); if (rcs) return rcs; }
    HIPCHK(dev_malloc(&d_prims, sizeof(float4) * (tri ? 3 : 1) * (size_t)n_pad));
    HIPCHK(dev_malloc(&d_box, sizeof(float4) * 2 * (size_t)total));
    const dim3 grd_pad((unsigned)((n_pad + 255) / 256));
    if (tri) hipLaunchKernelGGL(oa::k_bvh_gather<true>, grd_pad, blk, 0, c->stream, (const float *)c->d_tgt_xyz, (const float4 *)c->d_tri9,
                                (const int *)v_out.p, n, n_pad, d_prims);
    else hipLaunchKernelGGL(oa::k_bvh_gather<false>, grd_pad, blk, 0, c->stream, (const float *)c->d_tgt_xyz, (const float4 *)nullptr,
                            (const int *)v_out.p, n, n_pad, d_prims);
    HIPCHK(hipGetLastError());
    const dim3 grd1((unsigned)((bp.cnt[1] + 3) / 4));               // one wave per box
    if (tri) hipLaunchKernelGGL(oa::k_bvh_leaf_boxes<true>, grd1, blk, 0, c->stream, (const float4 *)d_prims, n_pad, bp.cnt[1], d_box + 2ll * bp.off[1]);
    else hipLaunchKernelGGL(oa::k_bvh_leaf_boxes<false>, grd1, blk, 0, c->stream, (const float4 *)d_prims, n_pad, bp.cnt[1], d_box + 2ll * bp.off[1]);
    HIPCHK(hipGetLastError());
    for (int l = 2; l <= bp.levels; ++l) {
        hipLaunchKernelGGL(oa::k_bvh_upper_boxes, dim3((unsigned)((bp.cnt[l] + 3) / 4)), blk, 0, c->stream,
                           (const float4 *)(d_box + 2ll * bp.off[l - 1]), bp.cnt[l - 1], bp.cnt[l], d_box + 2ll * bp.off[l]);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    ok = true;
    return OA_OK;
}

// Spatial (Morton) order of the source slots: the points a wave owns are neighbours in space, so the grid search's
// reads are shared instead of scattered and the brute-force filter's slow path fires for fewer waves.  The caller-
// order copy and the permutation stay around for the entry points that return per-point data in vlist order.
// bounding box of a float[3n] array -> Morton scaling (lo, 1023 / extent); ok = false for non-finite coordinates
int morton_frame(oa_ctx *c, const float *d_xyz, long long n_verts, float lo[3], float sc[3], bool &ok)
{
    ok = false;
    const int nb = 256;
    std::vector<float> bb(6 * nb);
    { int rcb = bbox_partials(c, d_xyz, n_verts, nb, bb.data()); if (rcb) return rcb; }
    float hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    lo[0] = lo[1] = lo[2] = INFINITY;
    for (int b = 0; b < nb; ++b)
        for (int a = 0; a < 3; ++a) {
            const float l = bb[6 * b + a], h = bb[6 * b + 3 + a];
            if (l != l || h != h) return OA_OK;                     // non-finite coordinates: keep the caller's order
            lo[a] = std::min(lo[a], l); hi[a] = std::max(hi[a], h);
        }
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] <= hi[a]) || !(fabsf(lo[a]) < 1e18f) || !(fabsf(hi[a]) < 1e18f)) return OA_OK;
        sc[a] = hi[a] > lo[a] ? 1023.0f / (hi[a] - lo[a]) : 0.f;
    }
    ok = true;
    return OA_OK;
}

// Positions (in the selection) of the points shard [begin, begin + count) of the Morton-ordered selection holds, in
// ascending order -> members (device).  Left empty (caller falls back to contiguous ranges) for non-finite input.
int spatial_shard_members(oa_ctx *c, const float *d_xyz, long long n_verts, const long long *d_vlist, long long step,
                          long long n_sel, long long begin, int count, DevTmp<int> &members)
{
    float lo[3], sc[3];
    bool ok = false;
    int rc = morton_frame(c, d_xyz, n_verts, lo, sc, ok);
    if (rc || !ok) return rc;
    const int n = (int)n_sel;
    DevTmp<float4> all4;
    DevTmp<int> all_sel, v_in, order, picked;
    DevTmp<unsigned> k_in, k_out;
    HIPCHK(all4.alloc((size_t)n)); HIPCHK(all_sel.alloc((size_t)n)); HIPCHK(v_in.alloc((size_t)n)); HIPCHK(order.alloc((size_t)n));
    HIPCHK(k_in.alloc((size_t)n)); HIPCHK(k_out.alloc((size_t)n));
    hipLaunchKernelGGL(oa::k_pack_source, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_xyz, d_vlist, step, 0ll,
                       (const int *)nullptr, n, n, all4.p, all_sel.p);
    hipLaunchKernelGGL(oa::k_morton_keys, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float4 *)all4.p, n,
                       lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], k_in.p, v_in.p);
    HIPCHK(hipGetLastError());
    { const int rcs = sort_pairs30(c, k_in.p, k_out.p, v_in.p, order.p, (size_t)n); if (rcs) return rcs; }
    // this shard's range of the order, back in ascending selection position (= the caller's order inside the shard)
    HIPCHK(members.alloc((size_t)count));
    DevTmp<int> ord2;
    HIPCHK(ord2.alloc((size_t)count));
    { const int rcs = sort_ints(c, order.p + begin, members.p, ord2.p, (size_t)count, bits_for(n_sel)); if (rcs) return rcs; }
    HIPCHK(hipStreamSynchronize(c->stream));                      // temporaries are released on return
    return OA_OK;
}

int sort_source_slots(oa_ctx *c, const float *d_xyz, long long n_verts)
{
    float lo[3], sc[3];
    bool ok = false;
    int rcf = morton_frame(c, d_xyz, n_verts, lo, sc, ok);
    if (rcf || !ok) return rcf;
    DevTmp<unsigned> k_in, k_out;
    DevTmp<int> v_in;
    HIPCHK(k_in.alloc((size_t)c->ns)); HIPCHK(k_out.alloc((size_t)c->ns)); HIPCHK(v_in.alloc((size_t)c->ns));
    HIPCHK(dev_malloc(&c->d_perm, sizeof(int) * (size_t)c->ns));
    hipLaunchKernelGGL(oa::k_morton_keys, dim3((c->ns + 255) / 256), dim3(256), 0, c->stream, (const float4 *)c->d_src4, c->ns,
                       lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], k_in.p, v_in.p);
    HIPCHK(hipGetLastError());
    { const int rcs = sort_pairs30(c, k_in.p, k_out.p, v_in.p, c->d_perm, (size_t)c->ns); if (rcs) return rcs; }
    // d_src4 / d_sel become the sorted images: the packed buffers change names (they ARE the caller-order copy now) and
    // k_apply_perm fills new ones -- every slot up to ns_pad -- instead of two device-to-device copies in front of it
    // (both new buffers first, the context's pointers only when both exist: a failed allocation leaves the unsorted source in
    //  place -- d_perm without sorted images would send later launches through null pointers)
    float4 *src_sorted = nullptr;
    int *sel_sorted = nullptr;
    {
        const hipError_t e1 = dev_malloc(&src_sorted, sizeof(float4) * (size_t)c->ns_pad);
        const hipError_t e2 = e1 == hipSuccess ? dev_malloc(&sel_sorted, sizeof(int) * (size_t)c->ns_pad) : e1;
        if (e2 != hipSuccess) { dev_free(src_sorted); dev_free(sel_sorted); dev_free(c->d_perm); return fail(OA_E_HIP, "sort_source_slots: %s", hipGetErrorString(e2)); }
    }
    dev_free(c->d_src4o);
    c->d_src4o = c->d_src4; c->d_src4 = src_sorted;
    int *selo = c->d_sel; c->d_sel = sel_sorted;
    hipLaunchKernelGGL(oa::k_apply_perm, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, (const float4 *)c->d_src4o,
                       (const int *)selo, (const int *)c->d_perm, c->ns, c->ns_pad, c->d_src4, c->d_sel);
    const hipError_t el = hipGetLastError();
    dev_free(selo);                                                // (stream ordered: behind the launch above; freed on every path)
    if (el != hipSuccess) return fail(OA_E_HIP, "k_apply_perm: %s", hipGetErrorString(el));
    HIPCHK(hipStreamSynchronize(c->stream));
    return OA_OK;
}

// uniform grid over the triangles' bounding boxes
// diag_sum_known: the sum of the triangles' bounding-box diagonals when the caller has it already (oa_set_target_mesh reads
// it back together with its index check: one host round trip instead of two)
int build_tri_grid(oa_ctx *c, const double *diag_sum_known)
{
    c->tri_grid_ok = false;
    c->tri_ring_ok = false; c->tri_iters = 0;
    c->tri_fine_ok = false;
    dev_free(c->d_tcell_start); dev_free(c->d_tcell_rec); dev_free(c->d_tri_ring);
    dev_free(c->d_tfine_table); dev_free(c->d_tfine_rec);
    if (!c->filter_ok || c->grid_mode == 0 || c->n_tris < 64) return OA_OK;
    if ((long long)c->n_tris > (long long)oa::TRI_REC_INDEX_MASK || (long long)c->n_tris > (long long)oa::TRI_POOL_MAX_TRIS) return OA_OK;     // (28-bit indices in the records, 26 in the pool entries: the tree takes such meshes)
    double diag_sum = 0.0;
    if (diag_sum_known) diag_sum = *diag_sum_known;
    else {
        constexpr size_t SLOT_WORDS = (size_t)oa::TOTAL_SLOTS * oa::TOTAL_STRIDE;
        DevTmp<double> d_sum;
        HIPCHK(d_sum.alloc(SLOT_WORDS));
        HIPCHK(hipMemsetAsync(d_sum, 0, sizeof(double) * SLOT_WORDS, c->stream));
        hipLaunchKernelGGL(oa::k_tri_diag_sum, dim3((c->n_tris + 255) / 256), dim3(256), 0, c->stream, c->d_tri9, c->n_tris, d_sum.p);
        HIPCHK(hipGetLastError());
        double slots[SLOT_WORDS];
        { int rcr = read_small(c, slots, d_sum, sizeof(slots)); if (rcr) return rcr; }
        for (int k = 0; k < oa::TOTAL_SLOTS; ++k) diag_sum += slots[(size_t)k * oa::TOTAL_STRIDE];
    }
    double ext[3], scale = 0.0, max_ext = 0.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = c->bb_hi[a] - c->bb_lo[a];
        max_ext = std::max(max_ext, ext[a]);
        scale = std::max(scale, std::max(fabs(c->bb_lo[a]), fabs(c->bb_hi[a])));
    }
    c->tri_mean_diag = diag_sum / (double)c->n_tris;
    double h = env_double("OA_TRI_CELL", 1.25) * diag_sum / (double)c->n_tris;  // mean triangle bbox diagonals per cell edge (1.5 until the scan
                                                                                 // became VALU-bound at the end of round 3: fewer records per cell now pay)
    if (!(h > 0.0) || !(h < INFINITY)) h = max_ext > 0.0 ? max_ext / 64.0 : 1.0;
    h = std::max(h, max_ext / 1023.0);
    const long long max_cells = 1ll << std::max(16, std::min(29, env_int("OA_TRI_MAX_CELLS_LOG2", 24)));
    DevTmp<int> d_counts;
    DevTmp<long long> d_off;
    oa::GridParams gp{};
    int n_cells = 0;
    unsigned long long entries = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        long long total = 1;
        for (int a = 0; a < 3; ++a) {
            long long n = ext[a] > 0.0 ? (long long)floor(ext[a] / h) + 1 : 1;
            n = std::max(1ll, std::min(n, 1024ll));
            gp.n[a] = (int)n;
            total *= n;
            gp.lo[a] = c->bb_lo[a]; gp.hi[a] = c->bb_hi[a];
        }
        if (total > max_cells) { h *= 1.3; n_cells = 0; continue; }
        gp.h = h; gp.inv_h = 1.0 / h;
        gp.r_max = std::min(env_int("OA_GRID_RMAX", 3), 3);
        gp.seeded_start = env_int("OA_TRI_SEEDED_START", 1) ? 1 : 0;
        gp.budget = std::max(1, std::min(env_int("OA_GRID_BUDGET", 192), 30000));
        gp.budget_moving = std::max(1, std::min((int)(gp.budget * env_double("OA_GRID_BUDGET_MOVING", 3.0)), 60000));   // (range lengths are 16-bit in the kernel; 2.0 until the scan was shared by the wave)
        gp.scale = scale;
        gp.slack = 1e-10 * scale + 1e-300;
        oa::grid_params_finish(gp);
        gp.eps_plane = (float)(8.0 * 5.9604644775390625e-08 * scale + 1e-37);
        gp.drop_over = env_int("OA_TRI_DROP_OVER", 1);
        gp.moving_h = env_double("OA_TRI_MOVING_FRAC", 0.25) * gp.h;
        gp.xcd_chunk = std::max(0, std::min(4096, env_int("OA_TRI_XCD_CHUNK", 8)));
        gp.xcd_moving_h = env_double("OA_TRI_XCD_MOVING_FRAC", 0.5) * gp.h;
        n_cells = (int)total;
        // (the entry total behind the counts, 8-byte aligned: one memset for both)
        const size_t total_at = ((size_t)n_cells + 2) & ~(size_t)1;
        constexpr size_t SLOT_WORDS = (size_t)oa::TOTAL_SLOTS * oa::TOTAL_STRIDE;       // 8-byte words of the spread total (oa_tri.hpp)
        HIPCHK(d_counts.alloc(total_at + 2 * SLOT_WORDS));
        unsigned long long *d_total = (unsigned long long *)(d_counts.p + total_at);
        HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int) * (total_at + 2 * SLOT_WORDS), c->stream));
        hipLaunchKernelGGL(oa::k_tri_grid_bin<false>, dim3((c->n_tris + 255) / 256), dim3(256), 0, c->stream, c->d_tri9,
                           c->n_tris, gp, d_counts.p, (const int *)nullptr, (float4 *)nullptr, d_total);
        HIPCHK(hipGetLastError());
        {
            unsigned long long slots[SLOT_WORDS];
            int rcr = read_small(c, slots, d_total, sizeof(slots));
            if (rcr) return rcr;
            entries = 0;
            for (int k = 0; k < oa::TOTAL_SLOTS; ++k) entries += slots[(size_t)k * oa::TOTAL_STRIDE];
        }
        // triangles much larger than a cell explode the lists: coarsen
        if (entries > 32ull * (unsigned long long)c->n_tris + (1ull << 20) || entries > (unsigned long long)oa::TRI_REC_MAX_ENTRIES) { h *= 2.0; n_cells = 0; continue; }
        break;
    }
    if (n_cells <= 0 || entries == 0) return OA_OK;
    HIPCHK(d_off.alloc((size_t)n_cells + 1));
    HIPCHK(dev_malloc(&c->d_tcell_start, sizeof(int) * (size_t)(n_cells + 1)));
    HIPCHK(dev_malloc(&c->d_tcell_rec, sizeof(float4) * 2 * ((size_t)entries + oa::TRI_REC_PAD)));
    HIPCHK(hipMemsetAsync(c->d_tcell_rec + 2 * (size_t)entries, 0, sizeof(float4) * 2 * oa::TRI_REC_PAD, c->stream));   // triangle 0, see tri_scan_shared
    DevTmp<char> scan_tmp;                                           // released on return, behind the wait below
    { int rcs = scan_counts(c, d_counts.p, n_cells, d_off.p, scan_tmp); if (rcs) return rcs; }
    hipLaunchKernelGGL(oa::k_grid_starts, dim3((n_cells + 256) / 256), dim3(256), 0, c->stream, d_off.p, n_cells, c->d_tcell_start, d_counts.p);
    hipLaunchKernelGGL(oa::k_tri_grid_bin<true>, dim3((c->n_tris + 255) / 256), dim3(256), 0, c->stream, c->d_tri9, c->n_tris,
                       gp, d_counts.p, (const int *)c->d_tcell_start, c->d_tcell_rec, (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tgp = gp;
    c->n_tri_entries = (long long)entries;
    c->tri_grid_ok = true;
    if (c->debug)
        fprintf(stderr, "[oa] tri grid: h=%g cells=%dx%dx%d entries=%llu (%.2f per triangle)\n", gp.h, gp.n[0], gp.n[1], gp.n[2],
                entries, (double)entries / c->n_tris);
    if (c->tri_ring) {
        HIPCHK(dev_malloc(&c->d_tri_ring, sizeof(int) * (size_t)oa::TRI_RING_STRIDE * (size_t)c->n_tris));
        if (c->tri_ring == 2) { const int rcr = build_tri_ring(c); if (rcr) return rcr; }
    }
    if (c->tri_fine == 2 || (c->tri_fine == 1 && c->n_tris >= c->tri_fine_min_tris)) return build_tri_fine(c);
    return OA_OK;
}

// The settled-pose structure (oa_tri_fine.hpp): sparse fine grid, inflated lists of whole triangles.  Built with the mesh (never
// inside a loop: it allocates).  Gives up quietly -- the general search serves everything -- when the lists would explode
// (triangles much larger than a cell after six coarsenings) or take more than OA_TRI_FINE_MAX_MB.
int build_tri_fine(oa_ctx *c)
{
    c->tri_fine_ok = false;
    dev_free(c->d_tfine_table); dev_free(c->d_tfine_rec);
#if !defined(OA_EXPERIMENTS)
    return OA_OK;                                                   // (OA_TRI_FINE is inert in the default library)
#else
    if (!c->tri_grid_ok || !c->tbvh_ok || !(c->tri_mean_diag > 0.0) || !(c->tri_mean_diag < INFINITY)) return OA_OK;
    oa::FineParams fp{};
    double h = env_double("OA_TRI_FINE_CELL", 0.5) * c->tri_mean_diag;
    const double rho_frac = std::max(0.01, std::min(env_double("OA_TRI_FINE_RHO", 0.3), 4.0));
    double ext[3], scale = 0.0, max_ext = 0.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = c->bb_hi[a] - c->bb_lo[a];
        max_ext = std::max(max_ext, ext[a]);
        scale = std::max(scale, std::max(fabs(c->bb_lo[a]), fabs(c->bb_hi[a])));
    }
    const unsigned long long max_entries = (unsigned long long)(std::max(16.0, env_double("OA_TRI_FINE_MAX_MB", 8192.0)) * 1048576.0 / 48.0);
    constexpr size_t SLOT_WORDS = (size_t)oa::TOTAL_SLOTS * oa::TOTAL_STRIDE;
    DevTmp<unsigned long long> d_total;
    HIPCHK(d_total.alloc(SLOT_WORDS));
    unsigned long long entries = 0;
    bool sized = false;
    for (int attempt = 0; attempt < 7 && !sized; ++attempt) {
        fp.rho = rho_frac * h;
        h = std::max(h, (max_ext + 2.0 * fp.rho * (1.0 + 1e-6)) / 1023.0);       // <= 1024 cells per axis: a cell id has 30 bits
        fp.rho = rho_frac * h;
        fp.rho_query = fp.rho * (1.0 - 1e-6);
        fp.h = h; fp.inv_h = 1.0 / h;
        for (int a = 0; a < 3; ++a) {
            fp.lo[a] = c->bb_lo[a] - fp.rho * (1.0 + 1e-6) - 1e-300;
            fp.hi[a] = c->bb_hi[a] + fp.rho * (1.0 + 1e-6) + 1e-300;
            const long long n = (long long)floor((fp.hi[a] - fp.lo[a]) / h) + 1;
            fp.n[a] = (int)std::max(1ll, std::min(n, 1024ll));
        }
        fp.scale = scale;
        fp.slack = 1e-10 * scale + 1e-300;
        HIPCHK(hipMemsetAsync(d_total, 0, sizeof(unsigned long long) * SLOT_WORDS, c->stream));
        hipLaunchKernelGGL(oa::k_tfine_total, dim3((unsigned)((c->n_tris + 255) / 256)), dim3(256), 0, c->stream, (const float4 *)c->d_tri9, c->n_tris, fp, d_total.p);
        HIPCHK(hipGetLastError());
        unsigned long long slots[SLOT_WORDS];
        { int rcr = read_small(c, slots, d_total, sizeof(slots)); if (rcr) return rcr; }
        entries = 0;
        for (int k = 0; k < oa::TOTAL_SLOTS; ++k) entries += slots[(size_t)k * oa::TOTAL_STRIDE];
        if (entries == 0) return OA_OK;
        if (entries > 40ull * (unsigned long long)c->n_tris + (1ull << 16) || entries > max_entries || entries >= (1ull << 31)) { h *= 1.5; continue; }
        sized = true;
    }
    if (!sized) return OA_OK;
    fp.cap = std::max(1, std::min(env_int("OA_TRI_FINE_CAP", 192), 1 << 20));
    fp.gate = env_double("OA_TRI_FINE_GATE", 1e30) * fp.rho;
    // the table: a power of two of slots, at most a third full for ordinary meshes (7-10 entries per occupied cell); a mesh whose
    // cells hold one entry each overflows the probe limit instead -- then four times the slots, twice
    unsigned long long want = std::max(1024ull, entries / 3ull);
    for (int attempt = 0; attempt < 3; ++attempt, want *= 4ull) {
        int log2 = 10;
        while ((1ull << log2) < want && log2 < 30) ++log2;
        const size_t n_slots = (size_t)1 << log2;
        fp.slots_mask = (unsigned)(n_slots - 1);
        fp.hash_shift = 32 - log2;
        DevTmp<int> d_counts;                                        // [n_slots] counts, then the overflow flag
        DevTmp<long long> d_off;
        DevTmp<char> scan_tmp;
        HIPCHK(dev_malloc(&c->d_tfine_table, sizeof(uint4) * n_slots));
        HIPCHK(dev_malloc(&c->d_tfine_rec, sizeof(float4) * 3 * (size_t)entries));
        HIPCHK(d_counts.alloc(n_slots + 2));
        HIPCHK(d_off.alloc(n_slots + 1));
        HIPCHK(hipMemsetAsync(c->d_tfine_table, 0xFF, sizeof(uint4) * n_slots, c->stream));
        HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int) * (n_slots + 2), c->stream));
        const dim3 tb((unsigned)((c->n_tris + 255) / 256));
        hipLaunchKernelGGL(oa::k_tfine_count, tb, dim3(256), 0, c->stream, (const float4 *)c->d_tri9, c->n_tris, fp, c->d_tfine_table, d_counts.p, d_counts.p + n_slots + 1);
        HIPCHK(hipGetLastError());
        { int rcs = scan_counts(c, d_counts.p, (int)n_slots, d_off.p, scan_tmp); if (rcs) return rcs; }
        hipLaunchKernelGGL(oa::k_tfine_fill, tb, dim3(256), 0, c->stream, (const float4 *)c->d_tri9, c->n_tris, fp, c->d_tfine_table, d_counts.p,
                           (const long long *)d_off.p, c->d_tfine_rec);
        hipLaunchKernelGGL(oa::k_tfine_finish, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, c->stream, c->d_tfine_table, (int)n_slots,
                           (const long long *)d_off.p);
        HIPCHK(hipGetLastError());
        int overflow = 0;
        { int rcr = read_small(c, &overflow, d_counts.p + n_slots + 1, sizeof(int)); if (rcr) return rcr; }   // (waits for the stream: the temporaries may go)
        if (!overflow) {
            c->tfp = fp;
            c->n_fine_entries = (long long)entries;
            c->tri_fine_ok = true;
            if (c->debug)
                fprintf(stderr, "[oa] tri fine grid: h=%g (%.2f mean diagonals) rho=%g cells=%dx%dx%d entries=%llu (%.2f per triangle, %.0f MB) table 2^%d slots\n",
                        fp.h, fp.h / c->tri_mean_diag, fp.rho, fp.n[0], fp.n[1], fp.n[2], entries, (double)entries / c->n_tris, entries * 48.0 / 1048576.0, log2);
            return OA_OK;
        }
        dev_free(c->d_tfine_table); dev_free(c->d_tfine_rec);
    }
    return OA_OK;
#endif
}

// neighbour lists + accept radii (oa_tri_ring.hpp): one launch on the context's stream, no wait.  The accept radii are written
// into d_tri9's spare lane, which k_pack_tris left at 0 = "never".
int build_tri_ring(oa_ctx *c)
{
#if !defined(OA_EXPERIMENTS)
    (void)c;
    return OA_OK;                                                   // (OA_TRI_RING is inert in the default library)
#else
    if (!c->tri_grid_ok || !c->d_tri_ring || c->tri_ring_ok) return OA_OK;
    const double cap = std::max(0.01, std::min(c->tri_ring_cap, 1.0)) * c->tgp.h;
    const dim3 blocks((unsigned)((c->n_tris + 255) / 256));
    if (c->grid_stats || c->debug) {
        DevTmp<unsigned long long> d_stats;
        HIPCHK(d_stats.alloc(oa::RING_STAT_N));
        HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * oa::RING_STAT_N, c->stream));
        hipLaunchKernelGGL(oa::k_tri_ring_build<true>, blocks, dim3(256), 0, c->stream, c->d_tri9, c->n_tris, c->tgp, (const int *)c->d_tcell_start,
                           (const float4 *)c->d_tcell_rec, cap, c->d_tri_ring, d_stats.p);
        HIPCHK(hipGetLastError());
        unsigned long long h[oa::RING_STAT_N];
        { int rcr = read_small(c, h, d_stats, sizeof(h)); if (rcr) return rcr; }
        const double nt = (double)std::max(1ull, h[oa::RING_STAT_TRIS]);
        fprintf(stderr, "[oa] tri ring: %llu triangles, %.2f%% never accept, %.2f%% at the cap (%.3g); per triangle: %.1f records read, %.1f triangles tested, %.2f neighbours\n",
                h[oa::RING_STAT_TRIS], 100.0 * h[oa::RING_STAT_NEVER] / nt, 100.0 * h[oa::RING_STAT_CAPPED] / nt, cap, h[oa::RING_STAT_RECORDS] / nt,
                h[oa::RING_STAT_TESTS] / nt, h[oa::RING_STAT_NEIGHBOURS] / nt);
    } else {
        hipLaunchKernelGGL(oa::k_tri_ring_build<false>, blocks, dim3(256), 0, c->stream, c->d_tri9, c->n_tris, c->tgp, (const int *)c->d_tcell_start,
                           (const float4 *)c->d_tcell_rec, cap, c->d_tri_ring, (unsigned long long *)nullptr);
        HIPCHK(hipGetLastError());
    }
    c->tri_ring_ok = true;
    return OA_OK;
#endif
}

int launch_tri_search(oa_ctx *c, bool acc)
{
    if (bvh_whole(c, c->tbvh_ok, tri_tree_max(c))) return launch_bvh<true>(c, nullptr, nullptr, -1, acc);
    const bool use_grid = c->tri_grid_ok && c->tbvh_ok && c->grid_mode != 0;
    if (c->debug)
        fprintf(stderr, "[oa] tri search: grid=%d acc=%d ns=%d n_tris=%d state=%p src4=%p tri9=%p prev=%p keys=%p todo=%p/%p cells=%p/%p\n",
                (int)use_grid, (int)acc, c->ns, c->n_tris, (void *)c->d_state, (void *)c->d_src4, (void *)c->d_tri9, (void *)c->d_prev,
                (void *)c->d_keys, (void *)c->d_todo_list, (void *)c->d_todo_count, (void *)c->d_tcell_start, (void *)c->d_tcell_rec);
    if (use_grid) {
        if (!acc && !c->loop_active) { HIPCHK(hipMemsetAsync(c->d_todo_count, 0, TODO_COUNT_INTS * sizeof(int), c->stream)); c->u_slot = 0; }
        if (c->loop_active) { const int rcr = tri_ring_lazy(c, true); if (rcr) return rcr; }
        const int *ring = c->tri_ring_ok ? c->d_tri_ring : nullptr;
        const bool dual = c->grid_mode == -1 && c->turns_on && c->ns <= tri_tree_early(c);
        // With the neighbour lists and seeds: k_tri_accept settles what seed + neighbours settle and lists the rest, the grid
        // search works through the list (oa_tri.hpp).  Not for the accumulating form (its rows go by workgroup), nor while tree
        // and grid take turns (DevState::tree_turn): those keep the test in the search's own prologue.
        bool split = ring && !acc && !dual && c->seeded && c->tri_split && c->d_ulist;
        const int *qlist = nullptr, *qcount = nullptr;
        int qcap = 0;
        // The settled-pose search (oa_tri_fine.hpp) in front: one thread per query, its own cell's list of whole triangles; what it
        // does not settle (reach beyond the lists' inflation, crowded cells, no seed) is the list the grid search works through
#if defined(OA_EXPERIMENTS)
        const bool fine = c->tri_fine_ok && !acc && !dual && c->seeded && c->d_ulist;
        if (fine) {
            const dim3 sblocks((unsigned)(((long long)c->ns * 4 + 255) / 256));     // four lanes per query
            qcap = oa::ulist_cap((int)sblocks.x * 4, 16);           // 16 queries per wave
#define OA_TSETTLE_ARGS (const oa::DevState *)c->d_state, (const float4 *)c->d_src4, c->ns, c->tfp, (const uint4 *)c->d_tfine_table, (const float4 *)c->d_tfine_rec, \
                        (const float4 *)c->d_tri9, (const int *)c->d_prev, c->d_keys, c->d_ulist, qcap, ucount_set(c, c->u_slot), ucount_set(c, c->u_slot ^ 1)
            if (c->grid_stats) {                                     // OA_GRID_STATS=1: instrumented launch, totals to stderr (synchronises)
                DevTmp<unsigned long long> d_stats;
                HIPCHK(d_stats.alloc(8));
                HIPCHK(hipMemsetAsync(d_stats, 0, 8 * sizeof(unsigned long long), c->stream));
                hipLaunchKernelGGL(oa::k_tri_settle<true>, sblocks, dim3(256), 0, c->stream, OA_TSETTLE_ARGS, d_stats.p);
                HIPCHK(hipGetLastError());
                unsigned long long h[8];
                { int rcr = read_small(c, h, d_stats, sizeof(h)); if (rcr) return rcr; }
                fprintf(stderr, "[oa] tri settle: %llu of %d queries settled (%.2f records each); the others: %llu no seed, %llu reach beyond rho, %llu outside the box, "
                                "%llu cell not listed / over the cap\n", h[0], c->ns, (double)h[5] / (double)std::max(1ull, h[0]), h[1], h[2], h[3], h[4]);
            } else
            hipLaunchKernelGGL(oa::k_tri_settle<false>, sblocks, dim3(256), 0, c->stream, OA_TSETTLE_ARGS, (unsigned long long *)nullptr);
#undef OA_TSETTLE_ARGS
            HIPCHK(hipGetLastError());
            qlist = c->d_ulist; qcount = ucount_set(c, c->u_slot);
            c->u_slot ^= 1;
            ring = nullptr;
            split = true;
        }
        else if (split) {
            hipLaunchKernelGGL(oa::k_tri_accept, dim3((unsigned)((c->ns + 255) / 256)), dim3(256), 0, c->stream, (const oa::DevState *)c->d_state,
                               (const float4 *)c->d_src4, c->ns, (float)c->tgp.scale * 1.000001f, (const float4 *)c->d_tri9, (const int *)c->d_prev, ring,
                               c->d_keys, c->d_ulist, (qcap = oa::ulist_cap((c->ns + 63) / 64, 64)), ucount_set(c, c->u_slot), ucount_set(c, c->u_slot ^ 1));
            HIPCHK(hipGetLastError());
            qlist = c->d_ulist; qcount = ucount_set(c, c->u_slot);
            c->u_slot ^= 1;
            ring = nullptr;                                          // (the list's queries failed that test already)
        }
#endif
        if (dual) { int rcb = launch_bvh<true>(c, nullptr, nullptr, 1); if (rcb) return rcb; }    // runs when DevState::tree_turn
        const int turn = dual ? 0 : -1;
        const int lanes = tri_lanes_for(c);
#define OA_TGRID_ARGS c->d_state, c->d_src4, c->ns, c->tgp, c->d_tcell_start, c->d_tcell_rec, c->d_tri9, c->d_prev, c->d_keys, c->d_todo_list, c->d_todo_count, turn
        const dim3 gblocks((unsigned)(((long long)c->ns * lanes + 255) / 256));
        if (split) {
            // the list's length decides the lanes per query, on the device (k_tri_search_grid: qmin / qmax): a 4-lane launch for
            // lists of up to `small` queries, and the shard's own geometry for longer ones
            const int small = std::min(c->ns, (c->n_tris >= 250000 ? 400 : 128) * c->n_cu);
#define OA_TGRID_LIST_TAIL(lo, hi) (unsigned long long *)nullptr, oa::BvhParams{}, (const float4 *)nullptr, (const float4 *)nullptr, oa::NormalTest{}, (double *)nullptr, (const int *)nullptr, qlist, qcount, lo, hi, qcap
            if (lanes != 4 && c->tri_split_lanes) {
                hipLaunchKernelGGL(oa::k_tri_search_grid<4>, dim3((unsigned)(((long long)small * 4 + 255) / 256)), dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(0, small));
                if (lanes == 2) hipLaunchKernelGGL(oa::k_tri_search_grid<2>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(small, 0x7FFFFFFF));
                else hipLaunchKernelGGL(oa::k_tri_search_grid<1>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(small, 0x7FFFFFFF));
            } else {
                if (lanes == 4) hipLaunchKernelGGL(oa::k_tri_search_grid<4>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(0, 0x7FFFFFFF));
                else if (lanes == 2) hipLaunchKernelGGL(oa::k_tri_search_grid<2>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(0, 0x7FFFFFFF));
                else hipLaunchKernelGGL(oa::k_tri_search_grid<1>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, OA_TGRID_LIST_TAIL(0, 0x7FFFFFFF));
            }
#undef OA_TGRID_LIST_TAIL
            HIPCHK(hipGetLastError());
            return launch_bvh<true>(c, c->d_todo_list, c->d_todo_count);       // the far queries: tree search
        }
        if (acc) {
            // the search finishes its own leftovers through the triangle tree and takes the pair test and the sums in its
            // epilogue (search_plan: PLAN_GRID, fast path): no list launch, no accumulation launch
#define OA_TGRID_ACC_ARGS OA_TGRID_ARGS, (unsigned long long *)nullptr, c->tbvh, (const float4 *)c->d_tbvh_box, (const float4 *)c->d_tbvh_prims, normal_test(c), c->d_partials, ring
            if (lanes == 4) hipLaunchKernelGGL((oa::k_tri_search_grid<4, false, true, true>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ACC_ARGS);
            else if (lanes == 2) hipLaunchKernelGGL((oa::k_tri_search_grid<2, false, true, true>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ACC_ARGS);
            else hipLaunchKernelGGL((oa::k_tri_search_grid<1, false, true, true>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ACC_ARGS);
#undef OA_TGRID_ACC_ARGS
            HIPCHK(hipGetLastError());
            return OA_OK;
        }
#define OA_TGRID_TAIL oa::BvhParams{}, (const float4 *)nullptr, (const float4 *)nullptr, oa::NormalTest{}, (double *)nullptr, ring, qlist, qcount
        // long plain searches: one wave per workgroup (k_tri_search_grid, BT = 64): a slot is free again when ITS wave is through
        const long long wblocks = ((long long)c->ns * lanes + 63) / 64;
        const bool wave_wgs = c->tri_wave_wgs && !qlist && !c->grid_stats && c->tri_share && wblocks >= (long long)c->n_cu * 32;
        if (wave_wgs) {
            const dim3 wb((unsigned)wblocks);
            if (lanes == 4) hipLaunchKernelGGL((oa::k_tri_search_grid<4, false, true, false, 64>), wb, dim3(64), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
            else if (lanes == 2) hipLaunchKernelGGL((oa::k_tri_search_grid<2, false, true, false, 64>), wb, dim3(64), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
            else hipLaunchKernelGGL((oa::k_tri_search_grid<1, false, true, false, 64>), wb, dim3(64), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
        }
        else if (lanes == 4) hipLaunchKernelGGL(oa::k_tri_search_grid<4>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
        else if (lanes == 2) hipLaunchKernelGGL(oa::k_tri_search_grid<2>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
#if defined(OA_EXPERIMENTS)
        else if (c->grid_stats) {                                   // OA_GRID_STATS=1: instrumented launch, totals to stderr (synchronises)
            DevTmp<unsigned long long> d_stats;                     // one row of counters per wave (atomics on a dozen shared words slowed the launch 8x)
            const size_t n_waves = (size_t)gblocks.x * 4;
            HIPCHK(d_stats.alloc(oa::TRI_STAT_N * n_waves));
            HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * oa::TRI_STAT_N * n_waves, c->stream));
            if (c->tri_share) hipLaunchKernelGGL((oa::k_tri_search_grid<1, true, true>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, d_stats.p, OA_TGRID_TAIL);
            else hipLaunchKernelGGL((oa::k_tri_search_grid<1, true, false>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, d_stats.p, OA_TGRID_TAIL);
            HIPCHK(hipGetLastError());
            unsigned long long h[oa::TRI_STAT_N] = { 0 };
            {
                std::vector<unsigned long long> rows(oa::TRI_STAT_N * n_waves);
                int rcr = read_small(c, rows.data(), d_stats, sizeof(unsigned long long) * rows.size());
                if (rcr) return rcr;
                for (size_t w = 0; w < n_waves; ++w) for (int k = 0; k < oa::TRI_STAT_N; ++k) h[k] += rows[w * oa::TRI_STAT_N + (size_t)k];
            }
            const double nq = (double)std::max(1ull, h[oa::TRI_STAT_QUERIES]), nw = (double)std::max(1ull, h[oa::TRI_STAT_WAVES]);
            fprintf(stderr, "[oa] tri grid stats: queries %llu | per query: rows %.2f entries %.2f (sphere passes %.2f) survivors %.2f evals %.2f | per wave: "
                            "eval trips %.1f max-lane entries %.1f max-lane rows %.1f | ring>=2 %.1f%% ring>=3 %.1f%% unsettled %.1f%% over budget %.1f%%\n",
                    h[oa::TRI_STAT_QUERIES], h[oa::TRI_STAT_ROWS] / nq, h[oa::TRI_STAT_ENTRIES] / nq, h[oa::TRI_STAT_SPHERE] / nq, h[oa::TRI_STAT_SURVIVORS] / nq,
                    h[oa::TRI_STAT_EVALS] / nq, h[oa::TRI_STAT_WAVE_TRIPS] / nw, h[oa::TRI_STAT_WAVE_MAX_ENTRIES] / nw,
                    h[oa::TRI_STAT_WAVE_MAX_ROWS] / nw, 100.0 * h[oa::TRI_STAT_RING2] / nq, 100.0 * h[oa::TRI_STAT_RING3] / nq,
                    100.0 * h[oa::TRI_STAT_UNSETTLED] / nq, 100.0 * h[oa::TRI_STAT_OVER] / nq);
            const double ct = (double)std::max(1ull, h[oa::TRI_STAT_CYC_TOTAL]);
            fprintf(stderr, "[oa] tri grid phases: per wave %.0f shader cycles, %.2f loop trips | prologue %.1f%% listing %.1f%% scan %.1f%% flush %.1f%% bookkeeping %.1f%% epilogue %.1f%%\n",
                    ct / nw, h[oa::TRI_STAT_LOOP_TRIPS] / nw, 100.0 * h[oa::TRI_STAT_CYC_PROLOGUE] / ct, 100.0 * h[oa::TRI_STAT_CYC_LIST] / ct,
                    100.0 * h[oa::TRI_STAT_CYC_SCAN] / ct, 100.0 * h[oa::TRI_STAT_CYC_FLUSH] / ct, 100.0 * h[oa::TRI_STAT_CYC_BOOK] / ct,
                    100.0 * (ct - h[oa::TRI_STAT_CYC_PROLOGUE] - h[oa::TRI_STAT_CYC_LIST] - h[oa::TRI_STAT_CYC_SCAN] - h[oa::TRI_STAT_CYC_FLUSH] - h[oa::TRI_STAT_CYC_BOOK]) / ct);
        }
        else if (!c->tri_share) hipLaunchKernelGGL((oa::k_tri_search_grid<1, false, false>), gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
#endif
        else hipLaunchKernelGGL(oa::k_tri_search_grid<1>, gblocks, dim3(256), 0, c->stream, OA_TGRID_ARGS, (unsigned long long *)nullptr, OA_TGRID_TAIL);
#undef OA_TGRID_TAIL
#undef OA_TGRID_ARGS
        HIPCHK(hipGetLastError());
        if (c->debug) {                                            // how many queries the grid handed over (debug only: syncs)
            int n_todo = 0;
            { int rcr = read_small(c, &n_todo, c->d_todo_count, sizeof(int)); if (rcr) return rcr; }
            fprintf(stderr, "[oa] tri grid handed %d of %d queries to the tree\n", n_todo, c->ns);
        }
        return launch_bvh<true>(c, c->d_todo_list, c->d_todo_count);       // the far queries: tree search
    } else {
        hipLaunchKernelGGL(oa::k_tri_search_all, dim3(std::min((c->ns + 255) / 256, 65535)), dim3(256), 0, c->stream,
                           c->d_state, c->d_src4, c->ns, c->d_tri9, c->n_tris, c->d_prev, c->d_keys);
    }
    HIPCHK(hipGetLastError());
    return OA_OK;
}
}  // namespace

OA_EXPORT int oa_set_target_mesh(oa_ctx *c, const float *xyz, int64_t n_verts, int on_device, const int32_t *tris,
                                 int64_t n_tris)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (!c->subs.empty() && c->loop_active) multi_abort(c);
    OA_ROUTE_ALL_PAR(c, oa_set_target_mesh(sub, xyz, n_verts, on_device, tris, n_tris));
    if (n_tris < 1 || !tris) return fail(OA_E_BAD_ARG, "oa_set_target_mesh: no triangles");
    if (n_tris > 0x2AAAAAA0ll) return fail(OA_E_BAD_ARG, "too many triangles");
    int rc = set_target_common(c, xyz, n_verts, on_device, false);  // vertex images + bbox + filter; no vertex grid / tree
    if (rc) return rc;
    if (n_verts < 1) return fail(OA_E_BAD_ARG, "oa_set_target_mesh: no vertices");
    DevTmp<int> d_tris;
    constexpr size_t SLOT_WORDS = (size_t)oa::TOTAL_SLOTS * oa::TOTAL_STRIDE;
    DevTmp<double> d_chk;                                           // [0]: (as int) corners that index outside the vertices; [1 ...]: the spread sum of the bounding-box diagonals (oa_tri.hpp: TOTAL_SLOTS)
    HIPCHK(d_tris.alloc(3 * (size_t)n_tris));
    HIPCHK(d_chk.alloc(1 + SLOT_WORDS));
    HIPCHK(hipMemcpyAsync(d_tris, tris, sizeof(int) * 3 * (size_t)n_tris, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_chk, 0, (1 + SLOT_WORDS) * sizeof(double), c->stream));
    HIPCHK(dev_malloc(&c->d_tri9, sizeof(float4) * 3 * (size_t)n_tris));
    hipLaunchKernelGGL(oa::k_pack_tris, dim3((unsigned)((n_tris + 255) / 256)), dim3(256), 0, c->stream, c->d_tgt_xyz,
                       (int)n_verts, (const int *)d_tris.p, (int)n_tris, c->d_tri9, (int *)d_chk.p);
    // the triangle grid's cell size comes from the mean bounding-box diagonal: summed here, so that the index check and the
    // sum come back in ONE host round trip (a bad index leaves a triangle of garbage corners: the sum is not used then)
    hipLaunchKernelGGL(oa::k_tri_diag_sum, dim3((unsigned)((n_tris + 255) / 256)), dim3(256), 0, c->stream, c->d_tri9, (int)n_tris, d_chk.p + 1);
    HIPCHK(hipGetLastError());
    double chk[1 + SLOT_WORDS];
    { int rcr = read_small(c, chk, d_chk, sizeof(chk)); if (rcr) return rcr; }
    int bad = 0;
    memcpy(&bad, &chk[0], sizeof(int));
    double diag_sum = 0.0;
    for (int k = 0; k < oa::TOTAL_SLOTS; ++k) diag_sum += chk[1 + (size_t)k * oa::TOTAL_STRIDE];
    if (bad) { dev_free(c->d_tri9); return fail(OA_E_BAD_ARG, "oa_set_target_mesh: %d triangle corners index outside 0..%lld", bad, (long long)n_verts - 1); }
    c->n_tris = (int)n_tris;
    c->surface = true;
    if ((rc = build_bvh(c, true))) return rc;
    return build_tri_grid(c, &diag_sum);
}

namespace {
// n floats of caller data, readable by c's device on c's stream: host data and pointers that live on ANOTHER GPU are
// copied into `tmp` (a kernel of device i must not dereference device j's memory: nothing enables peer access for
// caller buffers); a pointer on c's own device is used as it is
int stage_floats(oa_ctx *c, const float *src, size_t n, int on_device, DevTmp<float> &tmp, const float *&local)
{
    local = src;
    if (n == 0) return OA_OK;
    if (!on_device) {
        HIPCHK(tmp.alloc(n));
        HIPCHK(hipMemcpyAsync(tmp, src, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        local = tmp;
        return OA_OK;
    }
    const int dev = pointer_device(src);
    if (dev < 0 || dev == c->device) return OA_OK;
    HIPCHK(tmp.alloc(n));
    HIPCHK(hipMemcpyPeerAsync(tmp, c->device, src, dev, sizeof(float) * n, c->stream));
    local = tmp;
    return OA_OK;
}

// release the old shard and allocate everything a shard of `count` selected points needs (seeds cleared, keys empty)
int source_reset(oa_ctx *c, long long count, long long begin, long long n_verts)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loop_active = false;                                         // an open oa_iterate sequence ends with the old source
    dev_free(c->d_src4); dev_free(c->d_keys); dev_free(c->d_prev); dev_free(c->d_win); dev_free(c->d_wsafe); dev_free(c->d_sel); dev_free(c->d_src_n);
    dev_free(c->d_src4o); dev_free(c->d_perm); dev_free(c->d_members); dev_free(c->d_pos);
    c->h_members.clear();
    c->shard_begin = begin;
    c->normals_on = false;
    c->src_n_verts = n_verts;
    dev_free(c->d_valid); dev_free(c->d_b); dev_free(c->d_dist); dev_free(c->d_counts); dev_free(c->d_offsets);
    dev_free(c->d_A); dev_free(c->d_B);
    c->emit_cap = 0;
    c->ns = (int)count;
    // points per thread: 4 for big shards; small shards take 2 or 1 so that the launch still has enough workgroups
    c->R = c->R_env ? c->R_env : (count >= 65536 ? 4 : (count >= 16384 ? 2 : 1));
    const int chunk = oa::NN_THREADS * c->R;
    c->ns_pad = (int)(((count + chunk - 1) / chunk) * chunk);
    if (c->ns_pad == 0) c->ns_pad = chunk;
    HIPCHK(dev_malloc(&c->d_src4, sizeof(float4) * (size_t)c->ns_pad));
    HIPCHK(dev_malloc(&c->d_keys, sizeof(unsigned long long) * (size_t)c->ns_pad));
    HIPCHK(dev_malloc(&c->d_prev, sizeof(int) * (size_t)c->ns_pad));
    HIPCHK(dev_malloc(&c->d_win, sizeof(float4) * (size_t)c->ns_pad));
    dev_free(c->d_worder); dev_free(c->d_homes);
    if (c->nn_wave_order && c->grid_mode == 0) HIPCHK(dev_malloc(&c->d_worder, sizeof(unsigned short) * (size_t)c->ns_pad));   // (brute force only; never inside a loop)
    if (c->grid_safe && !c->surface) HIPCHK(dev_malloc(&c->d_wsafe, sizeof(uint2) * (size_t)c->ns_pad));   // (a vertex target set later allocates it: set_target_common)
    c->seeded = false; c->win_seeds = false;
    HIPCHK(dev_malloc(&c->d_sel, sizeof(int) * (size_t)c->ns_pad));
    dev_free(c->d_todo_list); dev_free(c->d_todo_count); dev_free(c->d_ulist);
    HIPCHK(dev_malloc(&c->d_todo_list, sizeof(int) * (size_t)c->ns_pad));
    HIPCHK(dev_malloc(&c->d_ulist, sizeof(int) * ((size_t)c->ns_pad + 64 * (size_t)oa::ULIST_PARTS)));   // (ULIST_PARTS regions, each rounded up to whole waves)
    HIPCHK(dev_malloc(&c->d_todo_count, TODO_COUNT_INTS * sizeof(int)));      // {entries of the hand-over list, most handed over by one wave, ...}, the two counter sets of d_ulist
    HIPCHK(hipMemsetAsync(c->d_todo_count, 0, TODO_COUNT_INTS * sizeof(int), c->stream));
    c->u_slot = 0;
    // no seeds, empty keys, sel = 0, hand-over counters at zero: one launch
    hipLaunchKernelGGL(oa::k_init_slots, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, c->d_prev, c->d_win, c->d_wsafe, c->d_keys, c->d_sel,
                       c->d_todo_count, c->ns_pad);
    HIPCHK(hipGetLastError());
    c->pivot[0] = c->pivot[1] = c->pivot[2] = 0.0;
    return OA_OK;
}

// one context takes shard shard_index of shard_count: packs (and, for several shards, first Morton-sorts) the
// selection by itself.  The one-process-per-GPU path, single-device contexts, and the fallback of multi-device ones.
int set_source_single(oa_ctx *c, const float *xyz, int64_t n_verts, int on_device, const int64_t *vlist, int64_t n_vlist,
                      long long step, long long n_sel, int32_t shard_index, int32_t shard_count)
{
    const long long per = (n_sel + shard_count - 1) / shard_count;
    const long long begin = std::min(n_sel, per * shard_index);
    const long long count = std::max(0ll, std::min(per, n_sel - begin));
    if (count > 0x7FF00000ll) return fail(OA_E_BAD_ARG, "shard too large");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = source_reset(c, count, begin, n_verts))) return rc;
    if (n_sel > 0) {
        const float *d_xyz = xyz;
        DevTmp<float> tmp_xyz;
        DevTmp<long long> d_vlist;
        if ((rc = stage_floats(c, xyz, 3 * (size_t)n_verts, on_device, tmp_xyz, d_xyz))) return rc;
        if (vlist) {
            HIPCHK(d_vlist.alloc((size_t)n_vlist));
            HIPCHK(hipMemcpyAsync(d_vlist, vlist, sizeof(long long) * (size_t)n_vlist, hipMemcpyHostToDevice, c->stream));
        }
        // pivot = first selected vertex of the WHOLE selection (identical on every shard)
        const long long v0 = vlist ? vlist[0] : 0;
        float p0[3];
        if (on_device) HIPCHK(hipMemcpyAsync(p0, d_xyz + 3 * v0, sizeof p0, hipMemcpyDeviceToHost, c->stream));
        else memcpy(p0, xyz + 3 * v0, sizeof p0);
        // Which points of the selection this shard holds.  One shard: all of them.  Several shards: the
        // shard_index-th of shard_count equal ranges of the selection IN MORTON ORDER (every rank sorts the whole
        // selection and gets the same order), so that a shard is a compact region of space -- the brute-force
        // filter and the grid both work per wave of neighbouring points, and a shard that is a sparse sample of the
        // whole cloud costs 16 % more per pair (1/8 of 1M: 7.3 ms vs 6.3 ms).  Fallback (non-finite coordinates,
        // OA_SHARD_SPATIAL=0): contiguous ranges of the selection in the caller's order.
        DevTmp<int> d_members;
        if (shard_count > 1 && c->ns > 0 && n_sel < 0x7FF00000ll && env_int("OA_SORT_SOURCE", 1) && env_int("OA_SHARD_SPATIAL", 1)) {
            int rcm = spatial_shard_members(c, d_xyz, n_verts, (const long long *)d_vlist.p, step, n_sel, begin, c->ns, d_members);
            if (rcm) return rcm;
            if (d_members.p) {                                      // kept: per-point outputs of a shard go back to vlist order through it
                HIPCHK(dev_malloc(&c->d_members, sizeof(int) * (size_t)c->ns));
                HIPCHK(hipMemcpyAsync(c->d_members, d_members.p, sizeof(int) * (size_t)c->ns, hipMemcpyDeviceToDevice, c->stream));
                if (c->parent) {
                    c->h_members.resize((size_t)c->ns);
                    HIPCHK(hipMemcpyAsync(c->h_members.data(), d_members.p, sizeof(int) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
                }
            }
        }
        if (c->ns > 0) {
            hipLaunchKernelGGL(oa::k_pack_source, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, d_xyz,
                               (const long long *)d_vlist.p, step, begin, (const int *)d_members.p, c->ns, c->ns_pad, c->d_src4, c->d_sel);
            HIPCHK(hipGetLastError());
        } else HIPCHK(hipMemsetAsync(c->d_src4, 0, sizeof(float4) * (size_t)c->ns_pad, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 3; ++k) c->pivot[k] = (double)p0[k];
        if (c->ns > 1 && env_int("OA_SORT_SOURCE", 1)) {
            int rcs = sort_source_slots(c, d_xyz, n_verts);
            if (rcs) return rcs;
        }
    } else {
        HIPCHK(hipMemsetAsync(c->d_src4, 0, sizeof(float4) * (size_t)c->ns_pad, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    plan_geometry(c);
    return OA_OK;
}

// The whole selection packed and Morton-sorted ONCE, on one device of a multi-device context (round 2: every device
// sorted the whole selection to find its range -- at 10M points, 8 sorts of 10M keys).  Slot j of the order holds the
// point at selection position pos[j]; shard g of n is the range [g per, g per + count_g) of it -- exactly the slots the
// per-context path (spatial_shard_members + sort_source_slots) ends up with, in the same order: both sort by the same
// 30-bit key with ties in selection order.
struct SelectionOrder {
    DevTmp<float4> pts;
    DevTmp<int> sel, pos;
    float p0[3] = { 0.f, 0.f, 0.f };
    bool ok = false;
};

int build_selection_order(oa_ctx *c, const float *xyz, int64_t n_verts, int on_device, const int64_t *vlist, int64_t n_vlist,
                          long long step, long long n_sel, SelectionOrder &o)
{
    o.ok = false;
    int rc = use_device(c);
    if (rc) return rc;
    const float *d_xyz = xyz;
    DevTmp<float> tmp_xyz;
    DevTmp<long long> d_vlist;
    if ((rc = stage_floats(c, xyz, 3 * (size_t)n_verts, on_device, tmp_xyz, d_xyz))) return rc;
    if (vlist) {
        HIPCHK(d_vlist.alloc((size_t)n_vlist));
        HIPCHK(hipMemcpyAsync(d_vlist, vlist, sizeof(long long) * (size_t)n_vlist, hipMemcpyHostToDevice, c->stream));
    }
    const long long v0 = vlist ? vlist[0] : 0;
    if (on_device) HIPCHK(hipMemcpyAsync(o.p0, d_xyz + 3 * v0, sizeof o.p0, hipMemcpyDeviceToHost, c->stream));
    else memcpy(o.p0, xyz + 3 * v0, sizeof o.p0);
    float lo[3], sc[3];
    bool finite = false;
    if ((rc = morton_frame(c, d_xyz, n_verts, lo, sc, finite))) return rc;   // synchronises the stream
    if (!finite) return OA_OK;                                     // caller falls back to contiguous ranges
    const int n = (int)n_sel;
    DevTmp<float4> all4;
    DevTmp<int> all_sel, v_in;
    DevTmp<unsigned> k_in, k_out;
    HIPCHK(all4.alloc((size_t)n)); HIPCHK(all_sel.alloc((size_t)n)); HIPCHK(v_in.alloc((size_t)n));
    HIPCHK(k_in.alloc((size_t)n)); HIPCHK(k_out.alloc((size_t)n));
    HIPCHK(o.pts.alloc((size_t)n)); HIPCHK(o.sel.alloc((size_t)n)); HIPCHK(o.pos.alloc((size_t)n));
    hipLaunchKernelGGL(oa::k_pack_source, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_xyz, (const long long *)d_vlist.p, step, 0ll,
                       (const int *)nullptr, n, n, all4.p, all_sel.p);
    hipLaunchKernelGGL(oa::k_morton_keys, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float4 *)all4.p, n,
                       lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], k_in.p, v_in.p);
    HIPCHK(hipGetLastError());
    { const int rcs = sort_pairs30(c, k_in.p, k_out.p, v_in.p, o.pos.p, (size_t)n); if (rcs) return rcs; }
    hipLaunchKernelGGL(oa::k_apply_perm, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float4 *)all4.p, (const int *)all_sel.p,
                       (const int *)o.pos.p, n, n, o.pts.p, o.sel.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));                      // temporaries are released on return; other devices read the order next
    o.ok = true;
    return OA_OK;
}

// device c takes the range [begin, begin + count) of the order built on `stage`'s device: three range copies over xGMI
// (24 B per point), then a sort of ITS points' selection positions to get back to the caller's order for per-point outputs
int adopt_shard(oa_ctx *c, const oa_ctx *stage, const SelectionOrder &o, long long begin, long long count, long long n_verts)
{
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = source_reset(c, count, begin, n_verts))) return rc;
    for (int k = 0; k < 3; ++k) c->pivot[k] = (double)o.p0[k];
    if (count == 0) {
        HIPCHK(hipMemsetAsync(c->d_src4, 0, sizeof(float4) * (size_t)c->ns_pad, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        plan_geometry(c);
        return OA_OK;
    }
    const int n = (int)count;
    DevTmp<int> pos, inv;
    HIPCHK(pos.alloc((size_t)n)); HIPCHK(inv.alloc((size_t)n));
    const bool local = stage->device == c->device;
#define OA_RANGE_COPY(dst, src, bytes)                                                                                   \
    HIPCHK(local ? hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToDevice, c->stream)                             \
                 : hipMemcpyPeerAsync((dst), c->device, (src), stage->device, (bytes), c->stream))
    OA_RANGE_COPY(c->d_src4, o.pts.p + begin, sizeof(float4) * (size_t)n);
    OA_RANGE_COPY(c->d_sel, o.sel.p + begin, sizeof(int) * (size_t)n);
    OA_RANGE_COPY(pos.p, o.pos.p + begin, sizeof(int) * (size_t)n);
#undef OA_RANGE_COPY
    // caller order inside the shard = ascending selection position
    HIPCHK(dev_malloc(&c->d_members, sizeof(int) * (size_t)n));
    HIPCHK(dev_malloc(&c->d_perm, sizeof(int) * (size_t)n));
    HIPCHK(dev_malloc(&c->d_src4o, sizeof(float4) * (size_t)c->ns_pad));
    // (inv = the shard's slots in ascending selection position: a stable argsort of pos; d_members = pos in that order)
    { const int rcs = sort_ints(c, pos.p, c->d_members, inv.p, (size_t)n, 31); if (rcs) return rcs; }
    hipLaunchKernelGGL(oa::k_finish_shard, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, (const int *)inv.p, n, c->ns_pad,
                       c->d_src4, c->d_sel, c->d_src4o, c->d_perm);
    HIPCHK(hipGetLastError());
    c->h_members.resize((size_t)n);
    HIPCHK(hipMemcpyAsync(c->h_members.data(), c->d_members, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    plan_geometry(c);
    return OA_OK;
}
}  // namespace

OA_EXPORT int oa_set_source(oa_ctx *c, const float *xyz, int64_t n_verts, int on_device, const int64_t *vlist,
                            int64_t n_vlist, int32_t stride, int32_t shard_index, int32_t shard_count)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (n_verts < 0 || (n_verts > 0 && !xyz)) return fail(OA_E_BAD_ARG, "bad source array");
    if (vlist && n_vlist < 0) return fail(OA_E_BAD_ARG, "negative vlist length");
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
        return fail(OA_E_BAD_ARG, "bad shard %d of %d", shard_index, shard_count);
    const long long step = stride > 1 ? stride : 1;                 // sample > 1 -> vlist[0::sample] (general.py:274)
    const long long n_all = vlist ? n_vlist : n_verts;
    const long long n_sel = (n_all + step - 1) / step;
    if (vlist)
        for (long long i = 0; i < n_vlist; ++i)
            if (vlist[i] < 0 || vlist[i] >= n_verts)
                return fail(OA_E_BAD_ARG, "vlist[%lld] = %lld outside 0..%lld", i, (long long)vlist[i], (long long)n_verts - 1);
    if (c->subs.empty()) return set_source_single(c, xyz, n_verts, on_device, vlist, n_vlist, step, n_sel, shard_index, shard_count);

    // multi-device: child i keeps shard i of n_dev
    if (shard_count != 1) return fail(OA_E_BAD_ARG, "oa_set_source: a multi-device context shards the source itself (pass shard 0 of 1)");
    if (c->loop_active) multi_abort(c);
    c->loop_active = false;
    const int n_dev = (int)c->subs.size();
    const long long per = (n_sel + n_dev - 1) / n_dev;
    if (n_dev > 1 && n_sel > 1 && n_sel < 0x7FF00000ll && env_int("OA_SORT_SOURCE", 1) && env_int("OA_SHARD_SPATIAL", 1) &&
        env_int("OA_PARTITION_ONCE", 1)) {
        // one Morton sort of the selection on the first device, then every device adopts its range of it
        oa_ctx *stage = c->subs[0];
        SelectionOrder order;
        int rc = build_selection_order(stage, xyz, n_verts, on_device, vlist, n_vlist, step, n_sel, order);
        if (rc) return rc;
        if (order.ok) {
            const SelectionOrder *po = &order;
            return route_all_parallel(c, [=](oa_ctx *sub) -> int {
                const long long begin = std::min(n_sel, per * sub->rank);
                const long long count = std::max(0ll, std::min(per, n_sel - begin));
                return adopt_shard(sub, stage, *po, begin, count, n_verts);
            });
        }
    }
    // fallback (non-finite coordinates, a single device, sorting switched off): every child finds its shard itself
    OA_ROUTE_ALL_PAR(c, set_source_single(sub, xyz, n_verts, on_device, vlist, n_vlist, step, n_sel, sub->rank, n_dev));
    return OA_OK;
}

OA_EXPORT int oa_set_normals(oa_ctx *c, const float *src_normals, int64_t n_verts, const float *tgt_normals, int64_t nt,
                             double max_angle_deg)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (!c->subs.empty() && c->loop_active) multi_abort(c);
    OA_ROUTE_ALL_PAR(c, oa_set_normals(sub, src_normals, n_verts, tgt_normals, nt, max_angle_deg));
    if (c->loop_active) { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipStreamSynchronize(c->stream)); c->loop_active = false; }
    c->normals_on = false;
    if (!src_normals || !(max_angle_deg > 0.0) || !(max_angle_deg < 180.0)) return OA_OK;       // switched off
    if (!c->d_src4 || !c->d_sel) return fail(OA_E_STATE, "oa_set_normals: call oa_set_source first");
    if (c->nt <= 0) return fail(OA_E_STATE, "oa_set_normals: call oa_set_target first");
    if (n_verts != c->src_n_verts) return fail(OA_E_BAD_ARG, "oa_set_normals: %lld source normals for %lld vertices", (long long)n_verts, c->src_n_verts);
    if (!c->surface && (!tgt_normals || nt != c->nt)) return fail(OA_E_BAD_ARG, "oa_set_normals: vertex mode needs one normal per target vertex");
    int rc = use_device(c);
    if (rc) return rc;
    DevTmp<float> tmp;
    HIPCHK(tmp.alloc(3 * (size_t)n_verts));
    HIPCHK(hipMemcpyAsync(tmp, src_normals, sizeof(float) * 3 * (size_t)n_verts, hipMemcpyHostToDevice, c->stream));
    dev_free(c->d_src_n); dev_free(c->d_tgt_n);
    HIPCHK(dev_malloc(&c->d_src_n, sizeof(float) * 3 * (size_t)std::max(1, c->ns)));
    if (c->ns > 0)
        hipLaunchKernelGGL(oa::k_gather_rows3, dim3((c->ns + 255) / 256), dim3(256), 0, c->stream, (const float *)tmp.p,
                           (const int *)c->d_sel, c->ns, c->d_src_n);
    HIPCHK(hipGetLastError());
    if (!c->surface) {
        HIPCHK(dev_malloc(&c->d_tgt_n, sizeof(float) * 3 * (size_t)nt));
        HIPCHK(hipMemcpyAsync(c->d_tgt_n, tgt_normals, sizeof(float) * 3 * (size_t)nt, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->cos_min = cos(max_angle_deg * 3.14159265358979323846 / 180.0);
    c->normals_on = true;
    return OA_OK;
}

OA_EXPORT int oa_set_matrices(oa_ctx *c, const float mx_align[16], const float mx_base[16])
{
    if (!c || !mx_align || !mx_base) return fail(OA_E_BAD_ARG, "oa_set_matrices: null argument");
    if (!c->subs.empty()) c->loop_active = false;
    OA_ROUTE_ALL(c, oa_set_matrices(sub, mx_align, mx_base));
    float i1[16], i2[16];
    if (!oa::m4_inverted(mx_align, i1)) return fail(OA_E_SINGULAR, "align matrix_world has no inverse");
    if (!oa::m4_inverted(mx_base, i2)) return fail(OA_E_SINGULAR, "base matrix_world has no inverse");
    memcpy(c->h_state.mx1, mx_align, sizeof i1);
    memcpy(c->h_state.mx2, mx_base, sizeof i1);
    memcpy(c->h_state.imx1, i1, sizeof i1);
    memcpy(c->h_state.imx2, i2, sizeof i2);
    c->have_mats = true;
    c->loop_active = false;
    c->last_todo_wave_max = -1;                                     // a new pose: what the last loop handed over says nothing about the next one
    return OA_OK;
}

OA_EXPORT int oa_reset_seeds(oa_ctx *c)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    if (!c->subs.empty()) { if (c->loop_active) multi_abort(c); c->loop_active = false; }
    OA_ROUTE_ALL(c, oa_reset_seeds(sub));
    c->seeded = false; c->win_seeds = false;
    c->loop_active = false;
    if (!c->d_prev) return OA_OK;
    int rc = use_device(c);
    if (rc) return rc;
    hipLaunchKernelGGL(oa::k_init_slots, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, c->d_prev, c->d_win, c->d_wsafe,
                       (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr, c->ns_pad);
    HIPCHK(hipGetLastError());
    return OA_OK;
}

OA_EXPORT int oa_get_stat(oa_ctx *c, int what, double *value)
{
    if (!c || !value) return fail(OA_E_BAD_ARG, "oa_get_stat: null argument");
    if (what == OA_STAT_CACHE_BYTES) { *value = (double)dev_cache().cached_bytes; return OA_OK; }
    if (what == OA_STAT_ENQUEUED_MIN || what == OA_STAT_ENQUEUED_MAX || what >= OA_STAT_ENQUEUED_CHILD) {
        // iterations the host enqueued for the children in the last oa_run (the agreement of docs/HISTORY.md 4.7: all equal)
        if (c->subs.empty()) { *value = (double)c->iter_enq; return OA_OK; }
        if (what >= OA_STAT_ENQUEUED_CHILD) {
            const size_t i = (size_t)(what - OA_STAT_ENQUEUED_CHILD);
            if (i >= c->subs.size()) return fail(OA_E_BAD_ARG, "oa_get_stat: child %zu of %zu", i, c->subs.size());
            *value = (double)c->subs[i]->enq_iters;
            return OA_OK;
        }
        long long lo = c->subs[0]->enq_iters, hi = lo;
        for (oa_ctx *sub : c->subs) { lo = std::min(lo, sub->enq_iters); hi = std::max(hi, sub->enq_iters); }
        *value = (double)(what == OA_STAT_ENQUEUED_MIN ? lo : hi);
        return OA_OK;
    }
    if (what == OA_STAT_NN_MS_MIN || what == OA_STAT_NN_MS_MAX) {
        if (c->subs.empty()) return fail(OA_E_STATE, "oa_get_stat: per-device search times are a multi-device context's");
        double lo = c->subs[0]->last_nn_ms, hi = lo;
        for (oa_ctx *sub : c->subs) { lo = std::min(lo, sub->last_nn_ms); hi = std::max(hi, sub->last_nn_ms); }
        *value = what == OA_STAT_NN_MS_MIN ? lo : hi;
        return OA_OK;
    }
    if (what == OA_STAT_TRI_RING_ACCEPTS) {
        // the prologue's test of k_tri_search_grid, counted: device state (pose) and seeds as the last search / loop left them
        *value = 0.0;
        if (!c->subs.empty()) {
            for (oa_ctx *sub : c->subs) { double v = 0.0; const int rc = oa_get_stat(sub, what, &v); if (rc) return rc; *value += v; }
            return OA_OK;
        }
        if (!c->surface || !c->tri_ring_ok || !c->d_tri_ring || !c->d_prev || !c->d_state || c->ns <= 0 || !c->seeded) return OA_OK;
#if defined(OA_EXPERIMENTS)
        int rc = use_device(c);
        if (rc) return rc;
        DevTmp<unsigned long long> d_n;
        HIPCHK(d_n.alloc(16));
        HIPCHK(hipMemsetAsync(d_n, 0, 16 * sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(oa::k_tri_ring_count, dim3((unsigned)((c->ns + 255) / 256)), dim3(256), 0, c->stream, (const oa::DevState *)c->d_state,
                           (const float4 *)c->d_src4, c->ns, c->tgp.scale, (const float4 *)c->d_tri9, (const int *)c->d_prev, d_n.p,
                           c->debug ? d_n.p : (unsigned long long *)nullptr);
        HIPCHK(hipGetLastError());
        unsigned long long n[16];
        if ((rc = read_small(c, n, d_n, sizeof(n)))) return rc;
        *value = (double)n[0];
        if (c->debug)
            fprintf(stderr, "[oa] tri ring: %llu of %d queries settled by seed + neighbours; the others: %llu no seed, %llu seeds that never accept, %llu not certified, "
                            "distance + margins within 1x / 2x / 4x / 8x of the radius %llu / %llu / %llu / %llu, farther %llu\n",
                    n[0], c->ns, n[1], n[2], n[3], n[4], n[5], n[6], n[7], n[8]);
#endif
        return OA_OK;
    }
    if (what == OA_STAT_EXCHANGE_US) {                                  // the slowest device's mean wait for the world's sums (one GPU: 0)
        *value = c->last_exchange_us;
        for (oa_ctx *sub : c->subs) *value = std::max(*value, sub->last_exchange_us);
        return OA_OK;
    }
    if (what == OA_STAT_BRUTE_QUEUE_WGS) { *value = (double)(c->subs.empty() ? c : c->subs[0])->last_queue_wgs; return OA_OK; }
    if (what == OA_STAT_SEARCH_CLOCK_MHZ) {                             // (the host mirror holds the device state of the last oa_run / oa_run_end)
        const oa_ctx *s = c->subs.empty() ? c : c->subs[0];
        const unsigned long long cyc = s->h_state.search_clk[0], ticks = s->h_state.search_clk[1];
        *value = ticks > 0 ? (double)cyc / (double)ticks * s->wall_clock_khz * 1e-3 : 0.0;
        return OA_OK;
    }
    if (what == OA_STAT_RCCL_FALLBACKS) { *value = c->xch && !c->parent ? (double)c->xch->rccl_fallbacks : 0.0; return OA_OK; }
    if (what == OA_STAT_RCCL_RANKS_LAST) { *value = c->xch && !c->parent ? (double)c->xch->rccl_ranks_last : 0.0; return OA_OK; }
    if (what == OA_STAT_WATCHDOG_ABORTS) { *value = c->xch && !c->parent ? (double)c->xch->watchdog_aborts : 0.0; return OA_OK; }
    if (what == OA_STAT_EXCHANGE || what == OA_STAT_RCCL_RANKS || what == OA_STAT_ENQUEUE_US || what == OA_STAT_HOST_THREADS) {
        if (c->subs.empty()) { *value = what == OA_STAT_EXCHANGE ? -1.0 : (what == OA_STAT_HOST_THREADS ? 1.0 : 0.0); return OA_OK; }
        if (what == OA_STAT_HOST_THREADS) { *value = (double)std::max<size_t>(1, c->groups.size()); return OA_OK; }
        if (what == OA_STAT_ENQUEUE_US) {
            double us = 0.0;
            for (oa_ctx *sub : c->subs) if (sub->enq_iters > 0) us = std::max(us, sub->enq_ns / (double)sub->enq_iters * 1e-3);
            *value = us;
            return OA_OK;
        }
        if (!c->loop_active) { const int rcx = exchange_resolve(c); if (rcx) return rcx; }
        if (what == OA_STAT_EXCHANGE) *value = c->xch->mode == OA_EXCHANGE_RCCL ? 1.0 : (c->xch->device_box ? 2.0 : 0.0);
        else *value = (double)exchange_rccl_ranks(c->xch);
        return OA_OK;
    }
    OA_ROUTE_FIRST(c, oa_get_stat(sub, what, value));
    switch (what) {
    case OA_STAT_GRID_CELLS: *value = c->grid_ok ? (double)c->n_cells : 0.0; return OA_OK;
    case OA_STAT_TRI_GRID_CELLS: *value = c->tri_grid_ok ? (double)c->tgp.n[0] * c->tgp.n[1] * c->tgp.n[2] : 0.0; return OA_OK;
    case OA_STAT_TRI_GRID_ENTRIES: *value = c->tri_grid_ok ? (double)c->n_tri_entries : 0.0; return OA_OK;
    case OA_STAT_N_TRIS: *value = (double)c->n_tris; return OA_OK;
    case OA_STAT_SURFACE: *value = c->surface ? 1.0 : 0.0; return OA_OK;
    case OA_STAT_BRUTE_KERNEL:
#if defined(OA_EXPERIMENTS)
        *value = !(c->filter_ok && c->use_filter) ? 0.0
                 : ((c->nn_mfma && c->d_tfm && c->R == 4 && c->tile_groups == oa::FTILE_GROUPS) ? 2.0 : ((c->nn_sort && c->d_tfs) ? 3.0 : 1.0));
#else
        *value = (c->filter_ok && c->use_filter && c->nn_sort && c->d_tfs) ? 3.0 : 0.0;       // (launch_nn_impl: the sorted kernel, or the plain one)
#endif
        return OA_OK;
    case OA_STAT_FAST_ITERATIONS: *value = (double)c->fast_iters; return OA_OK;
    case OA_STAT_HANDOVER_ENTRIES: *value = c->h_poll ? (double)c->h_poll[2] : 0.0; return OA_OK;
    case OA_STAT_HANDOVER_WAVE_MAX: *value = c->h_poll ? (double)c->h_poll[3] : 0.0; return OA_OK;
    case OA_STAT_SAFE_RADII: *value = (c->grid_safe && c->safe_ok) ? 1.0 : 0.0; return OA_OK;
    case OA_STAT_TRI_RING: *value = (c->tri_ring && c->tri_ring_ok) ? 1.0 : 0.0; return OA_OK;
    default: return fail(OA_E_BAD_ARG, "oa_get_stat: unknown key %d", what);
    }
}

OA_EXPORT int oa_get_matrix_world(oa_ctx *c, float mx_align[16])
{
    if (!c || !mx_align) return fail(OA_E_BAD_ARG, "null argument");
    OA_ROUTE_FIRST(c, oa_get_matrix_world(sub, mx_align));           // identical on every device
    if (!c->have_mats) return fail(OA_E_STATE, "matrices not set");
    memcpy(mx_align, c->h_state.mx1, sizeof(float) * 16);
    return OA_OK;
}

OA_EXPORT int64_t oa_num_selected(oa_ctx *c)
{
    if (!c) return 0;
    if (c->subs.empty()) return c->ns;
    int64_t n = 0;
    for (oa_ctx *sub : c->subs) n += sub->ns;                       // all shards
    return n;
}

OA_EXPORT int oa_get_pivot(oa_ctx *c, double pivot[3])
{
    if (!c || !pivot) return fail(OA_E_BAD_ARG, "null argument");
    OA_ROUTE_FIRST(c, oa_get_pivot(sub, pivot));
    for (int k = 0; k < 3; ++k) pivot[k] = c->pivot[k];
    return OA_OK;
}

// search time of every iteration of the last oa_run / oa_run_end, in ms (hipEvent pairs around the brute-force launches, GPU-side
// stamps otherwise): what oa_report::nn_ms_total sums.  A multi-device context answers for its first device.
OA_EXPORT int oa_get_search_ms(oa_ctx *c, int32_t max_n, double *ms)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    OA_ROUTE_FIRST(c, oa_get_search_ms(sub, max_n, ms));
    const int n = std::min((int)c->h_search_ms.size(), (int)std::max(0, max_n));
    for (int i = 0; i < n && ms; ++i) ms[i] = c->h_search_ms[(size_t)i];
    return n;
}

// What the chip's vector ALUs issue RIGHT NOW: burns of dependent-free v_add_f32 / v_min3_f32 chains on every SIMD for ~target_ms, timed
// with hipEvents and with the shader clock / the constant-rate clock read by the same wave at both ends.  bench.py runs it next to
// the headline's timed loop: the brute-force search is bound by VALU issue, the clock under that load is 1.8-2.1 GHz rather
// than the nominal 2.4 and moves from box to box -- with this figure beside it a 50 vs 58 ms launch explains itself.
// out[0] = T lane-ops/s of v_add_f32 (two register sources: the issue rate -- 32 lanes per SIMD and clock), out[1] = shader clock
// (MHz) during that burn, out[2] = duration of both burns (ms), out[3] = T lane-ops/s of v_min3_f32 (the half-rate class the
// search's minimum tree is made of: ~0.55 of out[0])
OA_EXPORT int oa_measure_valu_ceiling(oa_ctx *c, double target_ms, double out[4])
{
    if (!c || !out) return fail(OA_E_BAD_ARG, "oa_measure_valu_ceiling: null argument");
    OA_ROUTE_FIRST(c, oa_measure_valu_ceiling(sub, target_ms, out));
    int rc = use_device(c);
    if (rc) return rc;
    const int wgs_per_cu = 8;                                        // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    const unsigned blocks = (unsigned)(c->n_cu * wgs_per_cu);
    // ~2.7 cycles per wave-instruction and SIMD measured (tools/valu_microbench.hip): iterations for the time asked for
    const double per_iter_s = 8.0 * oa::VALU_BURN_CHAINS * 2.7 / 2.0e9;
    const int iters = ((int)std::max(64.0, std::min(4.0e6, std::max(0.05, std::min(target_ms, 200.0)) * 1e-3 / per_iter_s)) + 7) & ~7;   // (halves are multiples of 4)
    DevTmp<float> d_sink;
    DevTmp<unsigned long long> d_clk;
    HIPCHK(d_sink.alloc((size_t)blocks * 256));
    HIPCHK(d_clk.alloc(4));
    HIPCHK(hipMemsetAsync(d_clk, 0, 4 * sizeof(unsigned long long), c->stream));
    if ((rc = ensure_events(c, 2))) return rc;
    // (a short first launch loads the code object and wakes the clocks; then half the time on each instruction)
    hipLaunchKernelGGL(oa::k_valu_burn<1>, dim3(blocks), dim3(256), 0, c->stream, d_sink.p, 1.0000001f, 1e-9f, std::max(64, (iters / 8) & ~3), d_clk.p);
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    hipLaunchKernelGGL(oa::k_valu_burn<0>, dim3(blocks), dim3(256), 0, c->stream, d_sink.p, 1.0000001f, 1e-9f, iters / 2, d_clk.p);
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(oa::k_valu_burn<1>, dim3(blocks), dim3(256), 0, c->stream, d_sink.p, 1.0000001f, 1e-9f, iters / 2, d_clk.p + 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    unsigned long long clk[4];
    if ((rc = read_small(c, clk, d_clk, sizeof(clk)))) return rc;
    float ms_min3 = 0.f, ms_add = 0.f;
    HIPCHK(hipEventElapsedTime(&ms_min3, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms_add, c->ev[2], c->ev[3]));
    const double laneops = (double)blocks * 256.0 * (double)oa::VALU_BURN_CHAINS * (double)(iters / 2);
    out[0] = ms_add > 0.f ? laneops / ((double)ms_add * 1e-3) / 1e12 : 0.0;       // v_add_f32: the issue rate
    out[1] = clk[3] > 0 ? (double)clk[2] / (double)clk[3] * c->wall_clock_khz * 1e-3 : 0.0;
    out[2] = (double)(ms_min3 + ms_add);
    out[3] = ms_min3 > 0.f ? laneops / ((double)ms_min3 * 1e-3) / 1e12 : 0.0;     // v_min3_f32: the half-rate class
    return OA_OK;
}

// ================================================================================================
// correspondence search alone
// ================================================================================================
namespace {
// copies the host mirror (matrices, pivot) to the device with a neutral loop state
int push_state_for_oneshot(oa_ctx *c, double thresh, bool cutoff)
{
    oa_settings st{};
    st.iters = 1; st.use_target = 1; st.with_scale = 0; st.early_exit = 0; st.thresh = thresh; st.target_d = 0.0;
    init_loop_state(c, &st, 1, cutoff);
    HIPCHK(hipStreamSynchronize(c->stream));                    // the staging copy may still be in flight
    int rc = push_state(c);
    if (rc) return rc;
    c->loop_active = false;
    return OA_OK;
}
}  // namespace

OA_EXPORT int oa_nn_search(oa_ctx *c, int64_t *idx, float *d2, double *kernel_ms)
{
    if (c && !c->subs.empty()) {
        // multi-device: every child searches its shard; the answers go back to the caller's (vlist) order through the
        // shard membership each child kept.  kernel_ms = the slowest child (the children run one after the other here:
        // this entry point returns host arrays, it is not the loop).
        if (c->loop_active) multi_abort(c);                          // re-stages every child's device state: the sequence ends
        c->loop_active = false;
        if (kernel_ms) *kernel_ms = 0.0;
        std::vector<int64_t> ic;
        std::vector<float> dc;
        for (oa_ctx *sub : c->subs) {
            const size_t n = (size_t)sub->ns;
            if (idx) ic.resize(std::max<size_t>(1, n));
            if (d2) dc.resize(std::max<size_t>(1, n));
            double ms = 0.0;
            int rc = oa_nn_search(sub, idx ? ic.data() : nullptr, d2 ? dc.data() : nullptr, &ms);
            if (rc) return rc;
            if (kernel_ms && ms > *kernel_ms) *kernel_ms = ms;
            for (size_t k = 0; k < n; ++k) {
                const long long g = sub->h_members.empty() ? sub->shard_begin + (long long)k : (long long)sub->h_members[k];
                if (idx) idx[g] = ic[k];
                if (d2) d2[g] = dc[k];
            }
        }
        return OA_OK;
    }
    int rc = check_ready(c);
    if (rc) return rc;
    if ((rc = use_device(c))) return rc;
    if ((rc = ensure_common(c))) return rc;
    if ((rc = ensure_events(c, 1))) return rc;
    if ((rc = push_state_for_oneshot(c, 1.0, false))) return rc;
    if (kernel_ms) *kernel_ms = 0.0;
    if (c->ns <= 0) return OA_OK;
    DevTmp<long long> d_idx;
    DevTmp<float> d_d2;
    if (idx) HIPCHK(d_idx.alloc((size_t)c->ns));
    if (d2) HIPCHK(d_d2.alloc((size_t)c->ns));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if ((rc = launch_nn(c))) return rc;
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    hipLaunchKernelGGL(oa::k_decode_keys, dim3((c->ns_pad + 255) / 256), dim3(256), 0, c->stream, c->d_keys,
                       c->ns_pad, c->ns, (const int *)c->d_perm, d_idx.p, d_d2.p);
    HIPCHK(hipGetLastError());
    if (idx) HIPCHK(hipMemcpyAsync(idx, d_idx, sizeof(long long) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
    if (d2) HIPCHK(hipMemcpyAsync(d2, d_d2, sizeof(float) * (size_t)c->ns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (kernel_ms) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        *kernel_ms = ms;
    }
    return OA_OK;
}

// ================================================================================================
// contract 1: make_pairs
// ================================================================================================
OA_EXPORT int oa_make_pairs(oa_ctx *c, double thresh, int calc_stats, double *A, double *B, int64_t cap, int64_t *K,
                            double dstats[2])
{
    if (c && !c->subs.empty()) {
        // multi-device: every child pairs its shard (its pairs come out in ascending selection position, with that
        // position); the shards are merged back into vlist order on the host.  Statistics: the children's two-pass
        // means and population deviations combine exactly (K-weighted mean; variance = within + between).
        if (!K || cap < 0 || (cap > 0 && (!A || !B))) return fail(OA_E_BAD_ARG, "oa_make_pairs: bad output arguments");
        if (!(thresh > 0.0)) return fail(OA_E_BAD_THRESH, "thresh must be > 0 (the reference's make_pairs returns None)");
        if (c->loop_active) multi_abort(c);                          // re-stages every child's device state: the sequence ends
        c->loop_active = false;
        *K = 0;
        if (dstats) { dstats[0] = NAN; dstats[1] = NAN; }
        const size_t n_dev = c->subs.size();
        std::vector<std::vector<double>> Ac(n_dev), Bc(n_dev);
        std::vector<std::vector<int>> Pc(n_dev);
        std::vector<int64_t> Kc(n_dev, 0);
        std::vector<double> mean(n_dev, 0.0), dev(n_dev, 0.0);
        int64_t total = 0;
        for (size_t r = 0; r < n_dev; ++r) {
            oa_ctx *sub = c->subs[r];
            const int64_t capc = std::max(1, sub->ns);
            Ac[r].resize(3 * (size_t)capc); Bc[r].resize(3 * (size_t)capc);
            double st[2] = { NAN, NAN };
            int rc = oa_make_pairs(sub, thresh, calc_stats, Ac[r].data(), Bc[r].data(), capc, &Kc[r], st);
            if (rc) return rc;
            mean[r] = st[0]; dev[r] = st[1];
            Pc[r].resize((size_t)std::max<int64_t>(1, Kc[r]));
            if (Kc[r] > 0) {
                if ((rc = use_device(sub))) return rc;
                HIPCHK(hipMemcpyAsync(Pc[r].data(), sub->d_pos, sizeof(int) * (size_t)Kc[r], hipMemcpyDeviceToHost, sub->stream));
                HIPCHK(hipStreamSynchronize(sub->stream));
            }
            total += Kc[r];
        }
        if (total > cap) return fail(OA_E_CAPACITY, "oa_make_pairs: %lld pairs but capacity %lld", (long long)total, (long long)cap);
        std::vector<int64_t> head(n_dev, 0);
        for (int64_t k = 0; k < total; ++k) {                       // n_dev-way merge on the selection position
            size_t best = n_dev;
            for (size_t r = 0; r < n_dev; ++r)
                if (head[r] < Kc[r] && (best == n_dev || Pc[r][(size_t)head[r]] < Pc[best][(size_t)head[best]])) best = r;
            const int64_t h = head[best]++, capc = std::max(1, c->subs[best]->ns);
            for (int a = 0; a < 3; ++a) {
                A[(size_t)a * cap + k] = Ac[best][(size_t)a * capc + h];
                B[(size_t)a * cap + k] = Bc[best][(size_t)a * capc + h];
            }
        }
        *K = total;
        if (calc_stats && dstats && total > 0) {
            double m = 0.0;
            for (size_t r = 0; r < n_dev; ++r) if (Kc[r] > 0) m += (double)Kc[r] * mean[r];
            m /= (double)total;
            double var = 0.0;
            for (size_t r = 0; r < n_dev; ++r)
                if (Kc[r] > 0) var += (double)Kc[r] * (dev[r] * dev[r] + (mean[r] - m) * (mean[r] - m));
            dstats[0] = m;
            dstats[1] = sqrt(var / (double)total);
        }
        return OA_OK;
    }
    int rc = check_ready(c);
    if (rc) return rc;
    if (!K || cap < 0 || (cap > 0 && (!A || !B))) return fail(OA_E_BAD_ARG, "oa_make_pairs: bad output arguments");
    if (!(thresh > 0.0)) return fail(OA_E_BAD_THRESH, "thresh must be > 0 (the reference's make_pairs returns None)");
    if ((rc = use_device(c))) return rc;
    if ((rc = ensure_common(c))) return rc;
    *K = 0;
    if (dstats) { dstats[0] = NAN; dstats[1] = NAN; }
    if (c->ns == 0) return OA_OK;
    const int n_blocks = (c->ns + 255) / 256;
    if (c->emit_cap < c->ns) {
        dev_free(c->d_valid); dev_free(c->d_b); dev_free(c->d_dist); dev_free(c->d_counts); dev_free(c->d_offsets);
        dev_free(c->d_A); dev_free(c->d_B);
        HIPCHK(dev_malloc(&c->d_valid, (size_t)c->ns));
        HIPCHK(dev_malloc(&c->d_b, sizeof(float) * 3 * (size_t)c->ns));
        HIPCHK(dev_malloc(&c->d_dist, sizeof(double) * (size_t)c->ns));
        HIPCHK(dev_malloc(&c->d_counts, sizeof(int) * (size_t)n_blocks));
        HIPCHK(dev_malloc(&c->d_offsets, sizeof(long long) * (size_t)(n_blocks + 1)));
        HIPCHK(dev_malloc(&c->d_A, sizeof(double) * 3 * (size_t)c->ns));
        HIPCHK(dev_malloc(&c->d_B, sizeof(double) * 3 * (size_t)c->ns));
        dev_free(c->d_pos);
        if (c->parent) HIPCHK(dev_malloc(&c->d_pos, sizeof(int) * (size_t)c->ns));
        c->emit_cap = c->ns;
    }
    c->d_pivot0 = 0.0;
    if ((rc = push_state_for_oneshot(c, thresh, true))) return rc;
    if ((rc = launch_nn(c))) return rc;
    if ((rc = launch_accumulate(c, true, nullptr, nullptr))) return rc;
    if ((rc = launch_reduce(c, c->d_sums, oa::RowSel{ c->acc_blocks, 0, 0 }, false))) return rc;   // (emitting accumulation: k_pair_accumulate<true>)
    if (calc_stats) {
        // np.std is two-pass; redo the (cheap) accumulation around the mean of the first pass so that the
        // population std is accurate even when it is tiny compared with the mean
        double s1[oa::NSUMS];
        { int rcr = read_small(c, s1, c->d_sums, sizeof s1); if (rcr) return rcr; }
        if (s1[oa::S_K] > 0.0) {
            c->d_pivot0 = s1[oa::S_D] / s1[oa::S_K];
            if ((rc = push_state_for_oneshot(c, thresh, true))) { c->d_pivot0 = 0.0; return rc; }
            rc = launch_nn(c);
            if (!rc) rc = launch_accumulate(c, true, nullptr, nullptr);
            if (!rc) rc = launch_reduce(c, c->d_sums, oa::RowSel{ c->acc_blocks, 0, 0 }, false);
            if (rc) { c->d_pivot0 = 0.0; return rc; }
        }
    }
    hipLaunchKernelGGL(oa::k_block_counts, dim3(n_blocks), dim3(256), 0, c->stream, c->d_valid, c->ns, c->d_counts);
    hipLaunchKernelGGL(oa::k_scan_counts, dim3(1), dim3(1024), 0, c->stream, c->d_counts, n_blocks, c->d_offsets);
    hipLaunchKernelGGL(oa::k_scatter_pairs, dim3(n_blocks), dim3(256), 0, c->stream, c->d_valid, c->ns,
                       c->d_perm ? c->d_src4o : c->d_src4,
                       c->d_b, c->d_offsets, (long long)c->ns, c->d_A, c->d_B, (const int *)c->d_members, c->shard_begin,
                       c->parent ? c->d_pos : (int *)nullptr);
    HIPCHK(hipGetLastError());
    double sums[oa::NSUMS];
    long long total = 0;
    HIPCHK(hipMemcpyAsync(sums, c->d_sums, sizeof sums, hipMemcpyDeviceToHost, c->stream));
    { int rcr = read_small(c, &total, c->d_offsets + n_blocks, sizeof total); if (rcr) return rcr; }
    if (total > cap) return fail(OA_E_CAPACITY, "oa_make_pairs: %lld pairs but capacity %lld", total, (long long)cap);
    for (int a = 0; a < 3 && total > 0; ++a) {
        HIPCHK(hipMemcpyAsync(A + (size_t)a * cap, c->d_A + (size_t)a * c->ns, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(B + (size_t)a * cap, c->d_B + (size_t)a * c->ns, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    *K = total;
    if (calc_stats && dstats && total > 0) {
        const double kk = sums[oa::S_K];
        const double mean_dd = sums[oa::S_D] / kk;                  // relative to the pivot of the second pass
        double var = sums[oa::S_DD] / kk - mean_dd * mean_dd;
        if (var < 0.0) var = 0.0;
        dstats[0] = mean_dd + c->d_pivot0;                          // np.mean(dists)  (general.py:324)
        dstats[1] = sqrt(var);                                      // np.std(dists)   (general.py:325)
    }
    c->d_pivot0 = 0.0;
    return OA_OK;
}

// ================================================================================================
// contract 2: affine_matrix_from_points (shear=False, usesvd=True)
// ================================================================================================
namespace {
int solve_on_device(oa_ctx *c, const double *d_sums, const double pv[3], int with_scale, double M[16])
{
    hipLaunchKernelGGL(oa::k_solve_only, dim3(1), dim3(64), 0, c->stream, d_sums, pv[0], pv[1], pv[2], with_scale, c->d_solve);
    HIPCHK(hipGetLastError());
    double out[17];
    { int rcr = read_small(c, out, c->d_solve, sizeof out); if (rcr) return rcr; }
    if (out[16] != 1.0) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");
    memcpy(M, out, sizeof(double) * 16);
    return OA_OK;
}
}  // namespace

OA_EXPORT int oa_kabsch(oa_ctx *c, const double *A, const double *B, int64_t K, int64_t ld, int with_scale, double M[16])
{
    if (!c || !M) return fail(OA_E_BAD_ARG, "oa_kabsch: null argument");
    OA_ROUTE_FIRST(c, oa_kabsch(sub, A, B, K, ld, with_scale, M));
    if (K < 3) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");   // general.py:150-157
    if (!A || !B || ld < K) return fail(OA_E_BAD_ARG, "oa_kabsch: bad arrays");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = ensure_common(c))) return rc;
    DevTmp<double> dA, dB;
    HIPCHK(dA.alloc(3 * (size_t)K));
    HIPCHK(dB.alloc(3 * (size_t)K));
    for (int a = 0; a < 3; ++a) {
        HIPCHK(hipMemcpyAsync(dA.p + (size_t)a * K, A + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dB.p + (size_t)a * K, B + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
    }
    const double pv[3] = { A[0], A[ld], A[2 * ld] };
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(oa::ACC_MAX_BLOCKS, (K + oa::ACC_THREADS - 1) / oa::ACC_THREADS));
    hipLaunchKernelGGL(oa::k_accumulate_pairs, dim3(blocks), dim3(oa::ACC_THREADS), 0, c->stream, (const double *)dA.p,
                       (const double *)dB.p, (long long)K, (long long)K, pv[0], pv[1], pv[2], c->d_partials);
    hipLaunchKernelGGL(oa::k_reduce_partials, dim3(1), dim3(oa::RED_THREADS), 0, c->stream, (const oa::DevState *)nullptr,
                       (const double *)c->d_partials, oa::RowSel{ blocks, 0, 0 }, c->d_sums, (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    rc = solve_on_device(c, c->d_sums, pv, with_scale, M);      // synchronises the stream
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}

// the reference's full signature: any ndims (2..8 on the fixed-size kernels, up to 64 through a device workspace), shear (full
// affine) or rigid / similarity
OA_EXPORT int oa_affine_from_points(oa_ctx *c, const double *v0, const double *v1, int ndims, int64_t K, int64_t ld,
                                    int shear, int with_scale, double *M)
{
    if (!c || !M) return fail(OA_E_BAD_ARG, "oa_affine_from_points: null argument");
    OA_ROUTE_FIRST(c, oa_affine_from_points(sub, v0, v1, ndims, K, ld, shear, with_scale, M));
    if (ndims < 2 || K < ndims) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");   // general.py:150-157
    if (ndims > oa::AFF_MAXD_HEAP) return fail(OA_E_BAD_ARG, "oa_affine_from_points: ndims %d > %d", ndims, oa::AFF_MAXD_HEAP);
    if (!v0 || !v1 || ld < K) return fail(OA_E_BAD_ARG, "oa_affine_from_points: bad arrays");
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = ensure_common(c))) return rc;
    const int n = ndims, w = n + 1;
    if (n > oa::AFF_MAXD) {
        // more than 8 dimensions (no caller in the add-on; the reference's signature takes any): one workgroup per row and per
        // Gram entry, the solve with its matrices in a device workspace
        const int D = n <= 16 ? 16 : (n <= 32 ? 32 : 64), m = 2 * n;
        DevTmp<double> d0, d1, d_cs, d_gram, d_out, d_ws;
        HIPCHK(d0.alloc((size_t)n * K)); HIPCHK(d1.alloc((size_t)n * K));
        HIPCHK(d_cs.alloc((size_t)2 * D)); HIPCHK(d_gram.alloc((size_t)m * m)); HIPCHK(d_out.alloc((size_t)w * w + 1));
        HIPCHK(d_ws.alloc(oa::aff_ws_doubles(D)));
        for (int a = 0; a < n; ++a) {
            HIPCHK(hipMemcpyAsync(d0.p + (size_t)a * K, v0 + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(d1.p + (size_t)a * K, v1 + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
        }
        hipLaunchKernelGGL(oa::k_affine_rowsum_any, dim3((unsigned)m), dim3(256), 0, c->stream, (const double *)d0.p, (const double *)d1.p, n, D,
                           (long long)K, (long long)K, d_cs.p);
        hipLaunchKernelGGL(oa::k_affine_gram_any, dim3((unsigned)(m * m)), dim3(256), 0, c->stream, (const double *)d0.p, (const double *)d1.p, n, D,
                           (long long)K, (long long)K, (const double *)d_cs.p, d_gram.p);
#define OA_AFF_SOLVE(DD) hipLaunchKernelGGL((oa::k_affine_solve<DD, true>), dim3(1), dim3(64), 0, c->stream, (const double *)d_cs.p, (const double *)d_gram.p, n, \
                                            (long long)K, shear ? 1 : 0, with_scale ? 1 : 0, d_out.p, d_ws.p)
        if (D == 16) OA_AFF_SOLVE(16); else if (D == 32) OA_AFF_SOLVE(32); else OA_AFF_SOLVE(64);
#undef OA_AFF_SOLVE
        HIPCHK(hipGetLastError());
        std::vector<double> out((size_t)w * w + 1);
        { int rcr = read_small(c, out.data(), d_out, sizeof(double) * out.size()); if (rcr) return rcr; }
        if (out[(size_t)w * w] != 1.0) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");
        memcpy(M, out.data(), sizeof(double) * (size_t)w * w);
        return OA_OK;
    }
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(512, (K + 1023) / 1024));
    const long long cols_per_block = ((K + blocks - 1) / blocks + oa::AFF_TILE - 1) / oa::AFF_TILE * oa::AFF_TILE;
    const int gblocks = (int)((K + cols_per_block - 1) / cols_per_block);
    DevTmp<double> d0, d1, p1, p2, d_cs, d_gram, d_out;
    HIPCHK(d0.alloc((size_t)n * K)); HIPCHK(d1.alloc((size_t)n * K));
    HIPCHK(p1.alloc((size_t)blocks * oa::AFF_M2)); HIPCHK(p2.alloc((size_t)gblocks * oa::AFF_M2 * oa::AFF_M2));
    HIPCHK(d_cs.alloc(oa::AFF_M2)); HIPCHK(d_gram.alloc(oa::AFF_M2 * oa::AFF_M2)); HIPCHK(d_out.alloc((size_t)w * w + 1));
    for (int a = 0; a < n; ++a) {
        HIPCHK(hipMemcpyAsync(d0.p + (size_t)a * K, v0 + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d1.p + (size_t)a * K, v1 + (size_t)a * ld, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(oa::k_affine_colsums, dim3(blocks), dim3(256), 0, c->stream, (const double *)d0.p, (const double *)d1.p, n,
                       (long long)K, (long long)K, p1.p);
    hipLaunchKernelGGL(oa::k_affine_reduce, dim3(1), dim3(64), 0, c->stream, (const double *)p1.p, blocks, oa::AFF_M2, d_cs.p);
    hipLaunchKernelGGL(oa::k_affine_gram, dim3(gblocks), dim3(256), 0, c->stream, (const double *)d0.p, (const double *)d1.p, n,
                       (long long)K, (long long)K, (const double *)d_cs.p, cols_per_block, p2.p);
    hipLaunchKernelGGL(oa::k_affine_reduce, dim3(1), dim3(256), 0, c->stream, (const double *)p2.p, gblocks, oa::AFF_M2 * oa::AFF_M2, d_gram.p);
    hipLaunchKernelGGL((oa::k_affine_solve<oa::AFF_MAXD, false>), dim3(1), dim3(64), 0, c->stream, (const double *)d_cs.p, (const double *)d_gram.p, n,
                       (long long)K, shear ? 1 : 0, with_scale ? 1 : 0, d_out.p, (double *)nullptr);
    HIPCHK(hipGetLastError());
    std::vector<double> out((size_t)w * w + 1);
    { int rcr = read_small(c, out.data(), d_out, sizeof(double) * out.size()); if (rcr) return rcr; }
    if (out[(size_t)w * w] != 1.0) return fail(OA_E_TOO_FEW_PAIRS, "input arrays are of wrong shape or type");
    memcpy(M, out.data(), sizeof(double) * (size_t)w * w);
    return OA_OK;
}

OA_EXPORT int oa_kabsch_from_sums(oa_ctx *c, const double sums[OA_NSUMS], const double pivot[3], int with_scale,
                                  double M[16])
{
    if (!c || !sums || !M) return fail(OA_E_BAD_ARG, "oa_kabsch_from_sums: null argument");
    OA_ROUTE_FIRST(c, oa_kabsch_from_sums(sub, sums, pivot, with_scale, M));
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = ensure_common(c))) return rc;
    const double zero[3] = { 0, 0, 0 };
    HIPCHK(hipMemcpyAsync(c->d_sums, sums, sizeof(double) * oa::NSUMS, hipMemcpyHostToDevice, c->stream));
    return solve_on_device(c, c->d_sums, pivot ? pivot : zero, with_scale, M);
}

// ================================================================================================
// the loop
// ================================================================================================
OA_EXPORT int oa_run_begin(oa_ctx *c, const oa_settings *st)
{
    if (!c || !st) return fail(OA_E_BAD_ARG, "oa_run_begin: null argument");
    OA_NOT_MULTI(c, "oa_run_begin (the split-phase loop is for one process per GPU)");
    int rc = begin_loop(c, st, st->iters);
    if (rc) return rc;
    if ((rc = ensure_events(c, std::max(1, st->iters)))) return rc;
    HIPCHK(hipEventRecord(c->ev_loop0, c->stream));
    return OA_OK;
}

OA_EXPORT int oa_iter_partial(oa_ctx *c, double *d_sums)
{
    if (!c || !d_sums) return fail(OA_E_BAD_ARG, "oa_iter_partial: null argument");
    OA_NOT_MULTI(c, "oa_iter_partial");
    if (!c->loop_active) return fail(OA_E_STATE, "oa_iter_partial outside oa_run_begin/oa_run_end");
    int rc = use_device(c);
    if (rc) return rc;
    return iter_partial(c, d_sums, true);
}

OA_EXPORT int oa_iter_finish(oa_ctx *c, const double *d_sums)
{
    if (!c || !d_sums) return fail(OA_E_BAD_ARG, "oa_iter_finish: null argument");
    OA_NOT_MULTI(c, "oa_iter_finish");
    if (!c->loop_active) return fail(OA_E_STATE, "oa_iter_finish outside oa_run_begin/oa_run_end");
    int rc = use_device(c);
    if (rc) return rc;
    return iter_finish(c, d_sums);
}

OA_EXPORT int oa_run_end(oa_ctx *c, oa_report *rep)
{
    if (!c || !rep) return fail(OA_E_BAD_ARG, "oa_run_end: null argument");
    OA_NOT_MULTI(c, "oa_run_end");
    int rc = use_device(c);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_loop1, c->stream));
    if ((rc = fill_report(c, rep))) return rc;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_loop0, c->ev_loop1) == hipSuccess) rep->loop_ms = ms;
    c->loop_active = false;
    return status_error(rep->status);
}

OA_EXPORT int oa_run(oa_ctx *c, const oa_settings *st, oa_report *rep)
{
    if (!c || !st || !rep) return fail(OA_E_BAD_ARG, "oa_run: null argument");
    if (!c->subs.empty()) return multi_run(c, st, rep);
    int rc = oa_run_begin(c, st);
    if (rc) return rc;
    // The whole loop is enqueued ahead of the GPU.  With early exit on, iterations after convergence would still cost
    // three empty launches each (the kernels see DevState.halt and return) -- for a small mesh that converges in 7 of 50
    // iterations, more than the real work.  k_solve_update therefore mirrors the halt flag into a pinned host word,
    // and the host looks at it before it enqueues the next iteration: no copies, nothing added to the stream -- a
    // stale 0 only means a few more empty launches.
    // The host also stays at most `lag` iterations ahead of the GPU (the solve kernel mirrors its iteration counter
    // next to the halt flag): enqueuing is much faster than executing, and a host that is 40 iterations ahead learns
    // about the halt too late to save anything.  Two iterations are always queued, so the GPU never idles.
    // The adaptive grid path (grid_fast_now) needs recent news from the device as well: then the host stays close even
    // when the loop cannot end early.
    bool poll = c->h_poll && env_int("OA_RUN_POLL", 1) && (st->early_exit || (search_plan(c) == PLAN_GRID && c->grid_path == 0));
    const int lag = 2;
    volatile int32_t *progress = c->h_poll;
    for (int it = 0; it < st->iters; ++it) {
        if (poll) {
            const auto t_wait = std::chrono::steady_clock::now();
            while (!progress[0] && progress[1] < it - lag) {
                if (std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(5)) { poll = false; break; }   // never hang on a sick GPU; once is enough
                std::this_thread::yield();
            }
            if (progress[0]) break;
        }
        if ((rc = iter_fused(c, true))) { (void)hipStreamSynchronize(c->stream); c->loop_active = false; return rc; }
    }
    return oa_run_end(c, rep);
}

namespace {
bool same_loop_settings(const oa_settings &a, const oa_settings &b)
{
    return a.use_target == b.use_target && a.with_scale == b.with_scale && a.thresh == b.thresh && a.target_d == b.target_d;
}
}  // namespace

// A modal sequence lasts while oa_iterate is called with the same thresh / target_d / use_target / with_scale.  Changed
// settings, or any call that re-stages the device state in between (oa_set_matrices, oa_make_pairs, oa_nn_search,
// oa_run, a new upload), end it: the next oa_iterate starts a new sequence (n = 0, fresh convergence ring) from the
// current matrix_world.
OA_EXPORT int oa_iterate(oa_ctx *c, const oa_settings *st, double M_step[16], double stats[6])
{
    if (!c || !st) return fail(OA_E_BAD_ARG, "oa_iterate: null argument");
    const bool multi = !c->subs.empty();
    int rc;
    if (c->loop_active && !(c->iterate_mode && same_loop_settings(*st, c->settings))) {
        if (multi) multi_abort(c);
        else { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); c->loop_active = false; }
        if (!multi && (rc = fetch_state(c))) return rc;              // the pose reached so far is where the new sequence starts
        if (multi) for (oa_ctx *sub : c->subs) { if ((rc = use_device(sub))) return rc; if ((rc = fetch_state(sub))) return rc; }
    }
    if (!c->loop_active) {
        if ((rc = multi ? multi_begin(c, st, ITERATE_OPEN) : begin_loop(c, st, ITERATE_OPEN))) return rc;
        c->iterate_mode = true;
        for (oa_ctx *sub : c->subs) sub->iterate_mode = true;       // oa_get_history reads the first child's ring
    }
    oa_ctx *c0 = multi ? c->subs[0] : c;
    if (multi) {
        for (oa_ctx *sub : c->subs) sub->ev_used = 0;
        const bool rccl = c->xch->mode == OA_EXCHANGE_RCCL;             // then the wait is the watchdog's (multi_wait), not fetch_state's
        rc = multi_for_groups(c, [c, rccl](size_t g) -> int {            // every GPU's host thread: one iteration, then the state
            int rg = multi_iteration_group(c, c->groups[g], false);
            for (int i : c->groups[g]) { if (rg || rccl) break; oa_ctx *sub = c->subs[(size_t)i]; if (!(rg = use_device(sub))) rg = fetch_state(sub); }
            return rg;
        });
        if (rc) { const std::string keep = g_err; multi_abort(c, true); g_err = keep; return rc; }
        if (rccl) {
            if ((rc = multi_wait(c))) { for (oa_ctx *sub : c->subs) sub->loop_active = false; c->loop_active = false; return rc; }
            for (oa_ctx *sub : c->subs) { if ((rc = use_device(sub))) return rc; if ((rc = fetch_state(sub))) return rc; }
        }
    } else {
        if ((rc = use_device(c))) return rc;
        c->ev_used = 0;
        if ((rc = iter_fused(c, false))) { (void)hipStreamSynchronize(c->stream); c->loop_active = false; return rc; }
        if ((rc = fetch_state(c))) return rc;
    }
    const oa::DevState &s = c0->h_state;
    if (s.status != 0) {
        if (multi) multi_abort(c); else c->loop_active = false;
        if ((rc = status_error(s.status))) return rc;
        return fail(OA_E_HIP, "oa_iterate: device status %d", s.status);
    }
    if (s.n <= 0) return fail(OA_E_STATE, "oa_iterate: loop already halted");
    oa::StepRecord r;
    memcpy(&r, c0->h_hist_map + ((s.n - 1) % c0->max_records), sizeof r);           // fetch_state synchronised the stream
    if (M_step) memcpy(M_step, r.M, sizeof r.M);
    if (stats) {
        stats[0] = r.K; stats[1] = s.use_target ? r.mean_d : NAN; stats[2] = s.use_target ? r.std_d : NAN;
        stats[3] = r.trans; stats[4] = r.angle; stats[5] = (double)s.converged;
    }
    return OA_OK;
}

OA_EXPORT int oa_get_history(oa_ctx *c, int32_t max_n, double *step_M, float *step_new, int64_t *step_K,
                             double *step_stats, double *step_trans)
{
    if (!c) return fail(OA_E_BAD_ARG, "null context");
    OA_ROUTE_FIRST(c, oa_get_history(sub, max_n, step_M, step_new, step_K, step_stats, step_trans));
    if (use_device(c)) return OA_E_HIP;
    const int total = c->h_state.n, cap = c->max_records;
    const int held = std::min(total, cap);
    const int n = std::min(held, (int)max_n);
    if (n <= 0) return 0;
    std::vector<oa::StepRecord> h_local;
    if (!(c->h_hist_valid && (int)c->h_hist.size() >= held)) {
        h_local.resize((size_t)held);
        if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(OA_E_HIP, "oa_get_history: stream error");
        memcpy(h_local.data(), c->h_hist_map, sizeof(oa::StepRecord) * (size_t)held);
    }
    const std::vector<oa::StepRecord> &h = h_local.empty() ? c->h_hist : h_local;
    // a loop that fits the history: its first n iterations; a ring that wrapped (oa_iterate): the last n, oldest first
    const int first = (total > cap || c->iterate_mode) ? total - n : 0;
    for (int i = 0; i < n; ++i) {
        const oa::StepRecord &r = h[(size_t)((first + i) % cap)];
        if (step_M) memcpy(step_M + 16 * i, r.M, sizeof r.M);
        if (step_new) memcpy(step_new + 16 * i, r.new_mat, sizeof r.new_mat);
        if (step_K) step_K[i] = (int64_t)r.K;
        if (step_stats) { step_stats[2 * i] = r.mean_d; step_stats[2 * i + 1] = r.std_d; }
        if (step_trans) step_trans[i] = r.trans;
    }
    return n;
}
