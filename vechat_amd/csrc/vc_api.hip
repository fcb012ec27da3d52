// libvechat_hip.so: C ABI (include/vechat_hip.h) over the HIP kernels in vc_kernels.h.
//
// Host side of the path: stands where the reference's Polisher::polish (src/polisher.cpp:491-517)
// hands windows to Window::generate_consensus, shaped like its accelerated precedent
// CUDABatchProcessor (src/cuda/cudabatch.cpp:79-270): fill a batch, run it, read statuses back.
// There is no CPU fallback anywhere in this file: without a gfx950 device vc_create fails.
//
// Execution plan.  The batch is cut into chunks of CW windows; a chunk owns one workspace (graphs,
// row records, stored DP matrices, pair lists) and one HIP stream.  Inside a chunk the build loop of
// window.cpp:239-298 runs in lockstep: layer j of every window per iteration
//     k_rows / k_rows_sub -> k_fwd -> k_resolve -> k_tracew -> k_addaln
// then the prune rounds (k_prune_lcc -> k_topo -> [k_fwd, k_tracew]* -> k_addw) and the final local
// alignment.  Several chunks are in flight on separate streams so that the latency-bound single-lane
// kernels (k_tracew, k_addaln, k_rows, k_topo, k_prune_lcc) of one chunk overlap the k_fwd of another.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "vc_kernels.h"
// The persistent build pipeline (vc_pipe.h: built, bit-identical, slower than the lock-step plan -- DESIGN section 10) is an experiment:
// compiled only with -DVC_EXPERIMENTS (VC_EXPERIMENTS=1 python __graft_entry__.py); without it vc_set_pipeline(on) is refused.
#ifdef VC_EXPERIMENTS
#include "vc_pipe.h"
#endif

namespace {

thread_local std::string g_create_error;

enum KClass { KC_AVG = 0, KC_INIT, KC_TOPO, KC_FWD, KC_TRACE, KC_ADDALN, KC_PRUNE, KC_ADDW, KC_FINISH, KC_ROWS, KC_RESOLVE, KC_CONS, KC_PIPE, KC_N };
const char* kClassNames[KC_N] = {"k_avg", "k_init", "k_topo", "k_fwd", "k_trace", "k_addaln", "k_prune_lcc",
                                 "k_addw", "k_finish", "k_rows", "k_resolve", "k_consensus", "k_pipe"};

#ifndef VC_RING
#define VC_RING 8
#endif
constexpr int kRing = VC_RING;      // build phase: DP rows kept in LDS per alignment (the row builders mark rows needed from farther away)
#ifndef VC_RING_PRUNED
#define VC_RING_PRUNED 4
#endif
#ifndef VC_KEPT
#define VC_KEPT 6
#endif
constexpr int kKept = VC_KEPT;      // build phase: slots of the kept-row ring (0: plain ring of kRing rows); see vc_frec_kept
// Batches whose widest class has 32 or more columns per lane: a ring slot is 6 - 8 KB there, six of them left 3 waves on a CU.  Half the
// slots (the row builders hand out as many as the batch says, Batch::kept / ring / ring_pruned): six waves per CU at 64 columns per lane,
// eight at 48; the rows that no longer find their predecessor in the ring read it back from the stored matrix.
constexpr int kKeptWide = 3, kRingWide = 4, kRingPrunedWide = 2;
constexpr int kRingPruned = VC_RING_PRUNED;
// (re-alignment rounds and the final alignment: a pruned graph is nearly a chain, four rows hold its non-adjacent predecessors, and the
// smaller ring lets a fifth / sixth wave onto each SIMD)
constexpr int kMaxStreams = 16;
constexpr uint32_t kArenaSegs = 4;         // segments of the workspace arena (vc_ctx::arena): each a quarter of the budget
constexpr uint32_t kAutoStreamsMany = 8, kAutoStreamsFew = 4, kAutoStreamsFrom = 12288;   // windows from which a batch runs on the larger number of chunk streams
constexpr uint32_t kTraceTabRows = 8188;           // rows covered by k_tracew's first-in-edge table (4 tables x 2 B x 8192 = 64 KB); later rows are walked without speculation
constexpr uint32_t kLdsCap = 160 * 1024 - 1024;   // dynamic LDS a kernel may ask for (160 KB per CU minus room for static __shared__)
constexpr uint32_t kMaxColumns = 4096;     // longest sequence the packed-int16 k_fwd takes (64 lanes x 64 columns); longer ones go to k_fwd_wide
constexpr uint32_t kResolveGrid = 128;   // workgroups of k_resolve (each owns one DFS workspace in HBM)

uint32_t topo_lds_bytes(uint32_t NC, uint32_t EC, uint32_t STK, uint32_t MA) {
    return ((2 * NC + 15) & ~15u) + 4 * EC + 2 * MA * NC + 2 * ((NC + 15) & ~15u) + 2 * STK + 2 * NC + 64;
}

// one chunk's device workspace + stream
struct Work {
    hipStream_t stream = nullptr;
    VcGraph gr[2]{};
    VcDp dp{};
    int* d_wmat = nullptr; int* d_c0w = nullptr;         // k_fwd_wide: [jobs_cap * NC * wcols] tilted int32 scores, [jobs_cap * NC] column 0
    uint32_t* d_hmat = nullptr; int16_t* d_c0 = nullptr; uint8_t* d_resolve_ws = nullptr;
    uint32_t* d_bmat = nullptr; uint32_t* d_band_par = nullptr;     // banded matrix store: tiled band rows (vc_band_job_dwords per job), [jobs_cap * 2] band of a job
    uint32_t* d_redo_list = nullptr; uint32_t* d_redo_n = nullptr;  // jobs whose backtrack left the band (re-run with whole rows)
    uint8_t* d_big_ws = nullptr;        // [CW * big_ws_stride] graph images that do not fit the LDS (k_topo / k_prune_lcc / k_consensus)
    uint32_t* d_job_end = nullptr; uint8_t* d_job_type = nullptr;
    uint16_t* d_tie_rows = nullptr; uint32_t* d_tie_cnt = nullptr; uint32_t* d_tie_list = nullptr; uint32_t* d_tie_n = nullptr;
    uint32_t* d_pairs = nullptr; uint32_t* d_npairs = nullptr;       // build / final: [CW*PC]
    uint32_t* d_rpairs = nullptr; uint32_t* d_rnpairs = nullptr;     // realign: [CW*max_nseq*PC]
    uint32_t* d_maxn = nullptr;
    uint16_t* d_scratch16 = nullptr;                                 // k_addaln per-pair notes
    uint32_t* d_submask = nullptr;                                   // [CW*(NC/32+1)] Subgraph membership by node id
    uint32_t* h_maxn = nullptr;                                      // pinned: [0] max rows, [1] max edges after a prune round
    // persistent build pipeline (vc_pipe.h): counters, three queues, the layer every window is at; the backtrack and resolver
    // kernels run beside the forward kernel on streams of their own
    uint32_t* d_pipe_ctl = nullptr; unsigned long long* d_pipe_slots = nullptr; uint32_t* d_cur_layer = nullptr;
    uint32_t pipe_cap = 0;
    uint8_t* d_pipe_ws = nullptr; size_t pipe_ws_bytes = 0;
    hipStream_t st_t = nullptr;
    hipEvent_t ev_seed = nullptr, ev_t = nullptr;
    bool pruned_known = false;                                       // h_maxn describes the current graphs
    // state of the chunk currently in flight
    uint32_t w0 = 0, ns = 0, layers = 0, nseq_max = 0;
    int cur = 0;
    bool active = false;
};

}  // namespace

struct vc_ctx;

namespace {
struct Plan;
struct DevSlot { void* p = nullptr; size_t cap = 0; };

// Everything that belongs to ONE submitted batch.  A context holds two: while the chunks of one run on the chunk streams, the
// next one is validated and copied in on the context's own stream (vc_submit) and the one before is copied out (vc_collect) --
// the shape of the reference's accelerated polisher, which fills the next batch while one computes (src/cuda/cudapolisher.cpp:246-277).
struct Batch {
    bool have = false;                   // a batch is staged here (vc_submit succeeded)
    VcBatchDev b{};
    DevSlot slots[16];                   // its device buffers (inputs, per-window outputs, collect scratch): grow-only, kept between batches
    std::vector<uint32_t> h_win_seq_off;
    std::vector<uint8_t> h_layer_partial;   // [layer] does any window have a partial-span layer at this index?
    std::vector<uint8_t> h_pre_status;   // per-window status decided at submit (outside the envelope), empty = none
    uint32_t max_layers = 0, max_len = 0, max_backbone = 0;
    uint32_t cpl = 0, cpl_min = 0;       // width classes of this batch's sequences (kernel selection)
    uint32_t kept = 0;                   // slots of the build phase's kept-row ring for this batch (0: plain ring); needs 15-bit row distances
    uint32_t ring = VC_RING, ring_pruned = VC_RING_PRUNED;   // rows of the plain ring (build phase without the kept-row ring / pruned graphs)
    bool packed = false;                 // stored DP rows of this batch: byte-packed (both score sets within the byte bound at the widest class) or raw
    bool band = false;                   // banded matrix store (vc_band_start): packed rows, row builders that mark the rows read back, cooperative backtrack
    uint32_t lean = 0;                   // VcFwdArgs::lean: the widest classes' on-the-fly profile is usable for global (bit 0) / local (bit 1) alignments
    uint32_t cw_run = 0, n_streams = 1;  // chunk size and chunk streams of this batch
    // ---- run state (guarded by vc_ctx::qmu while queued)
    bool queued = false;                 // on the run queue: its chunks are being handed to the stream workers
    bool ran = false, collected = false; // a run was started (and not invalidated) / its results were handed out
    uint64_t run_seq = 0;                // order of the vc_run calls
    uint32_t next_chunk = 0, n_chunks = 0, chunks_done = 0;
    int run_rc = 0;
    hipEvent_t done_ev[16]{};            // per chunk stream: recorded behind the last chunk of this batch on that stream
    bool ev_rec[16]{};
    uint32_t first_stream = 0;           // chunk stream that ran chunk 0 of the last run (the debug digests look at that workspace)
    Plan* pl = nullptr;                  // the plan of the run in flight (owned)
    // ---- results
    uint64_t total_cons = 0;
    std::vector<uint32_t> h_cons_len;
    std::vector<uint8_t> h_status;
};
}  // namespace

struct vc_ctx {
    vc_params prm{};
    int device = 0;
    hipStream_t stream = nullptr;       // the context's own stream: uploads, collects, fills
    std::string err;
    std::vector<void*> allocs;          // lifetime of the context
    std::vector<void*> chunk_allocs;    // workspaces (re-created when capacities change)
    uint64_t chunk_bytes = 0;           // what they hold: counts as available when the next batch is planned
    // The arena: memory the workspaces are carved from, so that a batch of another shape costs no hipFree / hipMalloc.  Giving ~90 GiB back
    // and asking for it again is what made a growing stream of batches stall for seconds (round 6, profiles/r6_cold_start.txt: hipMalloc
    // 4.7 s behind a hipFree of 86 GiB -- the driver scrubs what it takes back; one hipMalloc of 96 GiB: 3.2 s, of 32 GiB: 0.000 s).  One
    // in kArenaSegs SEGMENTS (each far below the size where hipMalloc turns slow); made by vc_reserve, or by the first large vc_submit.
    struct ArenaSeg { char* p; size_t bytes, used; };
    std::vector<ArenaSeg> arena;
    size_t arena_bytes = 0;             // sum over the segments
    uint32_t arena_cur = 0;             // segment the next workspace pieces come from (alloc_work of stream s: s % segments)
    bool ws_packed = false;             // the workspaces hold band space (the batch they were made for stores byte-packed rows)
    bool auto_streams = false;          // vc_params.n_streams was 0: vc_submit picks the chunk streams per batch
    uint32_t streams_made = 0;          // streams created (>= n_streams)
    bool auto_arena = true;             // vc_submit makes the arena itself for the first large batch (development: VC_AUTO_ARENA=0)
    bool have_ws = false;               // workspaces exist (out of the arena or allocated piece by piece)

    Batch bt[2];
    Batch* cur = nullptr;               // the batch vc_submit staged last: what vc_run starts
    uint64_t run_counter = 0;
    uint32_t max_nseq = 0;              // deepest window the workspaces are laid out for (pair lists per window)
    uint32_t* d_lut_w = nullptr; double* d_lut_d = nullptr;
    unsigned long long* d_stat = nullptr;   // [VC_STAT_SLOTS][8] cells, rows, -, far-row reads, trace steps, speculated steps, rounds, -

    uint32_t wcols = 0;                  // != 0: some alignment may need k_fwd_wide; columns of its int32 matrices (multiple of 512)
    uint32_t MA = 4;                     // entries per aligned list: max(4, distinct bytes in the batch - 1), even
    uint32_t ws_cpl = 0, ws_max_len = 0;   // width class / longest sequence the workspaces are sized for
    uint32_t NC = 0, EC = 0, CW = 0, STK = 2048, PC = 0, jobs_cap = 0, group_max = 1, rgroup_max = 1, n_streams = 1;
    uint64_t hmat_dwords = 0;
    uint32_t big_ws_stride = 0;          // bytes per window of the HBM workspace for oversized graph images (0: all fit the LDS)
    bool big_ws_topo = false;            // the workspace also backs k_topo's optimistic LDS image of the first pruned graphs
    bool trace_wave = true;
    int prune_hbm = 1;              // 1: the first prune of a chunk works from the HBM workspace instead of LDS; 2: every prune; 0: LDS
    bool topo_hbm = false;          // k_topo of the pruned graphs from the HBM workspace
    uint32_t dbg_stop_kind = 0, dbg_stop_index = 0;   // vc_debug_stop_after: leave the chunk's graphs as they are after that stage
    bool force_dfs = false;       // test knob: settle every end-cell tie with the exact DFS as well
    bool trace_block = false;     // VC_EXPERIMENTS builds, VC_TRACEB=1: k_traceb (the walk out of LDS, vc_traceb.h) for byte-packed rows
    uint32_t trace_tl = 8;        // lanes per alignment of the lock-step k_tracew (development: VC_TRACE_TL=16)
    bool dt = true;               // global alignments on byte-packed rows run on k_fwd_dt (development: VC_DT=0 keeps them on k_fwd)
    bool band_raw = false;        // -DVC_EXPERIMENTS builds, VC_BAND_RAW=1: raw int16 rows (widest classes, scores outside the byte form) store the band as well -- bit-identical,
                                  //   6 % slower on 3 kb windows (profiles/r6_ab_raw_band.txt: 4.7 % of their alignments leave 384 columns), so off
    bool inline_redo = false;     // development (VC_INLINE_REDO=1, read once at vc_create): the redo pair inside every build round instead of catch-up rounds
    bool fold = true;             // launch_fwd: more than two width classes in one launch (development: VC_NO_FOLD=1 launches once per class)
    uint32_t dup = 0;             // development (VC_DUP): launch idempotent kernel classes twice to measure their marginal cost inside the job
    Work works[kMaxStreams];
    hipStream_t streams[kMaxStreams]{};      // the process's chunk streams of this device (pooled_stream): not owned
    hipStream_t own_stream = nullptr;        // this context's stream for copies, fills and small kernels
    // One host thread per chunk stream (started with the first vc_run, kept until vc_destroy) takes chunks off the run queue and
    // drives each through its phases on that stream: the one host wait of the path -- the pruned graphs' height before a
    // re-alignment round -- then stalls that stream only, a stream takes its next chunk as soon as it has queued the last one,
    // and when a batch has no chunk left it goes on with the next batch of the queue: no barrier between chunks, none between batches.
    bool host_threads = true;
    // persistent build pipeline: the build loop of a chunk as three resident kernels and device-side queues instead of six launches
    // per layer (vc_pipe.h); pipe_f / pipe_t / pipe_r: resident workgroups of the forward / backtrack / resolver kernels (0: default)
    bool pipe = false;
    uint32_t pipe_f = 0, pipe_t = 0;
    uint32_t pipe_patience_s = 20;       // seconds a wave of the pipeline waits for an item before it declares the run failed (VC_PIPE_PATIENCE)
    unsigned long long* d_pipe_prof = nullptr;   // [VC_PP_N] phase clocks of the pipeline's waves, summed over a run (vc_debug_pipe_prof)
    uint32_t* d_pipe_abort = nullptr;    // != 0: a wave of the pipeline ran out of patience (site code): the run failed
    uint32_t n_cu = 256;
    std::thread workers[kMaxStreams];
    uint32_t workers_made = 0;
    bool stop = false;                   // vc_destroy: the workers leave
    std::mutex qmu;                      // run queue and the run state of the batches
    std::condition_variable qcv;
    std::deque<Batch*> runq;
    std::mutex mu;                       // err, event pool / records, launch counters

    void* h_stage[2] = {nullptr, nullptr};   // pinned staging for uploads from pageable caller memory
    hipEvent_t stage_ev[2] = {nullptr, nullptr};

    vc_stats stats{};
    std::vector<hipEvent_t> ev_pool;
    struct EvRec { int cls; hipEvent_t a, b; };
    std::vector<EvRec> ev_recs;
    size_t ev_next = 0;
};

namespace {

int fail(vc_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) { std::lock_guard<std::mutex> lk(c->mu); c->err = buf; } else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail((c), VC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Chunk streams are a property of the PROCESS, not of a context.  HIP maps streams onto a small pool of hardware queues by creation
// order and priority (see vc_create); a second context that created eight streams of its own ran 18 % slower beside an idle first one
// (26.9 against 32.8 k windows/s, tools/gpu_probe_ctx.py) because its streams landed on shared queues.  So every context of a device
// takes stream s from the same per-device set, made once and kept: the mapping any context sees is the one a lone context sees.
// Contexts that run at the same time then interleave their launches on the same streams, which orders them but changes no result.
struct StreamSet {
    hipStream_t chunk[kMaxStreams]{};          // the chunk streams, priorities cycling (never two neighbours on one hardware queue)
    hipStream_t side[kMaxStreams]{};           // persistent pipeline only: the stream its backtrack kernel runs on beside chunk stream s
};
std::mutex g_streams_mu;
std::map<int, StreamSet> g_streams;            // by device

int stream_priority(uint32_t s, uint32_t shift) {
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int n_prio = prio_least - prio_greatest + 1;
    return n_prio > 1 && !getenv("VC_SAME_PRIORITY") ? prio_least - (int)((s + shift) % (uint32_t)n_prio) : 0;
}
// stream s of the device's set (made on first use); side = the pipeline's second stream.  nullptr: creation failed
hipStream_t pooled_stream(int device, uint32_t s, bool side) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    StreamSet& set = g_streams[device];
    hipStream_t& st = side ? set.side[s] : set.chunk[s];
    if (!st && hipStreamCreateWithPriority(&st, hipStreamNonBlocking, stream_priority(s, side ? 1u : 0u)) != hipSuccess) st = nullptr;
    return st;
}

template <typename T>
int dalloc(vc_ctx* c, std::vector<void*>& list, T** out, size_t n) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(n * sizeof(T), 256);
    if (&list == &c->chunk_allocs && !c->arena.empty()) { // workspaces come out of the arena while it lasts: the stream's own segment first
        const size_t ns_ = c->arena.size();
        for (size_t k = 0; k < ns_; ++k) {
            vc_ctx::ArenaSeg& sg = c->arena[(c->arena_cur + k) % ns_];
            const size_t at = (sg.used + 255) & ~(size_t)255;
            if (at + bytes <= sg.bytes) { sg.used = at + bytes; *out = reinterpret_cast<T*>(sg.p + at); return VC_OK; }
        }
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(c, VC_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    list.push_back(p);
    *out = static_cast<T*>(p);
    return VC_OK;
}

// Waits until every chunk of the batch's run has been queued on its stream by the workers and has finished there.  Only the
// batch's own events are waited for: the chunk streams are the process's, another batch (or another context) may be running on them.
int wait_batch(vc_ctx* c, Batch* bt) {
    {
        std::unique_lock<std::mutex> lk(c->qmu);
        c->qcv.wait(lk, [&] { return !bt->queued; });
    }
    for (uint32_t s = 0; s < kMaxStreams; ++s) {
        if (!bt->ev_rec[s]) continue;
        hipError_t e = hipEventSynchronize(bt->done_ev[s]);
        if (e != hipSuccess) return fail(c, VC_ERR_HIP, "waiting for chunk stream %u failed: %s", s, hipGetErrorString(e));
        if (c->works[s].st_t) (void)hipStreamSynchronize(c->works[s].st_t);      // (persistent pipeline: the backtrack kernel's stream)
    }
    return VC_OK;
}

// nothing of this context is in flight any more (runs, copies)
void drain(vc_ctx* c) {
    for (Batch& bt : c->bt) (void)wait_batch(c, &bt);
    (void)hipStreamSynchronize(c->stream);
}

// ... and nothing a stopped run (vc_debug_stop_after) left on the chunk streams either
void sync_all(vc_ctx* c) {
    drain(c);
    for (uint32_t k = 0; k < kMaxStreams; ++k) {
        if (c->works[k].st_t) (void)hipStreamSynchronize(c->works[k].st_t);
        if (c->streams[k]) (void)hipStreamSynchronize(c->streams[k]);
    }
}

template <typename T>
int salloc(vc_ctx* c, Batch* bt, int slot, T** out, size_t n) {
    DevSlot& s = bt->slots[slot];
    const size_t bytes = std::max<size_t>(n * sizeof(T), 256);
    if (s.cap < bytes) {
        // (the buffer belongs to a batch that is not in flight; hipFree / hipMalloc synchronise the device themselves -- a stall for a
        // batch that is running, which is why the buffers are grow-only and kept)
        (void)hipStreamSynchronize(c->stream);
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr; s.cap = 0;
        const size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&s.p, want);
        if (e != hipSuccess) return fail(c, VC_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        s.cap = want;
    }
    *out = static_cast<T*>(s.p);
    return VC_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel in this process, not of a launch: two batches (or two
// contexts) in flight must not lower each other's limit.  Grow-only.
std::mutex g_lds_mu;
std::map<const void*, int> g_lds_limit;
int lds_limit(vc_ctx* c, const void* func, uint32_t bytes) {
    std::lock_guard<std::mutex> lk(g_lds_mu);
    int& cur = g_lds_limit[func];
    if ((int)bytes <= cur) return VC_OK;
    HIPCHK(c, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur = (int)bytes;
    return VC_OK;
}

void free_list(std::vector<void*>& l) {
    for (void* p : l) (void)hipFree(p);
    l.clear();
}
// The workspaces a staged batch was planned on are gone (vc_release / vc_reserve): such a batch can no longer be run -- its plan would
// launch on freed pointers.  Results of a finished run stay collectable (they live in the batch's own buffers); vc_run asks for a new
// vc_submit.
void unstage(vc_ctx* c) {
    for (Batch& b : c->bt) if (!b.ran || b.collected) b.have = false;       // (a finished run that nobody has collected keeps its results)
    // (c->cur stays: the batch's own buffers -- statuses, diagnostics -- are not workspaces and can still be read: vc_debug_errinfo)
}
void free_workspaces(vc_ctx* c) {         // the chunk workspaces: what was allocated piece by piece, and the arena's bump pointer
    free_list(c->chunk_allocs);
    for (auto& sg : c->arena) sg.used = 0;
    c->chunk_bytes = 0;
    c->have_ws = false;
}

void free_arena(vc_ctx* c) {
    for (auto& sg : c->arena) (void)hipFree(sg.p);
    c->arena.clear();
    c->arena_bytes = 0; c->arena_cur = 0;
}
// Default workspace budget: 60 % of what is free, capped.  The cap was 96 GiB until the end of round 6; larger chunks run faster (config C,
// tools/gpu_scale.py 100000 64 500: 64 GiB 36.0 k, 96 GiB 38.6 k, 160 GiB 39.1 k, 220 GiB 39.5 k windows/s -- profiles/r6_ab_workspace_budget.txt), and an
// MI355X has 288 GB.  The cap is 128 GiB now: four arena segments of 32 GiB, the largest hipMalloc that comes back at once -- 4 x 43 GiB (a cap of 176)
// take 3.4 s and turn the cold start of a process from 2.9 into 6.3 s.  VC_SCRATCH_CAP_GB overrides the cap, vc_params.scratch_bytes the whole rule.
uint64_t default_budget(size_t free_b) {
    uint64_t cap = 128ull << 30;
    if (const char* d = getenv("VC_SCRATCH_CAP_GB")) { const double g = std::atof(d); if (g >= 1.0) cap = (uint64_t)(g * 1073741824.0); }
    return std::min<uint64_t>((uint64_t)(free_b * 0.6), cap);
}
// bytes = 0: the default budget (vc_params.scratch_bytes, or 60 % of the free memory up to the cap of default_budget).  One segment per chunk stream.
int make_arena(vc_ctx* c, uint64_t bytes) {
    free_arena(c);
    if (!bytes) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        bytes = c->prm.scratch_bytes ? c->prm.scratch_bytes : default_budget(free_b);
    }
    const bool tm = getenv("VC_TIME_SUBMIT") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now();
    // four segments whatever the streams: a batch on four chunk streams lays a workspace into each, one on eight two, on sixteen four --
    // a segment per stream of EIGHT halved what a four-stream batch (config E: 4 096 windows of 1 kb x 128) could use, and its rate with it
    const uint32_t nseg = kArenaSegs;
    const size_t each = std::max<size_t>((size_t)(bytes / nseg) & ~(size_t)0xFFFFF, (size_t)1 << 20);
    for (uint32_t k = 0; k < nseg; ++k) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, each);
        if (e != hipSuccess) { free_arena(c); return fail(c, VC_ERR_HIP, "hipMalloc(%zu) failed: %s", each, hipGetErrorString(e)); }
        c->arena.push_back({(char*)p, each, 0});
        c->arena_bytes += each;
        // first touch now, not under the first batch (the driver hands out cleared memory and clears it when it has to): 17 ms per 96 GiB
        if ((e = hipMemsetAsync(p, 0, each, c->stream)) != hipSuccess) { free_arena(c); return fail(c, VC_ERR_HIP, "clearing the reserved workspace failed: %s", hipGetErrorString(e)); }
    }
    const double t_1 = now();
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { free_arena(c); return fail(c, VC_ERR_HIP, "clearing the reserved workspace failed: %s", hipGetErrorString(e)); }
    if (tm) fprintf(stderr, "make_arena: %u segments of %.1f GiB: hipMalloc + fill queued %.3f s, fills done %.3f s\n", nseg, each / 1073741824.0, t_1 - t_0, now() - t_1);
    return VC_OK;
}

int alloc_graph(vc_ctx* c, VcGraph* g) {
    const size_t CW = c->CW, NC = c->NC, EC = c->EC;
    int rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->n_nodes, CW))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->n_edges, CW))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->code, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->in_first, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->in_last, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->out_first, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->out_last, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->al_cnt, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->al, CW * NC * c->MA))) return rc;
    g->ma = c->MA;
    if ((rc = dalloc(c, c->chunk_allocs, &g->e_tn, CW * EC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->e_hn, CW * EC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->e_w, CW * EC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->ord, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->pos, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->visits, CW * NC))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &g->nrec, CW * NC))) return rc;
    return VC_OK;
}

int alloc_work(vc_ctx* c, Work* wk) {
    const size_t CW = c->CW, NC = c->NC, EC = c->EC, PC = c->PC;
    int rc;
    if ((rc = alloc_graph(c, &wk->gr[0])) || (rc = alloc_graph(c, &wk->gr[1]))) return rc;
    if ((rc = dalloc(c, c->chunk_allocs, &wk->dp.nrows, CW)) || (rc = dalloc(c, c->chunk_allocs, &wk->dp.flags, CW)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->dp.rec, CW * NC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->dp.frec, CW * NC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->dp.fie, CW * (NC + 4) + 64)) ||      // (+ 64: k_tracew reads whole 8-byte chunks of a window's entries)
        (rc = dalloc(c, c->chunk_allocs, &wk->dp.rank2node, CW * NC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->dp.ovf, CW * EC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_hmat, c->hmat_dwords + 64)) ||      // (+ 64: k_traceb reads whole 48-byte windows of a row)
        (rc = dalloc(c, c->chunk_allocs, &wk->d_bmat, ((c->ws_packed || c->band_raw) ? c->hmat_dwords / 4 + (size_t)c->jobs_cap * VC_BAND_JOB_PAD_DWORDS : 0) + 64)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_band_par, (size_t)c->jobs_cap * 2)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_redo_list, c->jobs_cap)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_redo_n, 4)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_resolve_ws, (size_t)kResolveGrid * ((topo_lds_bytes(NC, c->EC, c->STK, c->MA) + 15u) & ~15u))) ||
        (c->big_ws_stride && (rc = dalloc(c, c->chunk_allocs, &wk->d_big_ws, (size_t)CW * c->big_ws_stride))) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_c0, (size_t)c->jobs_cap * NC)) ||
        (c->wcols && ((rc = dalloc(c, c->chunk_allocs, &wk->d_wmat, (size_t)c->jobs_cap * NC * c->wcols)) ||
                      (rc = dalloc(c, c->chunk_allocs, &wk->d_c0w, (size_t)c->jobs_cap * NC)))) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_job_end, c->jobs_cap)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_job_type, c->jobs_cap)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_tie_rows, (size_t)c->jobs_cap * VC_MAXTIE)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_tie_cnt, c->jobs_cap)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_tie_list, c->jobs_cap)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_tie_n, 4)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_pairs, CW * PC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_npairs, CW)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_rpairs, CW * c->max_nseq * PC)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_rnpairs, CW * c->max_nseq)) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_scratch16, CW * (4 * PC + NC))) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_submask, CW * (NC / 32 + 1))) ||
        (rc = dalloc(c, c->chunk_allocs, &wk->d_maxn, 2)))
        return rc;
    {
        uint32_t cap = 64;
        while (cap < CW) cap <<= 1;
        wk->pipe_cap = cap;
        // (the exact-DFS fallback of the end-cell resolver works from an HBM image of the graph: one per backtrack workgroup)
        wk->pipe_ws_bytes = (size_t)std::min<uint64_t>((CW + VC_TG - 1) / VC_TG, 256u * 8u) * ((topo_lds_bytes(NC, c->EC, c->STK, c->MA) + 15u) & ~15u);
        if ((rc = dalloc(c, c->chunk_allocs, &wk->d_cur_layer, CW))) return rc;
#ifdef VC_EXPERIMENTS
        if ((rc = dalloc(c, c->chunk_allocs, &wk->d_pipe_ctl, (size_t)VC_PC_N * VC_PIPE_CTL_STRIDE)) ||
            (rc = dalloc(c, c->chunk_allocs, &wk->d_pipe_slots, (size_t)2 * cap)) ||
            (rc = dalloc(c, c->chunk_allocs, &wk->d_pipe_ws, wk->pipe_ws_bytes)))
            return rc;
#endif
    }
    return VC_OK;
}


constexpr size_t kStageBytes = 64u << 20;

// H2D from pageable caller memory through two pinned staging buffers (copy-in of chunk k+1 overlaps
// the DMA of chunk k); small arrays go straight through hipMemcpyAsync
int h2d(vc_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes < (4u << 20)) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); return VC_OK; }
    for (int i = 0; i < 2; ++i) {
        if (!c->h_stage[i]) { HIPCHK(c, hipHostMalloc(&c->h_stage[i], kStageBytes)); HIPCHK(c, hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming)); }
    }
    size_t off = 0;
    int k = 0;
    while (off < bytes) {
        const size_t n = std::min(kStageBytes, bytes - off);
        HIPCHK(c, hipEventSynchronize(c->stage_ev[k]));                 // buffer k free again
        {   // one thread copies ~6 GB/s, the link takes ten times that: the staging copy, not the DMA, was what a 1 GB submit waited for
            const size_t T = n >= (8u << 20) ? 4 : 1;
            std::vector<std::thread> th;
            for (size_t t = 1; t < T; ++t)
                th.emplace_back([=]() { std::memcpy((char*)c->h_stage[k] + n * t / T, (const char*)src + off + n * t / T, n * (t + 1) / T - n * t / T); });
            std::memcpy(c->h_stage[k], (const char*)src + off, n / T);
            for (auto& x : th) x.join();
        }
        HIPCHK(c, hipMemcpyAsync((char*)dst + off, c->h_stage[k], n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->stage_ev[k], c->stream));
        off += n; k ^= 1;
    }
    return VC_OK;
}

struct Timer {
    vc_ctx* c; int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool timed = false;
    Timer(vc_ctx* c_, int cls_, hipStream_t st_) : c(c_), cls(cls_), st(st_) {
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->stats.launches[cls]++;
            timed = c->prm.profile == 1 || (c->prm.profile == 2 && cls == KC_FWD);
            if (!timed) return;
            if (c->ev_next + 2 > c->ev_pool.size()) {
                for (int i = 0; i < 2; ++i) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e); }
            }
            a = c->ev_pool[c->ev_next++]; b = c->ev_pool[c->ev_next++];
        }
        (void)hipEventRecord(a, st);
    }
    ~Timer() {
        if (!timed) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(c->mu);
        c->ev_recs.push_back({cls, a, b});
    }
};

void flush_events(vc_ctx* c) {
    // per class: the sum of the launch durations, and the length of the union of the launch intervals (launches of one class
    // on different chunk streams overlap each other)
    std::vector<std::pair<float, float>> iv[KC_N];
    for (auto& r : c->ev_recs) {
        float ms = 0, t0 = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        c->stats.ms[r.cls] += ms;
        if (hipEventElapsedTime(&t0, c->ev_recs.front().a, r.a) == hipSuccess) iv[r.cls].emplace_back(t0, t0 + ms);
    }
    for (int k = 0; k < KC_N; ++k) {
        std::sort(iv[k].begin(), iv[k].end());
        float end = -1e30f;
        double busy = 0;
        for (auto& x : iv[k]) {
            if (x.first > end) { busy += x.second - x.first; end = x.second; }
            else if (x.second > end) { busy += x.second - end; end = x.second; }
        }
        c->stats.busy_ms[k] += busy;
    }
    c->ev_recs.clear();
    c->ev_next = 0;
}

// nwonly: every alignment of the launch is global (mode 0 always; a re-alignment launch whose layers are all full-span)
// dt: global alignments on byte-packed rows take the doubly tilted kernel (k_fwd_dt, vc_fwd_dt.h) -- decided per batch (launch_fwd)
template <int CA, int CB>
void launch_fwd_t(hipStream_t st, const VcFwdArgs& a, uint32_t jobs, bool packed, bool nwonly, bool dt) {
    constexpr int KR = CB >= 32 ? kKeptWide : (kKept ? kKept : 1), PR = CB >= 32 ? kRingWide : kRing, QR = CB >= 32 ? kRingPrunedWide : kRingPruned;
    if constexpr (CB < 32) {
        if (dt && packed && (a.mode == 0 || (a.mode == 1 && nwonly))) {
            if (a.mode == 0 && a.kept) hipLaunchKernelGGL((k_fwd_dt<CA, CB, KR, true>), dim3(jobs), dim3(64), 0, st, a);
            else if (a.mode == 0) hipLaunchKernelGGL((k_fwd_dt<CA, CB, PR, false>), dim3(jobs), dim3(64), 0, st, a);
            else hipLaunchKernelGGL((k_fwd_dt<CA, CB, QR, false>), dim3(jobs), dim3(64), 0, st, a);
            return;
        }
    }
    if (a.mode == 0 && a.kept) {
        if (packed) hipLaunchKernelGGL((k_fwd<CA, CB, KR, true, true, true>), dim3(jobs), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_fwd<CA, CB, KR, false, true, true>), dim3(jobs), dim3(64), 0, st, a);
    } else if (a.mode == 0) {
        if (packed) hipLaunchKernelGGL((k_fwd<CA, CB, PR, true, false, true>), dim3(jobs), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_fwd<CA, CB, PR, false, false, true>), dim3(jobs), dim3(64), 0, st, a);
    } else if (nwonly && packed) {
        hipLaunchKernelGGL((k_fwd<CA, CB, QR, true, false, true>), dim3(jobs), dim3(64), 0, st, a);
    } else {
        if (packed) hipLaunchKernelGGL((k_fwd<CA, CB, QR, true, false, false>), dim3(jobs), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_fwd<CA, CB, QR, false, false, false>), dim3(jobs), dim3(64), 0, st, a);
    }
}

// lower class of a folded launch (launch_fwd), 0 when the batch's classes need no folding: what the backtrack reads the rows with
uint32_t fold_lo(const vc_ctx* c, const Batch* bt) {
    const uint32_t opts[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    int lo = -1, hi = -1;
    for (int i = 0; i < 11; ++i) { if (opts[i] == bt->cpl_min) lo = i; if (opts[i] == bt->cpl) hi = i; }
    return (c->fold && lo >= 0 && hi - lo > 1) ? opts[hi - 1] : 0u;
}

// One launch when the batch's sequences fall into one width class or two adjacent ones (the usual case:
// read pieces of a window differ by a few percent in length); otherwise one launch per class.
int launch_fwd(vc_ctx* c, const Batch* bt, hipStream_t st, const VcFwdArgs& a0, uint32_t jobs, const Work* wk = nullptr, bool nwonly = false) {
    if (c->dup & 16u) { const uint32_t d = c->dup; c->dup = 0; (void)launch_fwd(c, bt, st, a0, jobs, wk, nwonly); c->dup = d; (void)hipMemsetAsync(a0.tie_n, 0, 4, st); }
    const uint32_t opts[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    VcFwdArgs a = a0;
    a.do_init = 1;
    int lo = -1, hi = -1;
    for (int i = 0; i < 11; ++i) { if (opts[i] == bt->cpl_min) lo = i; if (opts[i] == bt->cpl) hi = i; }
    if (lo < 0 || hi < 0) return fail(c, VC_ERR_ARG, "unsupported cells-per-lane %u..%u", bt->cpl_min, bt->cpl);
    auto wide = [&]() {                     // alignments the packed-int16 kernels declined (their job_type is still 255)
        if (c->wcols && wk) { Timer t(c, KC_FWD, st); hipLaunchKernelGGL(k_fwd_wide, dim3(jobs), dim3(64), 0, st, a, wk->d_wmat, (uint64_t)c->NC * c->wcols, c->wcols, wk->d_c0w); }
    };
    // more than two classes (partial-span layers: pieces of reads of any length): one launch built for the two widest ones, every
    // narrower sequence in the lower of them -- four launches per layer, each waiting for its slowest alignment, become one
    if (hi - lo > 1 && c->fold) { lo = hi - 1; a.fold = 1; }
    // the doubly tilted form (vc_fwd_dt.h): judged on the workspaces' row capacity, so that every window of the batch takes the same kernel
    const bool dt = c->dt && bt->packed && bt->cpl < 32 && vc_dt_ok(c->prm.match, c->prm.mismatch, c->prm.gap, c->NC, bt->cpl);
    if (hi - lo == 1) {
        { Timer t(c, KC_FWD, st);
        switch (hi) {
#ifndef VC_FAST_BUILD          // development builds (-DVC_FAST_BUILD) carry only the width classes of the benchmark
            case 1: launch_fwd_t<4, 6>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 2: launch_fwd_t<6, 8>(st, a, jobs, bt->packed, nwonly, dt); break;
#endif
            case 3: launch_fwd_t<8, 10>(st, a, jobs, bt->packed, nwonly, dt); break;
#ifndef VC_FAST_BUILD
            case 4: launch_fwd_t<10, 12>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 5: launch_fwd_t<12, 16>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 6: launch_fwd_t<16, 20>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 7: launch_fwd_t<20, 24>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 8: launch_fwd_t<24, 32>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 9: launch_fwd_t<32, 48>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 10: launch_fwd_t<48, 64>(st, a, jobs, bt->packed, nwonly, dt); break;
#else
            default: return fail(c, VC_ERR_ARG, "development build: width classes 8 / 10 only");
#endif
        }
        }
        wide();
        return VC_OK;
    }
    for (int i = lo; i <= hi; ++i) {
        Timer t(c, KC_FWD, st);
        switch (opts[i]) {
#ifndef VC_FAST_BUILD
            case 4:  launch_fwd_t<4, 4>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 6:  launch_fwd_t<6, 6>(st, a, jobs, bt->packed, nwonly, dt); break;
#endif
            case 8:  launch_fwd_t<8, 8>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 10: launch_fwd_t<10, 10>(st, a, jobs, bt->packed, nwonly, dt); break;
#ifndef VC_FAST_BUILD
            case 12: launch_fwd_t<12, 12>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 16: launch_fwd_t<16, 16>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 20: launch_fwd_t<20, 20>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 24: launch_fwd_t<24, 24>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 32: launch_fwd_t<32, 32>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 48: launch_fwd_t<48, 48>(st, a, jobs, bt->packed, nwonly, dt); break;
            case 64: launch_fwd_t<64, 64>(st, a, jobs, bt->packed, nwonly, dt); break;
#else
            default: return fail(c, VC_ERR_ARG, "development build: width classes 8 / 10 only");
#endif
        }
        a.do_init = 0;
    }
    wide();
    return VC_OK;
}

#ifdef VC_EXPERIMENTS
// the forward kernel of the persistent build pipeline for this batch's width classes (one class, or two adjacent ones)
template <int CA, int CB>
int launch_pipe_fwd_t(vc_ctx* c, hipStream_t st, const VcPipeFwdArgs& a, uint32_t grid, uint32_t lds) {
    auto k = k_pipe_fwd<CA, CB, (kKept ? kKept : 1), true>;
    { int rc_ = lds_limit(c, (const void*)k, lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return VC_OK;
}
// The pipeline's forward kernel is built for the widest class of the batch and the one below it (everything narrower runs
// in that lower class: partial-span layers are short).  -> (CA, CB), CA == CB when the batch has one class.
bool pipe_classes(const Batch* bt, uint32_t* ca, uint32_t* cb) {
    const uint32_t opts[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    int lo = -1, hi = -1;
    for (int i = 0; i < 11; ++i) { if (opts[i] == bt->cpl_min) lo = i; if (opts[i] == bt->cpl) hi = i; }
    if (lo < 0 || hi < 0) return false;
    *cb = opts[hi]; *ca = lo < hi ? opts[hi - 1] : opts[hi];
#ifdef VC_FAST_BUILD
    return *cb == 10 || *cb == 8;
#else
    return *cb < 32;                   // (the wide classes -- 32 columns per lane and up -- run with their own ring / kept-row parameters, kKeptWide and
                                       //  bt->ring: build_pipe and k_pipe_fwd are written for kKept / kRing, so those batches take the lock-step plan)
#endif
}
int launch_pipe_fwd(vc_ctx* c, const Batch* bt, hipStream_t st, const VcPipeFwdArgs& a, uint32_t grid, uint32_t lds) {
    uint32_t lo = 0, hi = 0;
    if (!pipe_classes(bt, &lo, &hi)) return fail(c, VC_ERR_ARG, "persistent pipeline: unsupported cells-per-lane %u..%u", bt->cpl_min, bt->cpl);
#define VC_PF(A, B) if (lo == A && hi == B) return launch_pipe_fwd_t<A, B>(c, st, a, grid, lds);
    VC_PF(8, 10) VC_PF(8, 8) VC_PF(10, 10) VC_PF(6, 8)
#ifndef VC_FAST_BUILD
    VC_PF(4, 4) VC_PF(4, 6) VC_PF(6, 6) VC_PF(10, 12) VC_PF(12, 12) VC_PF(12, 16) VC_PF(16, 16) VC_PF(16, 20) VC_PF(20, 20)
    VC_PF(20, 24) VC_PF(24, 24)
#endif
#undef VC_PF
    return fail(c, VC_ERR_ARG, "persistent pipeline: width classes %u..%u not built", lo, hi);
}
#endif

uint32_t pick_cpl(uint32_t max_len) {
    const uint32_t opts[] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    for (uint32_t o : opts) if (64 * o >= max_len) return o;
    return 0;
}

// ---------------------------------------------------------------- one chunk, phase by phase
struct Plan {
    vc_ctx* c;
    Batch* bt;
    uint32_t NC, EC, PC, cpl, topo_lds, prune_lds, add_lds, rows_lds, cons_lds;
    uint64_t rowd;          // dwords per H row

    VcFwdArgs fwd_args(const Work& wk) const {
        VcFwdArgs fa{};
        fa.b = bt->b; fa.dp = wk.dp; fa.w0 = wk.w0; fa.nslots = wk.ns; fa.NC = NC; fa.EC = EC;
        fa.m = c->prm.match; fa.n = c->prm.mismatch; fa.g = c->prm.gap;
        fa.sm = c->prm.sw_match; fa.sn = c->prm.sw_mismatch; fa.sg = c->prm.sw_gap;
        fa.hmat = wk.d_hmat; fa.c0 = wk.d_c0;
        fa.job_end = wk.d_job_end; fa.job_type = wk.d_job_type; fa.tie_rows = wk.d_tie_rows; fa.tie_cnt = wk.d_tie_cnt; fa.tie_list = wk.d_tie_list; fa.tie_n = wk.d_tie_n;
        fa.stat = c->d_stat; fa.wcols = c->wcols; fa.kept = bt->kept;
        fa.bmat = wk.d_bmat; fa.band_par = wk.d_band_par; fa.band = bt->band ? 1 : 0; fa.redo_list = nullptr; fa.redo_n = nullptr;
        fa.fold = 0;                  // (launch_fwd decides)
        fa.lean = bt->lean;
        return fa;
    }
    VcTraceArgs trace_args(const Work& wk) const {
        VcTraceArgs ta{};
        ta.b = bt->b; ta.dp = wk.dp; ta.w0 = wk.w0; ta.nslots = wk.ns; ta.NC = NC; ta.EC = EC; ta.cpl = cpl;
        ta.m = c->prm.match; ta.n = c->prm.mismatch; ta.g = c->prm.gap;
        ta.sm = c->prm.sw_match; ta.sn = c->prm.sw_mismatch; ta.sg = c->prm.sw_gap;
        ta.wmat = wk.d_wmat; ta.wstride = (uint64_t)NC * c->wcols; ta.wcols = c->wcols; ta.c0w = wk.d_c0w; ta.only_wide = 0;
        ta.stat = c->d_stat; ta.hmat = wk.d_hmat; ta.c0 = wk.d_c0; ta.job_end = wk.d_job_end; ta.job_type = wk.d_job_type; ta.PC = PC; ta.packed = bt->packed ? 1 : 0; ta.kept = 0;
        ta.bmat = wk.d_bmat; ta.band_par = wk.d_band_par; ta.band = bt->band ? 1 : 0; ta.redo_list = nullptr; ta.redo_n = nullptr;
        ta.redo_out = wk.d_redo_list; ta.redo_out_n = wk.d_redo_n;
        ta.cpl_lo = fold_lo(c, bt);       // (the pipeline sets its own)
        return ta;
    }

    // the backtrack of `njobs` alignments (gsz per window) whose graphs have at most `max_rows` rows
    void launch_trace(Work& wk, VcTraceArgs& ta, uint32_t njobs, uint32_t gsz, uint32_t max_rows) {
        if (c->dup & 1u) { const uint32_t d = c->dup; c->dup = 0; launch_trace(wk, ta, njobs, gsz, max_rows); c->dup = d; }
        Timer t(c, KC_TRACE, wk.stream);
        if (!c->trace_wave) {
            hipLaunchKernelGGL(k_trace, dim3((njobs + VC_TRACE_LANES - 1) / VC_TRACE_LANES), dim3(64), 0, wk.stream, ta);
            return;
        }
        // eight alignments per wave, eight lanes each (development: VC_TRACE_TL=16 -- four alignments of sixteen lanes, the form the pipeline uses)
#ifdef VC_EXPERIMENTS
        if (c->trace_block && ta.packed) {
            // the walk out of LDS (vc_traceb.h; VC_TRACEB=1): bit-identical and a quarter SLOWER than k_tracew (profiles/r6_ab_traceb.txt) -- an experiment
            hipLaunchKernelGGL(k_traceb, dim3((njobs + 7) / 8), dim3(64), vc_traceb_lds_bytes(), wk.stream, ta);
            if (c->wcols) { ta.only_wide = 1; hipLaunchKernelGGL(k_trace, dim3((njobs + VC_TRACE_LANES - 1) / VC_TRACE_LANES), dim3(64), 0, wk.stream, ta); ta.only_wide = 0; }
            return;
        }
#endif
        const uint32_t tg = c->trace_tl == 16 ? 4u : 8u;
        ta.shared_table = gsz % tg == 0; ta.tab_rows = std::min(max_rows, kTraceTabRows);
        const uint32_t lds = vc_tracew_lds_bytes(ta.tab_rows, ta.shared_table != 0, tg);
        if (tg == 4) hipLaunchKernelGGL(k_tracew<16>, dim3((njobs + 3) / 4), dim3(64), lds, wk.stream, ta);
        else hipLaunchKernelGGL(k_tracew<8>, dim3((njobs + 7) / 8), dim3(64), lds, wk.stream, ta);
        if (c->wcols) { ta.only_wide = 1; hipLaunchKernelGGL(k_trace, dim3((njobs + VC_TRACE_LANES - 1) / VC_TRACE_LANES), dim3(64), 0, wk.stream, ta); ta.only_wide = 0; }
    }

    // banded store: the alignments whose backtrack needed a cell outside the band (a few per thousand) run once more with
    // whole rows, and are walked from there; both launches find their jobs on the list the first backtrack left
    int redo(Work& wk, VcFwdArgs fa, VcTraceArgs ta, uint32_t njobs, uint32_t max_rows, bool nwonly = false) {
        if (!bt->band || fa.mode == 2) return VC_OK;
        fa.redo_list = wk.d_redo_list; fa.redo_n = wk.d_redo_n; fa.band = 0;
        int rc = launch_fwd(c, bt, wk.stream, fa, njobs, nullptr, nwonly);
        if (rc) return rc;
        ta.redo_list = wk.d_redo_list; ta.redo_n = wk.d_redo_n; ta.band = 0;
        launch_trace(wk, ta, njobs, 1, max_rows);         // listed jobs are of any window: no shared first-in-edge table
        return VC_OK;
    }

    void begin(Work& wk, uint32_t w0, uint32_t ns) {
        wk.w0 = w0; wk.ns = ns; wk.cur = 0; wk.layers = 0; wk.nseq_max = 0; wk.active = true; wk.pruned_known = false;
        for (uint32_t w = w0; w < w0 + ns; ++w) {
            const uint32_t n = bt->h_win_seq_off[w + 1] - bt->h_win_seq_off[w];
            wk.nseq_max = std::max(wk.nseq_max, n);
            if (n >= 3) wk.layers = std::max(wk.layers, n - 1);
        }
        (void)hipMemsetAsync(wk.dp.nrows, 0, (size_t)ns * 4, wk.stream);      // skipped windows must not carry a stale height
        for (int gi = 0; gi < 2; ++gi) {                                       // ... nor stale graph sizes: prune() sizes the next round's LDS
            (void)hipMemsetAsync(wk.gr[gi].n_nodes, 0, (size_t)ns * 4, wk.stream);   // images from the maximum over the chunk, skipped windows
            (void)hipMemsetAsync(wk.gr[gi].n_edges, 0, (size_t)ns * 4, wk.stream);   // (fewer than three sequences) included
        }
        { Timer t(c, KC_AVG, wk.stream); hipLaunchKernelGGL(k_avg, dim3(ns), dim3(64), 0, wk.stream, bt->b, w0, ns); }
        { Timer t(c, KC_INIT, wk.stream); hipLaunchKernelGGL(k_init, dim3(ns), dim3(64), bt->kept ? vc_kept_lds_bytes(NC) : 0, wk.stream, bt->b, wk.gr[0], wk.dp, w0, ns, NC, EC, bt->ring, bt->kept, wk.d_cur_layer); }
    }

    // One round of the build loop (window.cpp:239-298) for every window of the chunk.  Every window is at a layer of its own
    // (Work::d_cur_layer; the reference runs a window as one sequential function, window.cpp:239-298, and the windows of a chunk need
    // not march together): the kernels take the layer from the cursor, k_addaln moves a window on when its alignment is in.
    //   inline_redo = false: a window whose backtrack left the stored band (0.14 % of the alignments) simply does not move on; the
    //     next round's forward launch aligns the same layer again with whole rows and the walk is done from there.  The redo
    //     launch pair that every chunk-layer used to pay (half of the build phase's k_fwd launches, each as long as one alignment)
    //     is gone; the chunk ends with a few catch-up rounds for the windows that fell behind (Plan::run_chunk).
    //   inline_redo = true: the redo pair inside the round, as before -- every window moves on in every round (catch-up rounds,
    //     and the lock-step walk of vc_debug_stop_after, whose digests want every window at layer j after round j).
    int build_layer(Work& wk, uint32_t j, bool inline_redo) {
        const uint32_t ns = wk.ns;
        // the rows of a full-span layer were made at the tail of the kernel that last changed the graph (k_init / k_addaln);
        // a partial-span layer aligns to a Subgraph (a window that lags may be at any earlier layer)
        bool any_partial = false;
        for (uint32_t q = 1; q <= j && q < bt->h_layer_partial.size(); ++q) any_partial = any_partial || bt->h_layer_partial[q];
        if (any_partial) {
            Timer t(c, KC_ROWS, wk.stream);
            const uint32_t sub_lds = std::max(8 * ((NC + 63) / 64) + 2 * NC + ((NC + 15) & ~15u) + 4 * (NC / 32 + 1) + 64, bt->kept ? vc_kept_lds_bytes(NC) : 0u);
            hipLaunchKernelGGL(k_rows_sub, dim3(ns), dim3(64), sub_lds, wk.stream, bt->b, wk.gr[wk.cur], wk.dp, wk.w0, ns, NC, EC, (int)j, bt->ring, wk.d_submask, bt->kept,
                               (const uint32_t*)wk.d_cur_layer);
        }
        VcFwdArgs fa = fwd_args(wk);
        fa.group = 1; fa.k0 = j; fa.mode = 0; fa.hstride = (uint64_t)NC * rowd; fa.cursor = wk.d_cur_layer;
        fa.tie_over = wk.d_pairs; fa.tie_over_stride = PC;      // the pair list of the job is written only after k_resolve
        if (j == 1 || c->dbg_stop_kind) {                      // later layers: k_addaln of the layer before cleared the counters
            HIPCHK(c, hipMemsetAsync(wk.d_tie_n, 0, 4, wk.stream));
            HIPCHK(c, hipMemsetAsync(wk.d_redo_n, 0, 4, wk.stream));
        }
        int rc = launch_fwd(c, bt, wk.stream, fa, ns, &wk);
        if (rc) return rc;
        { Timer t(c, KC_RESOLVE, wk.stream);
          const uint32_t rs_lds = 4 * ((NC + 31) / 32 + 1) + 2 * 256 + 16;
          hipLaunchKernelGGL(k_resolve, dim3(kResolveGrid), dim3(64), rs_lds, wk.stream, bt->b, wk.gr[wk.cur], wk.dp, wk.w0, ns, NC, EC, c->STK,
                             (const uint16_t*)wk.d_tie_rows, (const uint32_t*)wk.d_tie_cnt, (const uint32_t*)wk.d_pairs, PC, wk.d_job_end,
                             (const uint32_t*)wk.d_tie_list, (const uint32_t*)wk.d_tie_n, (const uint32_t*)wk.d_submask, (int)j,
                             wk.d_resolve_ws, (topo_lds + 15u) & ~15u, c->force_dfs ? 1 : 0, (const uint32_t*)wk.d_cur_layer); }
        VcTraceArgs ta = trace_args(wk);
        ta.group = 1; ta.k0 = j; ta.hstride = fa.hstride; ta.cursor = wk.d_cur_layer;
        ta.pairs = wk.d_pairs; ta.npairs = wk.d_npairs; ta.pair_group = 1; ta.pair_k0 = j; ta.kept = bt->kept;
        launch_trace(wk, ta, ns, 1, NC);
        if (inline_redo && (rc = redo(wk, fa, ta, ns, NC))) return rc;
        VcAddArgs aa{};
        aa.b = bt->b; aa.g = wk.gr[wk.cur]; aa.dp = wk.dp; aa.w0 = wk.w0; aa.nslots = ns; aa.NC = NC; aa.EC = EC; aa.layer = j;
        aa.pairs = wk.d_pairs; aa.npairs = wk.d_npairs; aa.PC = PC; aa.scratch = wk.d_scratch16; aa.ring = bt->ring;
        aa.make_rows = !(c->dbg_stop_kind == 1 && c->dbg_stop_index == j); aa.kept = bt->kept;
        aa.tie_n = wk.d_tie_n; aa.redo_n = wk.d_redo_n; aa.cursor = wk.d_cur_layer;
        { Timer t(c, KC_ADDALN, wk.stream); hipLaunchKernelGGL(k_addaln, dim3(ns), dim3(64), add_lds, wk.stream, aa); }
        return VC_OK;
    }

    // every layer of every window of the chunk: `layers` rounds, then as many catch-up rounds as the slowest window is behind
    int build_loop(Work& wk) {
        int rc;
        const bool defer = bt->band && !c->inline_redo;                 // (development: VC_INLINE_REDO=1 keeps the redo pair in every round)
        for (uint32_t j = 1; j <= wk.layers; ++j) if ((rc = build_layer(wk, j, !defer))) return rc;
        if (!defer) return VC_OK;
        // (the one host wait of the build phase; the re-alignment rounds have theirs: Plan::realign)
        HIPCHK(c, hipMemsetAsync(wk.d_maxn, 0, 8, wk.stream));
        hipLaunchKernelGGL(k_lag, dim3((wk.ns + 255) / 256), dim3(256), 0, wk.stream, bt->b, (const uint32_t*)wk.d_cur_layer, wk.w0, wk.ns, wk.d_maxn);
        HIPCHK(c, hipMemcpyAsync(wk.h_maxn + 4, wk.d_maxn, 4, hipMemcpyDeviceToHost, wk.stream));
        HIPCHK(c, hipStreamSynchronize(wk.stream));
        const uint32_t lag = std::min(wk.h_maxn[4], wk.layers);
        // with the redo pair inside the round every window moves on in every round: `lag` rounds bring the slowest one home
        for (uint32_t r = 0; r < lag; ++r) if ((rc = build_layer(wk, wk.layers, true))) return rc;
        return VC_OK;
    }

#ifndef VC_EXPERIMENTS
    bool pipe_ok() const { return false; }
    int build_pipe(Work&) { return VC_ERR_STATE; }
#else
    // The whole build loop of the chunk (window.cpp:239-298, every layer of every window) as ONE set of resident kernels working
    // off device-side queues (vc_pipe.h) instead of build_layer() once per layer.
    bool pipe_ok() const {
        uint32_t ca_ = 0, cb_ = 0;
        return c->pipe && bt->kept && bt->packed && bt->band && c->wcols == 0 && c->dbg_stop_kind == 0 && c->trace_wave && pipe_classes(bt, &ca_, &cb_);
    }
    int build_pipe(Work& wk) {
        const uint32_t ns = wk.ns, cap = wk.pipe_cap;
        uint32_t shift = 0;
        while ((1u << shift) < cap) ++shift;
        HIPCHK(c, hipMemsetAsync(wk.d_pipe_ctl, 0, (size_t)VC_PC_N * VC_PIPE_CTL_STRIDE * 4, wk.stream));
        HIPCHK(c, hipMemsetAsync(wk.d_pipe_slots, 0, (size_t)2 * cap * 8, wk.stream));
        VcPipe p{};
        auto ctl = [&](int i) { return wk.d_pipe_ctl + (size_t)i * VC_PIPE_CTL_STRIDE; };
        p.fq = VcQueue{wk.d_pipe_slots, ctl(VC_PC_FQ_RES), ctl(VC_PC_FQ_HEAD), cap - 1, shift};
        p.tq = VcQueue{wk.d_pipe_slots + cap, ctl(VC_PC_TQ_RES), ctl(VC_PC_TQ_HEAD), cap - 1, shift};
        p.n_active = ctl(VC_PC_ACTIVE); p.done = ctl(VC_PC_DONE); p.finished = ctl(VC_PC_FINISHED); p.abort_code = c->d_pipe_abort;
        p.cur_layer = wk.d_cur_layer;
        p.prof = c->d_pipe_prof;
        p.pub_time = getenv("VC_PIPE_PUBTIME") ? reinterpret_cast<unsigned long long*>(wk.d_scratch16 + (size_t)c->CW * (4 * PC + NC)) - c->CW : nullptr;   // development: tail of the note blocks
        p.spin_limit = c->pipe_patience_s * 100000000u;       // ticks of the 100 MHz clock: this long without an item is a protocol error
        if (!wk.st_t) {                                       // the backtrack kernel's stream beside this chunk stream, on first use
            wk.st_t = pooled_stream(c->device, (uint32_t)(&wk - c->works), true);
            if (!wk.st_t) return fail(c, VC_ERR_HIP, "hipStreamCreate failed");
        }
        hipLaunchKernelGGL(k_pipe_seed, dim3((ns + 255) / 256), dim3(256), 0, wk.stream, bt->b, p, wk.w0, ns);
        HIPCHK(c, hipEventRecord(wk.ev_seed, wk.stream));
        HIPCHK(c, hipStreamWaitEvent(wk.st_t, wk.ev_seed, 0));

        // Resident workgroups.  Both kernels take 96 registers -- five waves per SIMD in all -- so the forward kernel must leave
        // the backtrack waves their share: a forward grid that fills the device alone would wait for backtracks that can never start.
        const uint32_t GF = std::max(1u, std::min(ns, c->pipe_f ? c->pipe_f : c->n_cu * 15u));
        const uint32_t GT = std::max(1u, std::min((ns + VC_TG - 1) / VC_TG, c->pipe_t ? c->pipe_t : c->n_cu * 5u));

        VcPipeTraceArgs pt{};
        pt.ta = trace_args(wk);
        pt.ta.group = 1; pt.ta.k0 = 0; pt.ta.hstride = (uint64_t)NC * rowd;
        pt.ta.pairs = wk.d_pairs; pt.ta.npairs = wk.d_npairs; pt.ta.pair_group = 1; pt.ta.pair_k0 = 0; pt.ta.kept = bt->kept;
        pt.ta.shared_table = 0; pt.ta.tab_rows = std::min(NC, kTraceTabRows);
        { uint32_t cb_ = 0; (void)pipe_classes(bt, &pt.ta.cpl_lo, &cb_); }
        pt.p = p;
        pt.g = wk.gr[wk.cur]; pt.STK = c->STK;
        pt.tie_rows = wk.d_tie_rows; pt.tie_cnt = wk.d_tie_cnt; pt.tie_over = wk.d_pairs; pt.tie_over_stride = PC; pt.job_end = wk.d_job_end;
        pt.submask = wk.d_submask; pt.workspace = wk.d_pipe_ws; pt.ws_bytes = (topo_lds + 15u) & ~15u; pt.force_dfs = c->force_dfs ? 1 : 0;
        if ((size_t)GT * pt.ws_bytes > wk.pipe_ws_bytes) return fail(c, VC_ERR_ARG, "persistent pipeline: resolver workspace too small");
        { Timer t(c, KC_TRACE, wk.st_t);
          const uint32_t t_lds = std::max(vc_tracew_lds_bytes(pt.ta.tab_rows, false), 4 * ((NC + 31) / 32 + 1) + 2 * 256 + 16);
          hipLaunchKernelGGL(k_pipe_trace, dim3(GT), dim3(VC_TG * VC_TL), t_lds, wk.st_t, pt); }
        HIPCHK(c, hipEventRecord(wk.ev_t, wk.st_t));

        VcPipeFwdArgs pf{};
        pf.fa = fwd_args(wk);
        pf.fa.group = 1; pf.fa.k0 = 0; pf.fa.mode = 0; pf.fa.hstride = (uint64_t)NC * rowd; pf.fa.do_init = 1;
        pf.fa.tie_over = wk.d_pairs; pf.fa.tie_over_stride = PC;
        pf.aa.b = bt->b; pf.aa.g = wk.gr[wk.cur]; pf.aa.dp = wk.dp; pf.aa.w0 = wk.w0; pf.aa.nslots = ns; pf.aa.NC = NC; pf.aa.EC = EC; pf.aa.layer = 0;
        pf.aa.pairs = wk.d_pairs; pf.aa.npairs = wk.d_npairs; pf.aa.PC = PC; pf.aa.scratch = wk.d_scratch16; pf.aa.ring = (uint32_t)kRing;
        pf.aa.make_rows = 1; pf.aa.kept = bt->kept; pf.aa.tie_n = wk.d_tie_n; pf.aa.redo_n = wk.d_redo_n;
        pf.p = p; pf.submask = wk.d_submask; pf.force_fail_site = 0;
        bool any_partial = false;
        for (uint8_t x : bt->h_layer_partial) any_partial = any_partial || x;
        uint32_t lds = std::max((uint32_t)(kKept ? kKept : 1) * (bt->cpl / 2) * 64u * 4u, add_lds);
        if (any_partial) lds = std::max(lds, vc_rows_sub_lds_bytes(NC, bt->kept));
        lds = (lds + 255u) & ~255u;
        if (lds > kLdsCap) return fail(c, VC_ERR_ARG, "persistent pipeline: %u bytes of LDS per forward wave", lds);
        int rc;
        { Timer t(c, KC_PIPE, wk.stream);
          if ((rc = launch_pipe_fwd(c, bt, wk.stream, pf, std::min(GF, c->CW), lds))) return rc; }
        HIPCHK(c, hipStreamWaitEvent(wk.stream, wk.ev_t, 0));
        return VC_OK;
    }
#endif

    // PruneGraph + LargestSubgraph + its TopologicalSort (window.cpp:318-321,374-383); `more` = a
    // re-alignment round follows, so ask the device how tall the pruned graphs are
    int prune(Work& wk, bool more) {
        const uint32_t ns = wk.ns;
        // after the first round the host knows how large the pruned graphs are (realign() read the maxima
        // back), and a graph only shrinks from round to round: size the LDS images for that, not for NC/EC
        uint32_t NCl = NC, ECl = EC;
        if (wk.pruned_known) {
            NCl = std::min(NC, ((wk.h_maxn[0] + 63u) & ~63u));
            ECl = std::min(EC, ((wk.h_maxn[1] + 63u) & ~63u));
            if (NCl == 0) NCl = 64;
            if (ECl == 0) ECl = 64;
        }
        VcPruneArgs pa{};
        pa.b = bt->b; pa.src = wk.gr[wk.cur]; pa.dst = wk.gr[wk.cur ^ 1]; pa.w0 = wk.w0; pa.nslots = ns; pa.NC = NC; pa.EC = EC;
        pa.NCl = NCl; pa.ECl = ECl;
        // the first prune works on the whole graph (60 KB image at 2 240 nodes): in LDS only two windows fit a CU, and beside a
        // full house of k_fwd waves not even one until eight of them retire; from the HBM workspace every window of the chunk
        // is resident at once: 76 -> 34 ms per 32 768 windows alone, 250 -> 53 ms beside k_fwd, the job + 4 % (VC_PRUNE_HBM=0 / 2:
        // development switch; the later prunes work on graphs a quarter of the size and are faster from LDS).  The same for
        // k_topo (VC_TOPO_HBM=1) and for k_addaln's per-pair notes was measured and does not pay: k_addaln with 1.4 instead of
        // 7 KB of LDS is placed sooner, waits on HBM instead, and its 8 192 waves then sit on the slots k_tracew needs.
        const bool first_hbm = c->prune_hbm && (!wk.pruned_known || c->prune_hbm >= 2) && c->big_ws_stride >= vc_prune_lds_bytes(NCl, ECl);
        const bool pws = first_hbm || vc_prune_lds_bytes(NCl, ECl) > kLdsCap, tws = topo_lds_bytes(NCl, ECl, c->STK, c->MA) > kLdsCap;
        pa.ws = pws ? wk.d_big_ws : nullptr; pa.ws_stride = c->big_ws_stride;
        pa.min_conf = c->prm.min_confidence; pa.min_supp = c->prm.min_support;
        for (uint32_t rep = 0; rep < ((c->dup & 4u) ? 2u : 1u); ++rep) { Timer t(c, KC_PRUNE, wk.stream); hipLaunchKernelGGL(k_prune_lcc, dim3(ns), dim3(64), pws ? 0 : vc_prune_lds_bytes(NCl, ECl), wk.stream, pa); }
        wk.cur ^= 1;
        {
            // before the first re-alignment round the host does not know how small the pruned graphs are: size the LDS image
            // for what they usually are (a chain plus bubbles: ~1.2 nodes and ~1.5 edges per backbone base) and give every
            // workgroup its HBM workspace for the exception
            uint32_t NCt = NCl, ECt = ECl;
            if (!wk.pruned_known && c->big_ws_topo) {
                NCt = std::min(NC, (uint32_t)((bt->max_backbone * 5 / 4 + 127) & ~63u));
                ECt = std::min(EC, (uint32_t)((bt->max_backbone * 2 + 127) & ~63u));
            }
            bool tw = topo_lds_bytes(NCt, ECt, c->STK, c->MA) > kLdsCap;
            if (c->topo_hbm && c->big_ws_stride >= topo_lds_bytes(NCl, ECl, c->STK, c->MA)) { tw = true; NCt = NCl; ECt = ECl; }
            for (uint32_t rep = 0; rep < ((c->dup & 8u) ? 2u : 1u); ++rep) { Timer t(c, KC_TOPO, wk.stream);
              hipLaunchKernelGGL(k_topo, dim3(ns), dim3(64), tw ? 0 : topo_lds_bytes(NCt, ECt, c->STK, c->MA), wk.stream, bt->b, wk.gr[wk.cur], wk.dp, wk.w0, ns, NC, EC, c->STK, -1, 0, bt->ring_pruned, NCt, ECt,
                                 (tw || c->big_ws_topo) ? wk.d_big_ws : nullptr, c->big_ws_stride, tw ? 1 : 0); }
        }
        if (more) {
            HIPCHK(c, hipMemsetAsync(wk.d_maxn, 0, 8, wk.stream));
            hipLaunchKernelGGL(k_max_u32, dim3((ns + 255) / 256), dim3(256), 0, wk.stream, wk.dp.nrows, ns, wk.d_maxn);
            hipLaunchKernelGGL(k_max_u32, dim3((ns + 255) / 256), dim3(256), 0, wk.stream, (const uint32_t*)wk.gr[wk.cur].n_edges, ns, wk.d_maxn + 1);
            HIPCHK(c, hipMemcpyAsync(wk.h_maxn, wk.d_maxn, 8, hipMemcpyDeviceToHost, wk.stream));
        }
        return VC_OK;
    }

    // re-align every sequence to the pruned graph and add its weights (window.cpp:329-372)
    int realign(Work& wk) {
        const uint32_t ns = wk.ns;
        HIPCHK(c, hipStreamSynchronize(wk.stream));          // h_maxn
        uint32_t maxn = wk.h_maxn[0];
        wk.pruned_known = true;
        if (maxn == 0) maxn = 1;
        if (maxn > NC) maxn = NC;
        const uint64_t stride = (uint64_t)maxn * rowd;
        uint32_t group = (uint32_t)std::min<uint64_t>(c->hmat_dwords / (stride * ns), c->rgroup_max);
        if (group == 0) group = 1;
        group = std::min(group, wk.nseq_max);
        if (group >= 8) group &= ~7u;                          // whole waves of k_tracew share one first-in-edge table
        VcFwdArgs fa = fwd_args(wk);
        VcTraceArgs ta = trace_args(wk);
        for (uint32_t k0 = 0; k0 < wk.nseq_max; k0 += group) {
            const uint32_t gsz = std::min(group, wk.nseq_max - k0);
            fa.group = gsz; fa.k0 = k0; fa.mode = 1; fa.hstride = stride;
            if (bt->band) HIPCHK(c, hipMemsetAsync(wk.d_redo_n, 0, 4, wk.stream));
            bool nwonly = true;                                // no partial-span layer among these sequences in any window of the batch?
            for (uint32_t k = std::max(k0, 1u); k < k0 + gsz; ++k) nwonly = nwonly && !(k < bt->h_layer_partial.size() && bt->h_layer_partial[k]);
            ta.cpl_lo = fold_lo(c, bt);
            int rc = launch_fwd(c, bt, wk.stream, fa, ns * gsz, &wk, nwonly);
            if (rc) return rc;
            ta.group = gsz; ta.k0 = k0; ta.hstride = stride; ta.band_chain = 1;      // (mode 1 launches run the forward kernels with KEPT = false)
            ta.pairs = wk.d_rpairs; ta.npairs = wk.d_rnpairs; ta.pair_group = c->max_nseq; ta.pair_k0 = 0;
            launch_trace(wk, ta, ns * gsz, gsz, maxn);
            if ((rc = redo(wk, fa, ta, ns * gsz, maxn, nwonly))) return rc;
        }
        VcAddwArgs wa{};
        wa.b = bt->b; wa.g = wk.gr[wk.cur]; wa.dp = wk.dp; wa.w0 = wk.w0; wa.nslots = ns; wa.NC = NC; wa.EC = EC;
        wa.pairs = wk.d_rpairs; wa.npairs = wk.d_rnpairs; wa.PC = PC; wa.pair_group = c->max_nseq;
        { Timer t(c, KC_ADDW, wk.stream); hipLaunchKernelGGL(k_addw, dim3(ns), dim3(64), 0, wk.stream, wa); }
        return VC_OK;
    }

    // racon-linear overload: exact rank of the finished graph, heaviest bundle, coverage trim (window.cpp:138-171)
    int linear_tail(Work& wk) {
        const uint32_t ns = wk.ns;
        { Timer t(c, KC_TOPO, wk.stream);
          hipLaunchKernelGGL(k_topo, dim3(ns), dim3(64), topo_lds > kLdsCap ? 0 : topo_lds, wk.stream, bt->b, wk.gr[wk.cur], wk.dp, wk.w0, ns, NC, EC, c->STK, -1, 0, bt->ring_pruned, NC, EC,
                             topo_lds > kLdsCap ? wk.d_big_ws : nullptr, c->big_ws_stride, topo_lds > kLdsCap ? 1 : 0); }
        VcConsArgs ca{};
        ca.b = bt->b; ca.g = wk.gr[wk.cur]; ca.dp = wk.dp; ca.w0 = wk.w0; ca.nslots = ns; ca.NC = NC; ca.EC = EC;
        ca.trim = c->prm.trim; ca.window_type = c->prm.window_type;
        ca.ws = cons_lds > kLdsCap ? wk.d_big_ws : nullptr; ca.ws_stride = c->big_ws_stride;
        { Timer t(c, KC_CONS, wk.stream); hipLaunchKernelGGL(k_consensus, dim3(ns), dim3(64), cons_lds > kLdsCap ? 0 : cons_lds, wk.stream, ca); }
        wk.active = false;
        return VC_OK;
    }

    // final local alignment of the backbone + corrected sequence (window.cpp:391-394)
    int finish(Work& wk) {
        const uint32_t ns = wk.ns;
        VcFwdArgs fa = fwd_args(wk);
        fa.group = 1; fa.k0 = 0; fa.mode = 2; fa.hstride = (uint64_t)NC * rowd; fa.band = 0;
        int rc = launch_fwd(c, bt, wk.stream, fa, ns, &wk);
        if (rc) return rc;
        VcTraceArgs ta = trace_args(wk);
        ta.group = 1; ta.k0 = 0; ta.hstride = fa.hstride; ta.band = 0;
        ta.pairs = wk.d_pairs; ta.npairs = wk.d_npairs; ta.pair_group = 1; ta.pair_k0 = 0;
        launch_trace(wk, ta, ns, 1, NC);
        VcFinishArgs fn{};
        fn.b = bt->b; fn.g = wk.gr[wk.cur]; fn.dp = wk.dp; fn.w0 = wk.w0; fn.nslots = ns; fn.NC = NC;
        fn.pairs = wk.d_pairs; fn.npairs = wk.d_npairs; fn.PC = PC;
        { Timer t(c, KC_FINISH, wk.stream); hipLaunchKernelGGL(k_finish, dim3(ns), dim3(64), 0, wk.stream, fn); }
        wk.active = false;
        return VC_OK;
    }

    // every phase of one chunk on its stream, in order (window.cpp:176-428 for each window of the chunk)
    int run_chunk(Work& wk, uint32_t w0, uint32_t ns) {
        int rc;
        begin(wk, w0, ns);
        if (wk.layers && pipe_ok()) { if ((rc = build_pipe(wk))) return rc; }
        else if ((rc = build_loop(wk))) return rc;
        if (!wk.layers) { wk.active = false; return VC_OK; }
        if (c->prm.mode == 1) return linear_tail(wk);
        for (uint32_t r = 0; r < c->prm.num_prune; ++r) {
            const bool more = r + 1 < c->prm.num_prune;
            if ((rc = prune(wk, more))) return rc;
            if (!more) break;
            if ((rc = realign(wk))) return rc;
        }
        return finish(wk);
    }
};

// Host thread of chunk stream s.  It takes the next chunk of the oldest queued batch that still has one -- when a batch has none
// left it goes straight on to the batch queued behind it, so the device sees no gap between batches -- queues every phase of the
// chunk on its stream, and records the batch's event of that stream behind it.
void stream_worker(vc_ctx* c, uint32_t s) {
    (void)hipSetDevice(c->device);
    std::unique_lock<std::mutex> lk(c->qmu);
    for (;;) {
        Batch* bt = nullptr;
        uint32_t k = 0;
        for (Batch* q : c->runq)
            if (s < q->n_streams && q->next_chunk < q->n_chunks) { bt = q; k = q->next_chunk++; break; }
        if (!bt) {
            if (c->stop) return;
            c->qcv.wait(lk);
            continue;
        }
        const bool skip = bt->run_rc != VC_OK;               // a chunk of this run failed: the rest is not started
        if (k == 0) bt->first_stream = s;
        lk.unlock();
        int rc = VC_OK;
        if (!skip) {
            const uint32_t CW = bt->cw_run, nw = bt->b.n_windows, w0 = k * CW;
            rc = bt->pl->run_chunk(c->works[s], w0, std::min(CW, nw - w0));
            if (hipEventRecord(bt->done_ev[s], c->streams[s]) != hipSuccess && rc == VC_OK) rc = fail(c, VC_ERR_HIP, "hipEventRecord failed on chunk stream %u", s);
        }
        lk.lock();
        if (!skip) bt->ev_rec[s] = true;
        if (rc && bt->run_rc == VC_OK) bt->run_rc = rc;
        if (++bt->chunks_done == bt->n_chunks) {
            for (auto it = c->runq.begin(); it != c->runq.end(); ++it) if (*it == bt) { c->runq.erase(it); break; }
            bt->queued = false;
            c->qcv.notify_all();
        }
    }
}

// the launch-time parameters of a run: everything derives from the workspace capacities and the batch's own classes
Plan* make_plan(vc_ctx* c, Batch* bt) {
    Plan* pl = new Plan();
    pl->c = c; pl->bt = bt; pl->NC = c->NC; pl->EC = c->EC; pl->PC = c->PC; pl->cpl = bt->cpl;
    pl->topo_lds = topo_lds_bytes(c->NC, c->EC, c->STK, c->MA);
    pl->prune_lds = vc_prune_lds_bytes(c->NC, c->EC);
    pl->add_lds = std::max(2 * c->PC + 2 * (c->PC - c->NC) + 64, bt->kept ? vc_kept_lds_bytes(c->NC) : 0u);
    pl->rows_lds = 0;
    pl->cons_lds = vc_cons_lds_bytes(c->NC, c->EC);
    pl->rowd = 64ull * (bt->packed ? (uint64_t)vc_nds((int)c->ws_cpl) : c->ws_cpl / 2);
    return pl;
}

}  // namespace

extern "C" {

const char* vc_last_error(const vc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vc_create(vc_ctx** out, const vc_params* p) {
    if (!out || !p) return fail(nullptr, VC_ERR_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VC_ERR_NO_DEVICE, "no HIP device visible (%s); libvechat_hip has no CPU fallback",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, VC_ERR_ARG, "device %d out of range (%d visible)", p->device, ndev);
    if (p->mode != 0 && p->mode != 1) return fail(nullptr, VC_ERR_ARG, "mode %d unknown (0 haplotype overload, 1 racon-linear overload)", p->mode);
    if (p->num_prune == 0) return fail(nullptr, VC_ERR_ARG, "num_prune must be >= 1");
    if (p->gap >= 0 || p->sw_gap >= 0) return fail(nullptr, VC_ERR_ARG, "gap penalties must be negative");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) return fail(nullptr, VC_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VC_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", p->device, prop.gcnArchName);
    vc_ctx* c = new vc_ctx();
    c->prm = *p;
    c->device = p->device;
    // vc_params.n_streams = 0: the batch decides (vc_submit).  Round 4, after the backtrack got lighter: 8 chunk streams beat 4 by 4-8 %
    // from 16 384 windows up (config C, 100 000 windows: 32.4 -> 34.4 k windows/s), are level at 8 192 and lose at 4 000, where a chunk
    // of 500 windows no longer fills a launch
    c->auto_streams = p->n_streams == 0;
    c->n_streams = p->n_streams ? std::min<uint32_t>(p->n_streams, kMaxStreams) : 4;
    c->streams_made = c->auto_streams ? kAutoStreamsMany : c->n_streams;
    c->force_dfs = getenv("VC_RESOLVE_FORCE_DFS") != nullptr;
    if (const char* d = getenv("VC_DUP")) c->dup = (uint32_t)std::atoi(d);
    c->fold = getenv("VC_NO_FOLD") == nullptr;
    if (const char* d = getenv("VC_INLINE_REDO")) c->inline_redo = std::atoi(d) != 0;
#ifdef VC_EXPERIMENTS
    if (const char* d = getenv("VC_BAND_RAW")) c->band_raw = std::atoi(d) != 0;
#endif
    if (const char* d = getenv("VC_TRACE_TL")) c->trace_tl = std::atoi(d) == 16 ? 16u : 8u;
    if (const char* d = getenv("VC_TRACEB")) c->trace_block = std::atoi(d) != 0;
    if (const char* d = getenv("VC_HOST_THREADS")) c->host_threads = std::atoi(d) != 0;      // development: 0 = one host thread walks the streams in lockstep
    c->trace_wave = getenv("VC_TRACE_THREAD") == nullptr;      // development switch: the thread-per-alignment backtrack
    if (const char* d = getenv("VC_DT")) c->dt = std::atoi(d) != 0;
    if (const char* d = getenv("VC_AUTO_ARENA")) c->auto_arena = std::atoi(d) != 0;
#ifdef VC_EXPERIMENTS
    if (const char* d = getenv("VC_PIPE")) c->pipe = std::atoi(d) != 0;
#endif
    if (const char* d = getenv("VC_PIPE_F")) c->pipe_f = (uint32_t)std::atoi(d);
    if (const char* d = getenv("VC_PIPE_T")) c->pipe_t = (uint32_t)std::atoi(d);
    if (const char* d = getenv("VC_PIPE_PATIENCE")) c->pipe_patience_s = std::min(40u, std::max(1u, (uint32_t)std::atoi(d)));
    c->n_cu = (uint32_t)prop.multiProcessorCount;
    if (const char* d = getenv("VC_PRUNE_HBM")) c->prune_hbm = std::atoi(d);
    if (const char* d = getenv("VC_TOPO_HBM")) c->topo_hbm = std::atoi(d) != 0;
    if (hipSetDevice(c->device) != hipSuccess) { delete c; return fail(nullptr, VC_ERR_HIP, "hipSetDevice failed"); }
    // The chunk streams must run CONCURRENTLY.  HIP multiplexes streams of one priority onto a small pool of
    // hardware queues (GPU_MAX_HW_QUEUES, default 4) round-robin, so two of ours can land on the same queue
    // and serialise -- it depends on how many streams the process created before (measured: with RCCL
    // initialised first the 2-stream rate dropped from 19.9 k to 15.9 k windows/s).  Streams of different
    // priority never share a hardware queue, so the chunk streams cycle through the priority levels.
    // (the streams themselves belong to the process: pooled_stream)
    for (uint32_t s = 0; s < c->streams_made; ++s) {
        c->streams[s] = pooled_stream(c->device, s, false);
        if (!c->streams[s]) { delete c; return fail(nullptr, VC_ERR_HIP, "hipStreamCreate failed"); }
        c->works[s].stream = c->streams[s];
        if (hipHostMalloc((void**)&c->works[s].h_maxn, 64) != hipSuccess) { delete c; return fail(nullptr, VC_ERR_HIP, "hipHostMalloc failed"); }
        Work& wk = c->works[s];
        if (hipEventCreateWithFlags(&wk.ev_seed, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&wk.ev_t, hipEventDisableTiming) != hipSuccess) {
            delete c; return fail(nullptr, VC_ERR_HIP, "event creation failed");
        }
    }
    // the context's own stream for its copies, fills and small kernels: those must be able to run beside another context's chunks
    {
        int own_prio = 0;                                  // development: VC_OWN_PRIO = 0 / 1 / 2 picks the priority level of this stream
        if (const char* d = getenv("VC_OWN_PRIO")) own_prio = stream_priority((uint32_t)std::atoi(d), 0u);
        if (hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, own_prio) != hipSuccess) { delete c; return fail(nullptr, VC_ERR_HIP, "hipStreamCreate failed"); }
    }
    c->stream = c->own_stream;
    // lookup tables from this host's libm, like the reference computes them (graph.cpp:169, window.cpp:235)
    uint32_t lw[256]; double ld[256];
    vc_weight_lut(lw);
    for (int ch = 0; ch < 256; ++ch) ld[ch] = 1 - pow(10, (33 - (int)(signed char)ch) / 10.0);
    if (dalloc(c, c->allocs, &c->d_lut_w, 256) || dalloc(c, c->allocs, &c->d_lut_d, 256) || dalloc(c, c->allocs, &c->d_stat, VC_STAT_WORDS)
#ifdef VC_EXPERIMENTS
        || dalloc(c, c->allocs, &c->d_pipe_abort, 64) || dalloc(c, c->allocs, &c->d_pipe_prof, VC_PP_TOTAL)
#endif
        ) {
        g_create_error = c->err; vc_destroy(c); return VC_ERR_HIP;
    }
    (void)hipMemcpy(c->d_lut_w, lw, sizeof(lw), hipMemcpyHostToDevice);
    (void)hipMemcpy(c->d_lut_d, ld, sizeof(ld), hipMemcpyHostToDevice);
    for (Batch& bt : c->bt)
        for (uint32_t s = 0; s < kMaxStreams; ++s)
            if (hipEventCreateWithFlags(&bt.done_ev[s], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { g_create_error = "event creation failed"; vc_destroy(c); return VC_ERR_HIP; }
    c->stats.n_classes = KC_N;
    for (int i = 0; i < KC_N; ++i) std::snprintf(c->stats.names[i], sizeof(c->stats.names[i]), "%s", kClassNames[i]);
    *out = c;
    return VC_OK;
}

void vc_destroy(vc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drain(c);
    {
        std::lock_guard<std::mutex> lk(c->qmu);
        c->stop = true;
    }
    c->qcv.notify_all();
    for (uint32_t s = 0; s < c->workers_made; ++s) if (c->workers[s].joinable()) c->workers[s].join();
    free_workspaces(c);
    for (auto& sg : c->arena) (void)hipFree(sg.p);
    c->arena.clear();
    for (Batch& bt : c->bt) {
        for (auto& sl : bt.slots) if (sl.p) (void)hipFree(sl.p);
        for (hipEvent_t e : bt.done_ev) if (e) (void)hipEventDestroy(e);
        delete bt.pl;
    }
    free_list(c->allocs);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]); if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]); }
    for (uint32_t s = 0; s < kMaxStreams; ++s) {
        if (c->works[s].h_maxn) (void)hipHostFree(c->works[s].h_maxn);
        for (hipEvent_t e : {c->works[s].ev_seed, c->works[s].ev_t}) if (e) (void)hipEventDestroy(e);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

void* vc_stream(vc_ctx* c) { return c ? (void*)c->stream : nullptr; }

int vc_set_profile(vc_ctx* c, int profile) {
    if (!c || profile < 0 || profile > 2) return VC_ERR_ARG;
    drain(c);                            // the chunk threads of a running batch read prm.profile and write the event records
    c->prm.profile = profile;
    return VC_OK;
}

int vc_has_experiments(void) {
#ifdef VC_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int vc_set_pipeline(vc_ctx* c, int on, uint32_t forward_waves, uint32_t backtrack_waves) {
    if (!c) return VC_ERR_ARG;
#ifndef VC_EXPERIMENTS
    if (on) return fail(c, VC_ERR_ARG, "the persistent build pipeline is an experiment: this library was built without -DVC_EXPERIMENTS");
#endif
    drain(c);
    c->pipe = on != 0; c->pipe_f = forward_waves; c->pipe_t = backtrack_waves;
    return VC_OK;
}

int vc_reserve(vc_ctx* c, uint64_t bytes) {
    if (!c) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    drain(c);
    free_workspaces(c);
    unstage(c);
    return make_arena(c, bytes);
}

int vc_release(vc_ctx* c) {
    if (!c) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    drain(c);
    free_workspaces(c);
    unstage(c);
    free_arena(c);
    return VC_OK;
}

// The polishing parameters of a live context: what differs between the rounds of a driver run (scripts/vechat:59-93: round 1 `-f -p -d D -s S`,
// round 2 `-f [-u]`) -- overload, thresholds, prune rounds, trim, window type, scores.  Device, capacities, budget and streams stay as
// created; workspaces and arena stay where they are (that is the point: one warm context for both rounds and every --split chunk).
int vc_set_polish_params(vc_ctx* c, const vc_params* p) {
    if (!c || !p) return VC_ERR_ARG;
    if (p->mode != 0 && p->mode != 1) return fail(c, VC_ERR_ARG, "mode %d unknown (0 haplotype overload, 1 racon-linear overload)", p->mode);
    if (p->num_prune == 0) return fail(c, VC_ERR_ARG, "num_prune must be >= 1");
    if (p->gap >= 0 || p->sw_gap >= 0) return fail(c, VC_ERR_ARG, "gap penalties must be negative");
    if (p->window_type != 0 && p->window_type != 1) return fail(c, VC_ERR_ARG, "window_type %d unknown", p->window_type);
    drain(c);
    for (Batch& b : c->bt) if (!b.ran || b.collected) b.have = false;       // (a batch staged under the old scores was planned for them)
    c->prm.match = p->match; c->prm.mismatch = p->mismatch; c->prm.gap = p->gap;
    c->prm.sw_match = p->sw_match; c->prm.sw_mismatch = p->sw_mismatch; c->prm.sw_gap = p->sw_gap;
    c->prm.min_confidence = p->min_confidence; c->prm.min_support = p->min_support; c->prm.num_prune = p->num_prune;
    c->prm.mode = p->mode; c->prm.trim = p->trim; c->prm.window_type = p->window_type;
    return VC_OK;
}

int vc_set_window_type(vc_ctx* c, int window_type) {
    if (!c || (window_type != 0 && window_type != 1)) return VC_ERR_ARG;
    drain(c);
    c->prm.window_type = window_type;
    return VC_OK;
}

int vc_submit(vc_ctx* c, const vc_batch* hb) {
    if (!c || !hb) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t nw = hb->n_windows;
    if (nw == 0) return fail(c, VC_ERR_ARG, "empty batch");
    if (!hb->win_seq_off || !hb->seq_off || !hb->seq_begin || !hb->seq_end || !hb->seq_has_qual || !hb->bases ||
        !hb->quals || !hb->win_fasta) return fail(c, VC_ERR_ARG, "null array in batch");
    const uint64_t nseq = hb->win_seq_off[nw];
    const uint64_t nbytes = hb->seq_off[nseq];
    // validation (what createWindow / add_layer enforce, window.cpp:22-27,56-67)
    uint32_t max_layers = 0, max_len = 0, max_nseq = 0, min_len = 0xFFFFFFFFu, max_backbone = 0;
    uint64_t need_nodes = 0;
    std::vector<uint8_t> layer_partial;
    std::vector<uint8_t> pre(nw, 0);          // windows outside the device envelope: reported, not run
    bool any_pre = false;
    for (int pass = 0; pass < 2; ++pass) {
    max_layers = max_len = max_nseq = max_backbone = 0; min_len = 0xFFFFFFFFu; need_nodes = 0; any_pre = false;
    std::fill(layer_partial.begin(), layer_partial.end(), 0);
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t s0 = hb->win_seq_off[w], s1 = hb->win_seq_off[w + 1];
        if (s1 <= s0) return fail(c, VC_ERR_ARG, "window %u has no backbone", w);
        const uint64_t L = hb->seq_off[s0 + 1] - hb->seq_off[s0];
        if (L == 0 || L >= 65535) return fail(c, VC_ERR_ARG, "window %u: backbone length %llu unsupported", w, (unsigned long long)L);
        max_backbone = std::max<uint32_t>(max_backbone, (uint32_t)L);
        if (!hb->seq_has_qual[s0]) return fail(c, VC_ERR_ARG, "window %u: backbone needs a quality string (dummy '!' for FASTA targets)", w);
        uint64_t sum = 0;
        for (uint32_t s = s0; s < s1; ++s) {
            const uint64_t len = hb->seq_off[s + 1] - hb->seq_off[s];
            if (len == 0 || len >= 65535) return fail(c, VC_ERR_ARG, "window %u: sequence length %llu unsupported", w, (unsigned long long)len);
            if (s > s0) {
                const uint32_t b = hb->seq_begin[s], e2 = hb->seq_end[s];
                if (b >= e2 || b > L || e2 >= L) return fail(c, VC_ERR_ARG, "window %u: invalid layer positions (%u,%u)", w, b, e2);
                sum += len;
                const uint32_t offset = (uint32_t)(0.01 * (double)L);          // window.cpp:212,253-254
                const bool full = b < offset && e2 > L - offset;
                const uint32_t j = s - s0;
                if (layer_partial.size() <= j) layer_partial.resize(j + 1, 0);
                if (!full) layer_partial[j] = 1;
            }
            if (!pre[w]) {
                max_len = std::max<uint32_t>(max_len, (uint32_t)len);
                min_len = std::min<uint32_t>(min_len, (uint32_t)len);
            }
        }
        if (pre[w]) { any_pre = true; continue; }
        max_layers = std::max(max_layers, s1 - s0 - 1);
        max_nseq = std::max(max_nseq, s1 - s0);
        // graph growth saturates with depth: nodes ~ L + c * mean_layer_len * depth^0.55 (measured on
        // 8..128-read PacBio/ONT-profile windows; windows that still outgrow it report VC_WIN_OVERFLOW)
        const double depth = (double)(s1 - s0 - 1);
        const double est = depth > 0 ? 0.33 * ((double)sum / depth) * std::pow(depth, 0.55) : 0.0;
        need_nodes = std::max<uint64_t>(need_nodes, L + (uint64_t)std::ceil(est) + 64);
    }
    // k_addaln keeps two 16-bit notes per alignment pair in LDS (2 * PC + 2 * longest sequence bytes): a layer too long for that
    // beside this batch's graph capacity takes ITS window out -- reported as VC_WIN_OVERFLOW like any other capacity limit --
    // instead of failing the whole batch (one 38 k-base layer used to do that)
    if (pass == 0) {
        const uint64_t nc0 = c->prm.max_nodes ? c->prm.max_nodes : std::min<uint64_t>((need_nodes + 63) & ~63ull, 59968);
        const int64_t allowed = ((int64_t)kLdsCap - 64 - 48 - 2 * (int64_t)std::max<uint64_t>(nc0, !c->have_ws ? 0 : c->NC)) / 4;
        if ((int64_t)max_len <= allowed) break;
        for (uint32_t w = 0; w < nw; ++w)
            for (uint32_t q = hb->win_seq_off[w]; q < hb->win_seq_off[w + 1]; ++q)
                if ((int64_t)(hb->seq_off[q + 1] - hb->seq_off[q]) > std::max<int64_t>(allowed, 0)) pre[w] = VC_WIN_OVERFLOW;
    }
    }
    // Which of the two batch slots this batch takes.  The last one again when it is neither running nor holding results nobody has
    // collected (the serial caller: submit, run, collect, submit ...) -- else the other one, so that this batch is copied in while
    // that one runs (a run still in flight in the slot taken is waited for; results not collected from it are dropped).
    Batch* bt = c->cur ? c->cur : &c->bt[0];
    {
        bool busy;
        { std::lock_guard<std::mutex> lk(c->qmu); busy = bt->queued; }
        if (busy || (bt->ran && !bt->collected)) bt = bt == &c->bt[0] ? &c->bt[1] : &c->bt[0];
    }
    int rc;
    if ((rc = wait_batch(c, bt))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bt->have = false; bt->ran = false; bt->collected = false;
    uint32_t want_streams = c->n_streams;
    // (a context whose workspaces already serve the larger number keeps it for medium batches: in a stream of batches queued behind each
    // other a batch of 8 192 windows runs beside its neighbours, and on four streams it would leave the other four to them alone)
    if (c->auto_streams) want_streams = (nw >= kAutoStreamsFrom || (c->have_ws && c->n_streams == kAutoStreamsMany && nw >= 8192)) ? kAutoStreamsMany : kAutoStreamsFew;
    VcBatchDev& b = bt->b;
    b = VcBatchDev{};
    b.n_windows = nw;
    const bool tm_s = getenv("VC_TIME_SUBMIT") != nullptr;          // development: phases of a submit on stderr
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ts_0 = now_s();
    uint32_t* d_wso; uint64_t* d_so; uint32_t *d_sb, *d_se; uint8_t *d_hq, *d_ba, *d_qu, *d_wf;
    if ((rc = salloc(c, bt, 0, &d_wso, nw + 1)) || (rc = salloc(c, bt, 1, &d_so, nseq + 1)) ||
        (rc = salloc(c, bt, 2, &d_sb, nseq)) || (rc = salloc(c, bt, 3, &d_se, nseq)) ||
        (rc = salloc(c, bt, 4, &d_hq, nseq)) || (rc = salloc(c, bt, 5, &d_ba, nbytes + 16)) ||
        (rc = salloc(c, bt, 6, &d_qu, nbytes + 16)) || (rc = salloc(c, bt, 7, &d_wf, nw)) ||
        (rc = salloc(c, bt, 8, &b.win_avg, nw)) || (rc = salloc(c, bt, 9, &b.status, nw)) ||
        (rc = salloc(c, bt, 10, &b.cons_len, nw)) || (rc = salloc(c, bt, 11, &b.errinfo, nw)))
        return rc;
    const double ts_1 = now_s();
    HIPCHK(c, hipMemcpyAsync(d_wso, hb->win_seq_off, (nw + 1) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_so, hb->seq_off, (nseq + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_sb, hb->seq_begin, nseq * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_se, hb->seq_end, nseq * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_hq, hb->seq_has_qual, nseq, hipMemcpyHostToDevice, c->stream));
    if ((rc = h2d(c, d_ba, hb->bases, nbytes)) || (rc = h2d(c, d_qu, hb->quals, nbytes))) return rc;
    HIPCHK(c, hipMemcpyAsync(d_wf, hb->win_fasta, nw, hipMemcpyHostToDevice, c->stream));
    b.win_seq_off = d_wso; b.seq_off = d_so; b.seq_begin = d_sb; b.seq_end = d_se; b.seq_has_qual = d_hq;
    b.bases = d_ba; b.quals = d_qu; b.win_fasta = d_wf;
    b.lut_w = c->d_lut_w; b.lut_d = c->d_lut_d;
    bt->h_win_seq_off.assign(hb->win_seq_off, hb->win_seq_off + nw + 1);
    bt->h_layer_partial = layer_partial;
    bt->h_layer_partial.resize(std::max<size_t>(layer_partial.size(), max_nseq) + 2, 0);      // indexed by layer up to the deepest window of the batch
    bt->max_layers = max_layers; bt->max_len = max_len;
    bt->h_pre_status = any_pre ? pre : std::vector<uint8_t>();
    if (max_len == 0) { max_len = 1; min_len = 1; }              // every window was outside the envelope

    // alphabet of the batch -> entries per aligned list (an aligned group holds distinct bytes, graph.cpp:258-277)
    uint32_t MA = 4;
    {
        uint32_t* d_mask = nullptr;
        if ((rc = salloc(c, bt, 15, &d_mask, 8))) return rc;
        HIPCHK(c, hipMemsetAsync(d_mask, 0, 32, c->stream));
        hipLaunchKernelGGL(k_byte_presence, dim3(1024), dim3(256), 0, c->stream, (const uint8_t*)d_ba, nbytes, d_mask);
        uint32_t hm[8];
        HIPCHK(c, hipMemcpyAsync(hm, d_mask, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        uint32_t distinct = 0;
        for (int k = 0; k < 8; ++k) distinct += (uint32_t)__builtin_popcount(hm[k]);
        if (distinct > 5) MA = (distinct - 1 + 1) & ~1u;
        // A / C / G / T only?  Then the widest classes may build their profiles on the fly (VcFwdArgs::lean) where mismatch - gap == -1
        uint32_t other[8];
        for (int k = 0; k < 8; ++k) other[k] = hm[k];
        for (unsigned char ch : {'A', 'C', 'G', 'T'}) other[ch >> 5] &= ~(1u << (ch & 31));
        bool pure = true;
        for (int k = 0; k < 8; ++k) pure = pure && other[k] == 0;
        bt->lean = (pure && c->prm.mismatch - c->prm.gap == -1 ? 1u : 0u) | (pure && c->prm.sw_mismatch - c->prm.sw_gap == -1 ? 2u : 0u);
    }

    if (tm_s) fprintf(stderr, "vc_submit(%u windows): batch buffers %.3f s, copies in + byte census %.3f s\n", nw, ts_1 - ts_0, now_s() - ts_1);
    // A large batch of ordinary windows with no arena yet: make it now, once (vc_reserve's work; a caller that knows what is coming calls
    // that while it still parses).  Every later batch, whatever its shape, is then laid out inside it -- a stream of batches that grows
    // (the host-to-host loop starts small so that the device starts early) re-created ~90 GiB of workspaces twice on the way, seconds each.
    // Small batches (tests, single targets) and windows beyond 2 kb (their chunk size IS the memory: section "big alignments" below)
    // keep the piece-by-piece workspaces.
    if (c->arena.empty() && c->auto_arena && nw >= 4096 && max_len <= 2048) {
        drain(c);
        free_workspaces(c);
        for (Batch& ob : c->bt) if (&ob != bt && (!ob.ran || ob.collected)) ob.have = false;      // (a batch staged on the old workspaces cannot run on the new ones)
        if ((rc = make_arena(c, 0))) return rc;
    }
    // capacities
    uint32_t NC = c->prm.max_nodes ? c->prm.max_nodes : (uint32_t)std::min<uint64_t>(need_nodes, 59968);
    NC = (NC + 63) & ~63u;
    if (!c->prm.max_nodes && NC > 59968) NC = 59968;
    // Workspaces are grow-only across batches: a batch whose own estimate is a little smaller than its predecessor's keeps
    // the capacities that exist (a few percent of difference would otherwise re-create ~100 GB of buffers, seconds per
    // batch).  Results do not depend on capacities; pinned capacities (max_nodes / max_edges) are taken literally.
    const bool have_ws = c->have_ws;
    if (have_ws && !c->prm.max_nodes) {
        // (a batch that needs a little more than the workspaces hold gets a sixteenth on top: a stream of batches whose estimates
        // creep upwards must not re-create the workspaces -- and wait for the batch in flight -- every time)
        if (NC > c->NC) NC = std::min<uint32_t>(std::max(NC, (c->NC + c->NC / 16 + 63) & ~63u), 59968);
        NC = std::max(NC, c->NC);
    }
    if (have_ws) { MA = std::max(MA, c->MA); max_nseq = std::max(max_nseq, c->max_nseq); }
    const uint32_t ws_max_len = have_ws ? std::max(max_len, c->ws_max_len) : max_len;
    uint32_t EC = c->prm.max_edges ? c->prm.max_edges : (uint32_t)std::min<uint64_t>((uint64_t)(2.6 * NC), 32000);
    EC = (EC + 63) & ~63u;
    if (EC > 32000) EC = 32000;                     // k_prune_lcc keeps 2E adjacency offsets in 16 bits
    if (have_ws && !c->prm.max_edges) EC = std::max(EC, c->EC);
    if (NC > 59968) return fail(c, VC_ERR_ARG, "max_nodes %u exceeds the 16-bit id space (59968)", NC);
    bt->cpl = pick_cpl(std::min(max_len, kMaxColumns));                  // width classes of THIS batch (kernel selection)
    bt->packed = vc_row_packed(c->prm.match, c->prm.mismatch, c->prm.gap, (int)bt->cpl) &&
                 vc_row_packed(c->prm.sw_match, c->prm.sw_mismatch, c->prm.sw_gap, (int)bt->cpl);
    bt->cpl_min = pick_cpl(std::min(min_len, kMaxColumns));
    const uint32_t cpl = pick_cpl(std::min(ws_max_len, kMaxColumns));   // width class the matrices are sized for
    const uint32_t lds_cap = kLdsCap;
    // graph images that do not fit the LDS are worked on in an HBM workspace (slower, not refused)
    uint32_t big = 0;
    if (topo_lds_bytes(NC, EC, c->STK, MA) > lds_cap) big = std::max(big, topo_lds_bytes(NC, EC, c->STK, MA));
    if (vc_prune_lds_bytes(NC, EC) > lds_cap) big = std::max(big, vc_prune_lds_bytes(NC, EC));
    if (c->prm.mode == 1 && vc_cons_lds_bytes(NC, EC) > lds_cap) big = std::max(big, vc_cons_lds_bytes(NC, EC));
    c->big_ws_topo = c->prm.mode == 0;                    // k_topo's LDS image of the first pruned graphs is sized optimistically: its fallback lives here
    if (c->big_ws_topo) big = std::max(big, std::max(topo_lds_bytes(NC, EC, c->STK, MA), vc_prune_lds_bytes(NC, EC)));   // (and the first prune's image, see Plan::prune)
    bt->max_backbone = max_backbone;
    big = (big + 255u) & ~255u;
    const uint32_t PC = NC + ws_max_len + 8;

    // chunk size from the scratch budget (split over the streams)
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    free_b += c->chunk_bytes;            // our own workspaces are reusable: the plan must not depend on whether they exist yet
    // the workspaces of more streams than this batch wants are kept (it simply uses the first few): a small batch in a stream of
    // large ones must not re-create them
    const uint32_t S = c->have_ws ? std::max(want_streams, c->n_streams) : want_streams;
    // default budget: 60 % of what is free, capped (default_budget) -- config C runs at 91 % of its 220-GiB rate with 64 GiB, 97.7 % with 96
    // (chunks of 4 096 windows) and at 86 % with 32 GiB (2 048), so holding more than that buys nothing (profiles/r3c_footprint.txt)
    uint64_t budget = (c->prm.scratch_bytes ? c->prm.scratch_bytes : default_budget(free_b)) / S;
    const uint64_t budget_default = budget;
    if (!c->arena.empty()) {                             // the arena IS the budget: a segment per stream (less the padding between its pieces)
        const uint64_t per_seg = (S + c->arena.size() - 1) / c->arena.size();          // workspaces that share a segment
        const uint64_t seg = c->arena[0].bytes - std::min<size_t>(c->arena[0].bytes, (size_t)per_seg << 20);
        budget = seg / per_seg;
    }
    const uint64_t rowd = 64ull * (bt->packed ? (uint64_t)vc_nds((int)cpl) : cpl / 2);      // dwords per stored row: byte-packed (NDS per lane) or raw int16 pairs
    const uint64_t per_slot_fixed = 2ull * (NC * (1 + 8 + 1 + 2ull * MA + 6 + 16) + EC * 12ull + 8) + (NC * (16ull + 16 + 2 + 2 + 16 + 2 + 1) + EC * 2ull + 36) +
                                    PC * 4ull + 4 + (uint64_t)max_nseq * (PC * 4ull + 4) + big;
    // Can any alignment of this batch leave the packed-int16 kernel's envelope (vc_fwd_body's check, vc_int16_ok, on the worst
    // case the capacities allow)?  Then k_fwd_wide and its int32 matrices are needed.
    bool maybe_wide = ws_max_len > kMaxColumns || (have_ws && c->wcols);
    if (!vc_int16_ok(c->prm.match, c->prm.mismatch, c->prm.gap, NC, cpl, true) ||
        !vc_int16_ok(c->prm.sw_match, c->prm.sw_mismatch, c->prm.sw_gap, NC, cpl, false)) maybe_wide = true;
    if (cpl >= 32 && bt->lean != 3u) maybe_wide = true;       // classes of 32+ columns per lane exist in the lean form only: other alphabets / scores take k_fwd_wide
    const uint32_t wcols = maybe_wide ? ((ws_max_len + 64 * VC_WIDE_CPL - 1) / (64 * VC_WIDE_CPL)) * (64 * VC_WIDE_CPL) : 0;
    // whole rows, + a quarter for the band (byte-packed rows; raw int16 rows -- wide classes, unusual scores -- only with VC_BAND_RAW=1)
    const uint64_t per_job = NC * rowd * ((bt->packed || c->band_raw) ? 5 : 4) + NC * 2 + 24 + 2 * VC_MAXTIE + 8 + (maybe_wide ? (uint64_t)NC * wcols * 4 + NC * 4ull : 0ull);
    // big alignments (3 kb reads: 58 MB of raw rows each; the int32 matrices of k_fwd_wide: 86 MB more): there the chunk size IS the
    // budget, and the 96-GiB cap would leave a few hundred alignments per stream -- take the 60 % whole
    if (!c->prm.scratch_bytes && c->arena.empty() && (per_slot_fixed + per_job) * 1024ull > budget) budget = std::max<uint64_t>(budget, (uint64_t)(free_b * 0.6) / S);
    uint32_t CW = c->prm.chunk_windows ? c->prm.chunk_windows : 8192;
    CW = std::min(CW, (nw + S - 1) / S);
    if (CW == 0) CW = 1;
    // a reservation that cannot hold a sensible chunk of this batch does not bind the plan: the usual budget applies and what does
    // not fit the arena is allocated piece by piece
    if (!c->arena.empty() && (per_slot_fixed + per_job) * std::min(CW, 64u) > budget) budget = std::max(budget, budget_default);
    if ((per_slot_fixed + per_job) * CW > budget) {          // as many windows per chunk as the budget holds (whole waves of 64 where it can)
        CW = (uint32_t)std::max<uint64_t>(budget / (per_slot_fixed + per_job), 1);
        if (CW > 64) CW &= ~63u;
    }
    if ((per_slot_fixed + per_job) * CW > budget) return fail(c, VC_ERR_ARG, "scratch budget %llu too small", (unsigned long long)budget);
    // spare matrix space lets re-alignment rounds run several sequences of a window per launch
    uint64_t spare = budget - (per_slot_fixed + per_job) * CW;
    uint32_t group_max = 1 + (uint32_t)std::min<uint64_t>(spare / (per_job * CW), 7);
    if (group_max > max_nseq) group_max = max_nseq;

    const bool same = have_ws && S == c->n_streams && c->ws_packed == bt->packed && c->wcols == wcols && c->MA == MA && c->NC == NC && c->EC == EC && c->CW >= CW && c->ws_cpl == cpl && c->PC == PC &&
                      c->big_ws_stride == big && c->max_nseq == max_nseq;
    if (!same) {
        const bool tm = getenv("VC_TIME_SUBMIT") != nullptr;          // development: where a submit that re-creates the workspaces spends its time
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_0 = now();
        drain(c);                           // the other batch may be running on the workspaces that go
        const double t_1 = now();
        free_workspaces(c);
        const double t_2 = now();
        c->n_streams = S;
        c->ws_packed = bt->packed; c->wcols = wcols; c->MA = MA; c->NC = NC; c->EC = EC; c->CW = CW; c->ws_cpl = cpl; c->PC = PC; c->group_max = group_max; c->max_nseq = max_nseq;
        c->ws_max_len = ws_max_len;
        c->big_ws_stride = big;
        // re-alignment rounds work on pruned graphs (a quarter of NC rows, typically), so more alignments per window fit the
        // same matrix space than full-height ones: the small per-job arrays are sized for up to 16
        c->rgroup_max = wcols ? group_max : std::min(std::max(group_max, 16u), std::max(max_nseq, 1u));
        c->jobs_cap = CW * c->rgroup_max;
        c->hmat_dwords = (uint64_t)CW * group_max * NC * rowd;
        size_t f0 = 0, f1 = 0, tt = 0;
        (void)hipMemGetInfo(&f0, &tt);
        for (uint32_t s = 0; s < S; ++s) {
            c->arena_cur = c->arena.empty() ? 0u : s % (uint32_t)c->arena.size();
            if ((rc = alloc_work(c, &c->works[s]))) return rc;
        }
        (void)hipMemGetInfo(&f1, &tt);
        c->chunk_bytes = f0 > f1 ? f0 - f1 : 0;
        c->have_ws = true;
        if (tm) fprintf(stderr, "vc_submit: workspaces re-created for %u windows: drain %.3f s, free %.3f s, allocate %.3f s (%.1f GiB on %u streams, CW %u)\n",
                        nw, t_1 - t_0, t_2 - t_1, now() - t_2, c->chunk_bytes / 1073741824.0, S, CW);
    }
    b.cons_cap = NC;
    const bool wide_cls = bt->cpl >= 32;                 // (launch_fwd_t instantiates the forward kernel with the same numbers)
    bt->kept = (kKept && NC < 32768 && !getenv("VC_PLAIN_RING")) ? (uint32_t)(wide_cls ? kKeptWide : kKept) : 0u;
    bt->ring = (uint32_t)(wide_cls ? kRingWide : kRing); bt->ring_pruned = (uint32_t)(wide_cls ? kRingPrunedWide : kRingPruned);
    // (round 6, VC_BAND_RAW=1: raw int16 rows banded too -- the widest classes and scores outside the byte form; measured slower, off)
    bt->band = (bt->packed || c->band_raw) && bt->kept && c->trace_wave && !getenv("VC_NO_BAND");
    if ((rc = salloc(c, bt, 12, &b.cons, (size_t)nw * b.cons_cap))) return rc;
    HIPCHK(c, hipMemsetAsync(b.status, 0, nw, c->stream));
    HIPCHK(c, hipMemsetAsync(b.cons_len, 0, nw * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // chunks of this batch: as large as the workspace allows, and equal, so that the last round of chunks is not a lone one
    {
        const uint64_t SB = want_streams;
        const uint64_t per_round = SB * c->CW;
        const uint64_t rounds = (nw + per_round - 1) / per_round;
        uint64_t cw = (nw + SB * rounds - 1) / (SB * rounds);
        cw = std::min<uint64_t>((cw + 63) & ~63ull, c->CW);
        bt->cw_run = (uint32_t)std::max<uint64_t>(cw, 1);
    }
    bt->n_streams = want_streams;
    c->stats.max_nodes = NC; c->stats.max_edges = EC; c->stats.chunk_windows = bt->cw_run;
    bt->have = true;
    c->cur = bt;
    return VC_OK;
}

int vc_run(vc_ctx* c) {
    if (!c) return VC_ERR_ARG;
    Batch* bt = c->cur;
    if (!bt || !bt->have) return fail(c, VC_ERR_STATE, "vc_run before vc_submit (or after vc_release / vc_reserve gave the staged batch's workspaces back)");
    if (!c->have_ws) return fail(c, VC_ERR_STATE, "vc_run: the workspaces are gone; submit the batch again");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = wait_batch(c, bt))) return rc;             // the same batch again: its last run must have left its buffers
    const VcBatchDev& b = bt->b;
    delete bt->pl;
    Plan* const plp = bt->pl = make_plan(c, bt);
    Plan& pl = *plp;
    if ((rc = lds_limit(c, (const void*)k_topo, std::min(pl.topo_lds, kLdsCap))) || (rc = lds_limit(c, (const void*)k_prune_lcc, std::min(pl.prune_lds, kLdsCap)))) return rc;
    if (pl.add_lds > kLdsCap) return fail(c, VC_ERR_ARG, "a layer of %u bases on graphs of %u nodes needs %u bytes of LDS in k_addaln (limit %u)", c->ws_max_len, c->NC, pl.add_lds, kLdsCap);
    if ((rc = lds_limit(c, (const void*)k_addaln, pl.add_lds))) return rc;
    if (bt->kept) {
        const uint32_t sub_lds = std::max(8 * ((c->NC + 63) / 64) + 2 * c->NC + ((c->NC + 15) & ~15u) + 4 * (c->NC / 32 + 1) + 64, vc_kept_lds_bytes(c->NC));
        if ((rc = lds_limit(c, (const void*)k_init, vc_kept_lds_bytes(c->NC))) || (rc = lds_limit(c, (const void*)k_rows_sub, sub_lds))) return rc;
    }
    if (c->prm.mode == 1 && (rc = lds_limit(c, (const void*)k_consensus, std::min(pl.cons_lds, kLdsCap)))) return rc;
    bool alone;                                          // no other run of this context is in flight: the counters are this run's
    { std::lock_guard<std::mutex> lk(c->qmu); alone = c->runq.empty(); }
    if (alone) {
        HIPCHK(c, hipMemsetAsync(c->d_stat, 0, 8 * VC_STAT_WORDS, c->stream));
#ifdef VC_EXPERIMENTS
        HIPCHK(c, hipMemsetAsync(c->d_pipe_abort, 0, 64, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pipe_prof, 0, VC_PP_TOTAL * 8, c->stream));
#endif
    }
    HIPCHK(c, hipMemsetAsync(b.status, 0, b.n_windows, c->stream));
    if (!bt->h_pre_status.empty()) HIPCHK(c, hipMemcpyAsync(b.status, bt->h_pre_status.data(), b.n_windows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b.errinfo, 0, (size_t)b.n_windows * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(b.cons_len, 0, (size_t)b.n_windows * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (alone) {
        std::lock_guard<std::mutex> lk(c->mu);
        for (int i = 0; i < KC_N; ++i) { c->stats.ms[i] = 0; c->stats.busy_ms[i] = 0; c->stats.launches[i] = 0; }
    }
    c->stats.chunk_windows = bt->cw_run; c->stats.n_streams = bt->n_streams;

    const uint32_t S = bt->n_streams, CW = bt->cw_run;
    bt->collected = false; bt->run_rc = VC_OK; bt->run_seq = ++c->run_counter;
    for (uint32_t s = 0; s < kMaxStreams; ++s) bt->ev_rec[s] = false;
    if (c->host_threads && c->dbg_stop_kind == 0) {
        try {
            while (c->workers_made < S) { c->workers[c->workers_made] = std::thread(stream_worker, c, c->workers_made); c->workers_made++; }
        } catch (const std::exception& e) {                  // (no exception may cross the C boundary)
            return fail(c, VC_ERR_STATE, "cannot start a chunk thread: %s", e.what());
        }
        {
            std::lock_guard<std::mutex> lk(c->qmu);
            bt->next_chunk = 0; bt->chunks_done = 0; bt->n_chunks = (b.n_windows + CW - 1) / CW;
            bt->queued = true;
            c->runq.push_back(bt);
        }
        c->qcv.notify_all();
        bt->ran = true;                     // vc_sync / vc_collect wait for the workers and report what they met
        return VC_OK;
    }
    // development / test hooks (vc_debug_stop_after, VC_HOST_THREADS=0): this thread walks the streams in lock-step, nothing else in flight
    drain(c);
    bt->first_stream = 0;
    for (uint32_t g0 = 0; g0 < b.n_windows; g0 += S * CW) {
        // S chunks advance in lockstep, each on its own stream
        uint32_t max_layers = 0;
        for (uint32_t s = 0; s < S; ++s) {
            const uint32_t w0 = g0 + s * CW;
            c->works[s].active = false;
            if (w0 >= b.n_windows) continue;
            pl.begin(c->works[s], w0, std::min(CW, b.n_windows - w0));
            max_layers = std::max(max_layers, c->works[s].layers);
        }
        for (uint32_t j = 1; j <= max_layers; ++j) {
            for (uint32_t s = 0; s < S; ++s)
                if (c->works[s].active && j <= c->works[s].layers && (rc = pl.build_layer(c->works[s], j, true))) return rc;
            if (c->dbg_stop_kind == 1 && c->dbg_stop_index == j) { bt->ran = false; return VC_OK; }
        }
        if (c->prm.mode == 1) {
            for (uint32_t s = 0; s < S; ++s) {
                if (!c->works[s].active) continue;
                if (c->works[s].layers) { if ((rc = pl.linear_tail(c->works[s]))) return rc; }
                else c->works[s].active = false;
            }
            continue;
        }
        for (uint32_t r = 0; r < c->prm.num_prune; ++r) {
            const bool more = r + 1 < c->prm.num_prune;
            for (uint32_t s = 0; s < S; ++s)
                if (c->works[s].active && c->works[s].layers && (rc = pl.prune(c->works[s], more))) return rc;
            if (c->dbg_stop_kind == 2 && c->dbg_stop_index == r) { bt->ran = false; return VC_OK; }
            if (!more) break;
            for (uint32_t s = 0; s < S; ++s)
                if (c->works[s].active && c->works[s].layers && (rc = pl.realign(c->works[s]))) return rc;
            if (c->dbg_stop_kind == 3 && c->dbg_stop_index == r) { bt->ran = false; return VC_OK; }
        }
        for (uint32_t s = 0; s < S; ++s) {
            if (!c->works[s].active) continue;
            if (c->works[s].layers) { if ((rc = pl.finish(c->works[s]))) return rc; }
            else c->works[s].active = false;
        }
    }
    for (uint32_t s = 0; s < S; ++s) { HIPCHK(c, hipEventRecord(bt->done_ev[s], c->streams[s])); bt->ev_rec[s] = true; }
    HIPCHK(c, hipGetLastError());
    bt->ran = true;
    return VC_OK;
}

namespace {
// the run of a batch is over: what it met
int finish_run(vc_ctx* c, Batch* bt) {
    int rc = wait_batch(c, bt);
    if (rc) { bt->ran = false; return rc; }
    if (bt->run_rc != VC_OK) { bt->ran = false; return bt->run_rc; }       // the run failed: its own error text stands (vc_last_error)
    if (c->pipe && c->d_pipe_abort) {
        uint32_t ab = 0;
        HIPCHK(c, hipMemcpyAsync(&ab, c->d_pipe_abort, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (ab) { bt->ran = false; bt->run_rc = VC_ERR_HIP; return fail(c, VC_ERR_HIP, "persistent build pipeline gave up waiting (site %u): the run has no result", ab); }
    }
    return VC_OK;
}
// the batch vc_collect / vc_result_size speak of: the oldest run whose results nobody has taken; none such: the latest run
Batch* collect_target(vc_ctx* c) {
    Batch* best = nullptr;
    for (Batch& bt : c->bt) if (bt.have && bt.ran && !bt.collected && (!best || bt.run_seq < best->run_seq)) best = &bt;
    if (best) return best;
    for (Batch& bt : c->bt) if (bt.have && bt.ran && (!best || bt.run_seq > best->run_seq)) best = &bt;
    return best;
}
}  // namespace

int vc_sync(vc_ctx* c) {
    if (!c) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = VC_OK;
    for (int pass = 0; pass < 2; ++pass)                     // in the order they were run
        for (Batch& bt : c->bt) {
            if (!bt.have || !bt.ran) continue;
            const bool older = !c->cur || &bt != c->cur;
            if ((pass == 0) != older) continue;
            const int r = finish_run(c, &bt);
            if (r && !rc) rc = r;
        }
    if (c->prm.profile) { std::lock_guard<std::mutex> lk(c->mu); flush_events(c); }
    if (rc) return rc;
    HIPCHK(c, hipGetLastError());
    return VC_OK;
}

static int fetch_lengths(vc_ctx* c, Batch** out) {
    Batch* bt = collect_target(c);
    if (!bt) {
        for (Batch& x : c->bt) if (x.have && x.run_rc != VC_OK) return x.run_rc;      // the run failed: its own error text stands
        return fail(c, VC_ERR_STATE, "no finished run");
    }
    int rc = finish_run(c, bt);
    if (rc) return rc;
    const uint32_t nw = bt->b.n_windows;
    bt->h_cons_len.resize(nw); bt->h_status.resize(nw);
    HIPCHK(c, hipMemcpyAsync(bt->h_cons_len.data(), bt->b.cons_len, (size_t)nw * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(bt->h_status.data(), bt->b.status, nw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t tot = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        if (bt->h_status[w] > VC_WIN_UNPOLISHED) bt->h_cons_len[w] = 0;
        tot += bt->h_cons_len[w];
    }
    bt->total_cons = tot;
    *out = bt;
    return VC_OK;
}

int vc_result_size(vc_ctx* c, uint64_t* cons_bytes) {
    if (!c || !cons_bytes) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Batch* bt = nullptr;
    int rc = fetch_lengths(c, &bt);
    if (rc) return rc;
    *cons_bytes = bt->total_cons;
    return VC_OK;
}

int vc_result_windows(vc_ctx* c, uint32_t* n_windows) {
    if (!c || !n_windows) return VC_ERR_ARG;
    Batch* bt = collect_target(c);
    if (!bt) return fail(c, VC_ERR_STATE, "no run to collect");
    *n_windows = bt->b.n_windows;
    return VC_OK;
}

namespace {
int collect_device(vc_ctx* c, Batch* bt, void* d_cons, uint64_t cons_cap, void* d_cons_off, void* d_status) {
    if (bt->total_cons > cons_cap) return fail(c, VC_ERR_CAPACITY, "consensus needs %llu bytes, buffer has %llu",
                                               (unsigned long long)bt->total_cons, (unsigned long long)cons_cap);
    const uint32_t nw = bt->b.n_windows;
    std::vector<uint64_t> off(nw + 1, 0);
    for (uint32_t w = 0; w < nw; ++w) off[w + 1] = off[w] + bt->h_cons_len[w];
    HIPCHK(c, hipMemcpyAsync(d_cons_off, off.data(), (size_t)(nw + 1) * 8, hipMemcpyHostToDevice, c->stream));
    // statuses > UNPOLISHED publish no bytes: zero their lengths on the device view used by the gather
    HIPCHK(c, hipMemcpyAsync(bt->b.cons_len, bt->h_cons_len.data(), (size_t)nw * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_cons, dim3(nw), dim3(64), 0, c->stream, bt->b, (const uint64_t*)d_cons_off, (uint8_t*)d_cons, cons_cap);
    if (d_status) HIPCHK(c, hipMemcpyAsync(d_status, bt->b.status, nw, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VC_OK;
}
}  // namespace

int vc_collect_device(vc_ctx* c, void* d_cons, uint64_t cons_cap, void* d_cons_off, void* d_status) {
    if (!c || !d_cons || !d_cons_off) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Batch* bt = nullptr;
    int rc = fetch_lengths(c, &bt);
    if (rc) return rc;
    if ((rc = collect_device(c, bt, d_cons, cons_cap, d_cons_off, d_status))) return rc;
    bt->collected = true;
    return VC_OK;
}

int vc_collect(vc_ctx* c, vc_result* r) {
    if (!c || !r || !r->cons_off || !r->status) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Batch* bt = nullptr;
    int rc = fetch_lengths(c, &bt);
    if (rc) return rc;
    if (bt->total_cons > r->cons_cap || (bt->total_cons && !r->cons))
        return fail(c, VC_ERR_CAPACITY, "consensus needs %llu bytes, buffer has %llu",
                    (unsigned long long)bt->total_cons, (unsigned long long)r->cons_cap);
    const uint32_t nw = bt->b.n_windows;
    uint8_t* d_out = nullptr; uint64_t* d_off = nullptr;
    if ((rc = salloc(c, bt, 13, &d_out, bt->total_cons + 16)) || (rc = salloc(c, bt, 14, &d_off, (size_t)nw + 1))) return rc;
    rc = collect_device(c, bt, d_out, bt->total_cons + 16, d_off, nullptr);
    if (rc == VC_OK && bt->total_cons) {
        HIPCHK(c, hipMemcpyAsync(r->cons, d_out, bt->total_cons, hipMemcpyDeviceToHost, c->stream));      // (on the context's own stream: the next batch's chunks keep running)
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (rc) return rc;
    r->cons_off[0] = 0;
    for (uint32_t w = 0; w < nw; ++w) {
        r->cons_off[w + 1] = r->cons_off[w] + bt->h_cons_len[w];
        r->status[w] = bt->h_status[w];
    }
    bt->collected = true;
    return VC_OK;
}

int vc_debug_errinfo(vc_ctx* c, uint32_t* out) {
    if (!c || !out || !(c->cur && c->cur->b.errinfo)) return VC_ERR_ARG;      // (the last batch submitted: its diagnostics outlive vc_release)
    HIPCHK(c, hipSetDevice(c->device));
    sync_all(c);
    HIPCHK(c, hipMemcpy(out, c->cur->b.errinfo, (size_t)c->cur->b.n_windows * 4, hipMemcpyDeviceToHost));
    return VC_OK;
}

// Test hook (tests/test_gpu.py::test_stage_digests_on_the_device): make vc_run return after a given stage -- kind 1: build
// layer `index` added; 2: prune + LargestSubgraph number `index` done; 3: AddWeights round `index` done; 0: run to the end --
// so that vc_debug_stage_digest can look at the graphs where they stand.  A run stopped this way has no result to collect.
int vc_debug_stop_after(vc_ctx* c, uint32_t kind, uint32_t index) {
    if (!c || kind > 3) return VC_ERR_ARG;
    c->dbg_stop_kind = kind; c->dbg_stop_index = index;
    return VC_OK;
}

// Digest of window w's current graph and of the alignment that was walked last, in the record format of
// oracle/ref_harness.cpp:vcref_window_stages (nodes, edges, hash(nodes: byte, aligned ids), hash(edges: tail, head, weight),
// pairs, hash(pairs: node id or -1, sequence position or -1)); out[0..1] are left to the caller.  Single-chunk batches only.
int vc_debug_stage_digest(vc_ctx* c, uint32_t w, int with_pairs, uint64_t* out) {
    if (!c || !out || !(c->cur && c->cur->have) || w >= c->cur->b.n_windows) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->cur->b.n_windows > c->cur->cw_run) return fail(c, VC_ERR_ARG, "vc_debug_stage_digest: the batch spans several chunks");
    sync_all(c);
    const Work& wk = c->works[c->cur->first_stream];
    const VcGraph& g = wk.gr[wk.cur];
    const uint32_t slot = w, NC = c->NC, EC = c->EC, MA = c->MA;
    uint32_t N = 0, E = 0, P = 0;
    HIPCHK(c, hipMemcpy(&N, g.n_nodes + slot, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&E, g.n_edges + slot, 4, hipMemcpyDeviceToHost));
    if (N > NC || E > EC) return fail(c, VC_ERR_STATE, "vc_debug_stage_digest: graph sizes out of range");
    std::vector<uint8_t> code(NC), alc(NC);
    std::vector<uint16_t> al((size_t)NC * MA), r2n(NC);
    std::vector<uint32_t> etn(EC), ehn(EC), ew(EC);
    HIPCHK(c, hipMemcpy(code.data(), g.code + (size_t)slot * NC, NC, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(alc.data(), g.al_cnt + (size_t)slot * NC, NC, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(al.data(), g.al + (size_t)slot * NC * MA, (size_t)NC * MA * 2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(etn.data(), g.e_tn + (size_t)slot * EC, (size_t)EC * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ehn.data(), g.e_hn + (size_t)slot * EC, (size_t)EC * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ew.data(), g.e_w + (size_t)slot * EC, (size_t)EC * 4, hipMemcpyDeviceToHost));
    struct Fnv {
        uint64_t h = 1469598103934665603ull;
        void u8(uint8_t b) { h ^= b; h *= 1099511628211ull; }
        void u32(uint32_t v) { for (int i = 0; i < 4; ++i) u8((uint8_t)(v >> (8 * i))); }
        void u64(uint64_t v) { for (int i = 0; i < 8; ++i) u8((uint8_t)(v >> (8 * i))); }
    } hn, he, hp;
    for (uint32_t v = 0; v < N; ++v) {
        hn.u8(code[v]); hn.u32(alc[v]);
        for (uint32_t k = 0; k < alc[v]; ++k) hn.u32(al[(size_t)v * MA + k]);
    }
    for (uint32_t e = 0; e < E; ++e) { he.u32(etn[e] & 0xFFFF); he.u32(ehn[e] & 0xFFFF); he.u64(ew[e]); }
    out[2] = N; out[3] = E; out[4] = hn.h; out[5] = he.h; out[6] = 0; out[7] = 0;
    if (with_pairs) {
        HIPCHK(c, hipMemcpy(&P, wk.d_npairs + slot, 4, hipMemcpyDeviceToHost));
        if (P > c->PC) return fail(c, VC_ERR_STATE, "vc_debug_stage_digest: pair count out of range");
        std::vector<uint32_t> pr(P);
        if (P) HIPCHK(c, hipMemcpy(pr.data(), wk.d_pairs + (size_t)slot * c->PC, (size_t)P * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(r2n.data(), wk.dp.rank2node + (size_t)slot * NC, (size_t)NC * 2, hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < P; ++f) {                    // stored tail first as (row << 16) | column, 0 = none
            const uint32_t pv = pr[P - 1 - f], row = pv >> 16, col = pv & 0xFFFF;
            hp.u32(row ? (uint32_t)r2n[row - 1] : 0xFFFFFFFFu);
            hp.u32(col ? col - 1 : 0xFFFFFFFFu);
        }
        out[6] = P; out[7] = hp.h;
    }
    return VC_OK;
}

// development (tools/gpu_rowstats.py): the row records (backtrack view, 4 dwords per row) that the next alignment of window w will
// use, after a run stopped with vc_debug_stop_after; returns the number of rows through *nrows.  Single-chunk batches only.
int vc_debug_rows(vc_ctx* c, uint32_t w, uint32_t* out, uint32_t cap_rows, uint32_t* nrows) {
    if (!c || !out || !nrows || !(c->cur && c->cur->have) || w >= c->cur->b.n_windows || c->cur->b.n_windows > c->cur->cw_run) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    sync_all(c);
    const Work& wk = c->works[c->cur->first_stream];
    uint32_t n = 0;
    HIPCHK(c, hipMemcpy(&n, wk.dp.nrows + w, 4, hipMemcpyDeviceToHost));
    if (n > c->NC || n > cap_rows) return fail(c, VC_ERR_CAPACITY, "vc_debug_rows: %u rows", n);
    HIPCHK(c, hipMemcpy(out, wk.dp.rec + (size_t)w * c->NC, (size_t)n * 16, hipMemcpyDeviceToHost));
    *nrows = n;
    return VC_OK;
}

#ifdef VC_LAB
// development (tools/gpu_fwd_lab.py): build the first chunk up to `layer`, prepare that layer's rows, then time `reps`
// launches of k_fwd alone with parts of its row loop switched off (VcFwdArgs::dbg).  Nothing downstream runs.
int vc_debug_fwd_lab(vc_ctx* c, uint32_t layer, uint32_t reps, uint32_t flags, float* ms_out, unsigned long long* cells_out) {
    if (!c || !(c->cur && c->cur->have)) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Batch* const bt = c->cur;
    delete bt->pl;
    bt->pl = make_plan(c, bt);
    Plan& pl = *bt->pl;
    { int rc_ = lds_limit(c, (const void*)k_addaln, pl.add_lds); if (rc_) return rc_; }
    HIPCHK(c, hipMemsetAsync(c->cur->b.status, 0, c->cur->b.n_windows, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    Work& wk = c->works[0];
    static uint32_t built_to = 0;
    int rc;
    if (built_to != layer) {
        pl.begin(wk, 0, std::min(c->cur->cw_run, c->cur->b.n_windows));
        for (uint32_t j = 1; j < layer; ++j) if ((rc = pl.build_layer(wk, j, true))) return rc;
        built_to = layer;
    }
    VcFwdArgs fa = pl.fwd_args(wk);
    fa.group = 1; fa.k0 = layer; fa.mode = 0; fa.hstride = (uint64_t)pl.NC * pl.rowd; fa.dbg = flags;
    HIPCHK(c, hipMemsetAsync(c->d_stat, 0, 8 * VC_STAT_WORDS, wk.stream));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipMemsetAsync(wk.d_tie_n, 0, 4, wk.stream));
    if ((rc = launch_fwd(c, bt, wk.stream, fa, wk.ns))) return rc;            // warm-up
    HIPCHK(c, hipEventRecord(e0, wk.stream));
    for (uint32_t r = 0; r < reps; ++r) {
        HIPCHK(c, hipMemsetAsync(wk.d_tie_n, 0, 4, wk.stream));
        if ((rc = launch_fwd(c, bt, wk.stream, fa, wk.ns))) return rc;
    }
    HIPCHK(c, hipEventRecord(e1, wk.stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / reps;
    unsigned long long raw[VC_STAT_WORDS], cells = 0, rows = 0;
    HIPCHK(c, hipMemcpy(raw, c->d_stat, sizeof(raw), hipMemcpyDeviceToHost));
    for (int i = 0; i < VC_STAT_SLOTS; ++i) { cells += raw[i * 8]; rows += raw[i * 8 + 1]; }
    cells_out[0] = cells / (reps + 1); cells_out[1] = rows / (reps + 1);

    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return VC_OK;
}
#endif

// development: the persistent pipeline's counters and the per-window words its waves hand to each other, for the first `n` windows
// of chunk stream 0: out[0..VC_PC_N) counters, then per window {layer, job_end, job_type, npairs}
int vc_debug_pipe_state(vc_ctx* c, uint32_t* out, uint32_t n) {
    if (!c || !out || !(c->cur && c->cur->have)) return VC_ERR_ARG;
#ifndef VC_EXPERIMENTS
    (void)n;
    return fail(c, VC_ERR_ARG, "built without -DVC_EXPERIMENTS: no persistent pipeline");
#else
    HIPCHK(c, hipSetDevice(c->device));
    sync_all(c);
    const Work& wk = c->works[c->cur->first_stream];
    std::vector<uint32_t> ctl((size_t)VC_PC_N * VC_PIPE_CTL_STRIDE);
    HIPCHK(c, hipMemcpy(ctl.data(), wk.d_pipe_ctl, ctl.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 10; ++i) out[i] = i < VC_PC_N ? ctl[(size_t)i * VC_PIPE_CTL_STRIDE] : 0u;
    HIPCHK(c, hipMemcpy(&out[9], c->d_pipe_abort, 4, hipMemcpyDeviceToHost));
    n = std::min(n, c->cur->cw_run);
    std::vector<uint32_t> a(n), b(n), d(n); std::vector<uint8_t> t(n);
    HIPCHK(c, hipMemcpy(a.data(), wk.d_cur_layer, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(b.data(), wk.d_job_end, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(t.data(), wk.d_job_type, n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(d.data(), wk.d_npairs, n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        out[10 + 4 * i] = a[i]; out[10 + 4 * i + 1] = b[i]; out[10 + 4 * i + 2] = t[i]; out[10 + 4 * i + 3] = d[i];
        if (d[i] && d[i] <= c->PC) {                       // pairs with a sequence position, in place of job_type
            std::vector<uint32_t> pr(d[i]);
            HIPCHK(c, hipMemcpy(pr.data(), wk.d_pairs + (size_t)i * c->PC, (size_t)d[i] * 4, hipMemcpyDeviceToHost));
            uint32_t nv = 0;
            for (uint32_t x : pr) nv += (x & 0xFFFF) != 0;
            out[10 + 4 * i + 2] = nv;
        }
    }
    return VC_OK;
#endif
}

// development: phase clocks of the persistent pipeline's waves over the last run (VC_PP_* order, ticks of 100 MHz / counts)
int vc_debug_pipe_prof(vc_ctx* c, unsigned long long* out) {
    if (!c || !out) return VC_ERR_ARG;
#ifndef VC_EXPERIMENTS
    return fail(c, VC_ERR_ARG, "built without -DVC_EXPERIMENTS: no persistent pipeline");
#else
    HIPCHK(c, hipSetDevice(c->device));
    sync_all(c);
    HIPCHK(c, hipMemcpy(out, c->d_pipe_prof, VC_PP_TOTAL * 8, hipMemcpyDeviceToHost));
    return VC_OK;
#endif
}

int vc_get_stats(vc_ctx* c, vc_stats* s) {
    if (!c || !s) return VC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    drain(c);                            // launch counters and event records belong to the chunk threads until they are done
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, raw[VC_STAT_WORDS];
    HIPCHK(c, hipMemcpy(raw, c->d_stat, sizeof(raw), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8 * VC_STAT_SLOTS; ++i) st[i % 8] += raw[i];
    c->stats.cells = st[0]; c->stats.dp_rows = st[1];
    c->stats.far_row_reads = st[3];
    c->stats.trace_steps = st[4]; c->stats.trace_spec = st[5]; c->stats.trace_rounds = st[6];
    c->stats.band_redo = st[7];
    c->stats.fwd_shader_cycles = 0; c->stats.fwd_wall_ticks = 0;
    for (int i = 0; i < VC_STAT_SLOTS; ++i) { c->stats.fwd_shader_cycles += raw[8 * VC_STAT_SLOTS + 2 * i]; c->stats.fwd_wall_ticks += raw[8 * VC_STAT_SLOTS + 2 * i + 1]; }
    {
        uint64_t held = c->chunk_bytes + c->arena_bytes;
        for (Batch& bt : c->bt) for (auto& sl : bt.slots) held += sl.cap;
        c->stats.device_bytes = held;
    }
#ifdef VC_ADD_PROF
    { unsigned long long h[8] = {0}; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vc_add_prof), sizeof(h));
      fprintf(stderr, "[add prof] waves %llu  ticks per wave: A %llu  B %llu  C %llu  D %llu  rows %llu\n", h[7], h[7] ? h[0] / h[7] : 0, h[7] ? h[1] / h[7] : 0, h[7] ? h[2] / h[7] : 0, h[7] ? h[3] / h[7] : 0, h[7] ? h[4] / h[7] : 0); }
#endif
    c->stats.alignments = c->stats.launches[KC_FWD];
    *s = c->stats;
    return VC_OK;
}

}  // extern "C"
