// The persistent build pipeline (gfx950): the build loop of Window::generate_consensus (src/window.cpp:239-298) for a whole chunk
// of windows WITHOUT a kernel boundary per layer and without lock-step between the windows.
//
// The lock-step plan (vc_api.hip: Plan::build_layer) launches k_fwd -> k_resolve -> k_tracew -> redo k_fwd -> redo k_tracew ->
// k_addaln once per layer for every window of a chunk: ~6 launches x 63 layers x 16 chunks per step, each launch a barrier over
// the chunk.  Here two resident kernels with different register / LDS footprints work off device-side queues instead:
//
//   k_pipe_fwd      "forward" waves (one per alignment, the k_fwd footprint).  An item is a window whose backtrack is done: the
//                   wave adds the alignment to the graph (AddAlignment, graph.cpp:182-299), makes the row records of the next
//                   layer and runs that layer's forward DP (sisd_alignment_engine.cpp:118-360) -- the graph arrays it has just
//                   written are still in its cache -- then hands the window on.
//   k_pipe_trace    backtrack waves, VC_TG alignments of DIFFERENT windows and layers per wave (16 lanes each, the k_tracew
//                   footprint): sisd_alignment_engine.cpp:362-459.  A walk that leaves the stored band sends its window back to
//                   the forward queue as a "redo" item (whole rows), everything else as an "add" item.
//                   An alignment that ended in a tie between sinks (sisd :353-355, ~1.5 % of them) is settled by the same wave
//                   first (vc_resolve_one) -- a third kernel for it would need a guaranteed slot beside the other two.
//
// A window is owned by exactly one wave at any time and travels  fq -> tq -> fq ...  until its last layer is added;
// which wave takes it next is decided by the queues, so the mix of forward and backtrack work on a CU follows the load by
// itself.  Hand-over protocol (cdna_hip_programming.md section 6, Guideline 16): the producer wave finishes its stores,
// agent-scope release, `s_waitcnt vmcnt(0)`, then ONE lane publishes an 8-byte {lap tag, value} granule with a relaxed
// agent-scope store; the consumer takes a ticket, polls that one granule relaxed (s_sleep between polls), then ONE agent-scope
// acquire, then plain loads.  Every spin is bounded: a wave that waits longer than VcPipe::spin_limit polls sets the abort word,
// every wave leaves, and the host reports the run as failed -- a protocol error can cost a run, never hang the device.
#pragma once
#include "vc_kernels.h"

#define VC_Q_NONE 0xFFFFFFFFu
#define VC_Q_SLOT_MASK 0x00FFFFFFu
// item kinds (bits 24..27)
#define VC_QK_START 0u      // fq: first layer (rows made by k_init)
#define VC_QK_ADD   1u      // fq: backtrack done: AddAlignment, next layer's rows, next layer's forward pass
#define VC_QK_REDO  2u      // fq: the backtrack left the band: forward pass again, whole rows; tq: walk the whole rows
#define VC_QK_TIE   3u      // tq: the end cell is tied between sinks: settle it (vc_resolve_one), then walk

// one word per 128-byte line: the counters are hammered by different CUs
#define VC_PIPE_CTL_STRIDE 32u
enum { VC_PC_FQ_RES = 0, VC_PC_FQ_HEAD, VC_PC_TQ_RES, VC_PC_TQ_HEAD, VC_PC_ACTIVE, VC_PC_DONE, VC_PC_FINISHED, VC_PC_N };
// phase clocks of the waves (ticks of the 100 MHz wall clock, summed over waves): where a resident wave's time goes
enum { VC_PP_F_WAIT = 0, VC_PP_F_ADD, VC_PP_F_FWD, VC_PP_F_HAND, VC_PP_F_ITEMS, VC_PP_F_WAVES, VC_PP_T_WAIT, VC_PP_T_WALK, VC_PP_T_HAND, VC_PP_T_ROUNDS,
       VC_PP_T_ITEMS, VC_PP_T_WAVES, VC_PP_T_TIES, VC_PP_F_PICKUP, VC_PP_T_PICKUP, VC_PP_T_TIETIME, VC_PP_N = 16,
       VC_PP_TL = 32, VC_PP_TL_TICKS = 2000000,     // timeline: 32 buckets of 20 ms since the wave started: F wait, F busy, T rounds, T items
       VC_PP_TOTAL = VC_PP_N + 4 * VC_PP_TL };

struct VcQueue {
    unsigned long long* slots;     // [mask + 1] granules: (lap + 1) << 32 | value
    uint32_t* res;                 // producers' reservation counter
    uint32_t* head;                // consumers' ticket counter
    uint32_t mask, shift;          // capacity - 1 (a power of two >= windows of the chunk: a window has one item in flight), log2(capacity)
};

struct VcPipe {
    VcQueue fq, tq;
    uint32_t* n_active;            // windows that entered the pipeline (k_pipe_seed)
    uint32_t* done;                // windows retired
    uint32_t* finished;            // done == n_active: every waiting wave leaves
    uint32_t* abort_code;          // != 0: a spin ran out (site code); NOT cleared between the chunks of a run
    uint32_t* cur_layer;           // [CW] layer the window is at
    unsigned long long* prof;      // [VC_PP_N] phase clocks (nullptr: not kept)
    unsigned long long* pub_time;  // [CW] development: wall clock at which the window's current item was published
    uint32_t spin_limit;           // patience of a waiting wave, in ticks of the constant 100 MHz clock (wall_clock64)
};

// ---- queue primitives.  Everything here is WAVE-UNIFORM: all 64 lanes are active on entry and on exit, results are scalars,
// and the single-lane memory operations run under an exec mask set inside the instruction sequence itself (as k_fwd's band store
// does) -- no `if (lane == 0)` region and no loop with a lane-dependent exit for the compiler to restructure.  (The first
// versions of this file, written with __hip_atomic_* under `if (lane == 0)`, lost lanes from the exec mask behind the multi-exit
// spin loops: the library is built with -structurizecfg-skip-uniform-regions.)  Control words are read through the VECTOR
// memory path at agent scope (sc1: served by the L2, never by this CU's L1); a uniform plain load could be selected as a scalar
// load, and the scalar cache would answer every poll with the value it saw first.
__device__ __forceinline__ uint32_t vq_ldu(const uint32_t* p) {
    uint32_t v, r;
    asm volatile("global_load_dword %0, %2, off sc1\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0" : "=&v"(v), "=s"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void vq_stu(uint32_t* p, uint32_t x) {                 // lane 0 stores
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_dword %0, %1, off sc1\n\ts_mov_b64 exec, -1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void vq_stu64(unsigned long long* p, uint32_t lo, uint32_t hi) {
    const unsigned long long g = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_dwordx2 %0, %1, off sc1\n\ts_mov_b64 exec, -1" :: "v"(p), "v"(g) : "memory");
}
// old value of *p, *p += k (agent scope, as the compiler issues __hip_atomic_fetch_add(relaxed, agent) on gfx950)
__device__ __forceinline__ uint32_t vq_add(uint32_t* p, uint32_t k) {
    uint32_t v = k, r;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %0, off sc0\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_mov_b64 exec, -1"
                 : "+v"(v), "=s"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t vq_inc(uint32_t* p) {
    uint32_t v = 1u, r;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %0, off sc0\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_mov_b64 exec, -1"
                 : "+v"(v), "=s"(r) : "v"(p) : "memory");
    return r;
}
// the current value of a word that is only ever changed by atomics (queue counters): read with an atomic too (fetch-add 0) --
// a plain agent-scope load is served by this XCD's L2, which may hold the word as it was before another XCD's atomics
// (measured: the backtrack waves saw "no second item" three times out of four while thousands were queued)
__device__ __forceinline__ uint32_t vq_ldc(uint32_t* p) {
    uint32_t v = 0u, r;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %0, off sc0\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_mov_b64 exec, -1"
                 : "+v"(v), "=s"(r) : "v"(p) : "memory");
    return r;
}
// compare-and-swap: old value of *p; *p = desired when it was `expected`
__device__ __forceinline__ uint32_t vq_cas(uint32_t* p, uint32_t expected, uint32_t desired) {
    unsigned long long dc = ((unsigned long long)expected << 32) | desired;       // VDATA[0] = new value, VDATA[1] = compare value
    uint32_t v, r;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_cmpswap %0, %2, %3, off sc0\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_mov_b64 exec, -1"
                 : "=&v"(v), "=s"(r) : "v"(p), "v"(dc) : "memory");
    return r;
}
// polls the TAG (high dword) of a granule until it equals `want`, at most `n` times; returns the last tag seen
__device__ __forceinline__ uint32_t vq_poll_tag(const unsigned long long* g, uint32_t want, uint32_t n, uint32_t sleep_long) {
    uint32_t v, tag, cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    want = (uint32_t)__builtin_amdgcn_readfirstlane((int)want);            // (an "s" operand is taken as it stands: make sure these ARE scalars)
    sleep_long = (uint32_t)__builtin_amdgcn_readfirstlane((int)sleep_long);
    asm volatile("1:\n\tglobal_load_dword %0, %3, off offset:4 sc1\n\ts_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\t"
                 "s_cmp_eq_u32 %1, %4\n\ts_cbranch_scc1 2f\n\ts_sub_u32 %2, %2, 1\n\ts_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 2f\n\t"
                 "s_cmp_eq_u32 %5, 0\n\ts_cbranch_scc1 3f\n\ts_sleep 32\n\ts_branch 1b\n3:\n\ts_sleep 8\n\ts_branch 1b\n2:"
                 : "=&v"(v), "=&s"(tag), "+s"(cnt) : "v"(g), "s"(want), "s"(sleep_long) : "memory", "scc");
    return tag;
}

// per-lane agent-scope load
__device__ __forceinline__ uint32_t vq_ldv(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ bool vw_stop(const VcPipe& p) { return (vq_ldu(p.finished) | vq_ldu(p.abort_code)) != 0u; }

// after the wave's release.  The granule of ticket t - capacity was emptied by its consumer long ago (a window has one item in
// flight and the capacity is at least the number of windows); waiting for the empty granule makes that a checked fact.
__device__ __forceinline__ void vw_push(const VcPipe& p, const VcQueue& q, uint32_t value) {
    const uint32_t t = vq_inc(q.res);
    unsigned long long* s = &q.slots[t & q.mask];
    const unsigned long long t0 = wall_clock64();
    uint32_t tag = 1u, late = 0u;
    while (tag != 0u && late == 0u) {
        tag = vq_poll_tag(s, 0u, 64u, 0u);
        if (tag != 0u && wall_clock64() - t0 > (unsigned long long)p.spin_limit) late = 1u;
    }
    if (tag != 0u) { vq_stu(p.abort_code, 9u); return; }
    vq_stu64(s, value, (t >> q.shift) + 1u);
}
// the value behind ticket t; VC_Q_NONE when the pipeline has finished or aborted
__device__ __forceinline__ uint32_t vw_wait(const VcPipe& p, const VcQueue& q, uint32_t t, uint32_t site) {
    const uint32_t want = (t >> q.shift) + 1u;
    unsigned long long* s = &q.slots[t & q.mask];
    const unsigned long long t0 = wall_clock64();
    uint32_t tag = 0u, state = 0u;                            // state 1: got it; 2: stop seen; 3: out of patience
    while (state == 0u) {
        tag = vq_poll_tag(s, want, 8u, 1u);
        if (tag == want) state = 1u;
        else if (vw_stop(p)) state = 2u;
        else if (wall_clock64() - t0 > (unsigned long long)p.spin_limit) state = 3u;
    }
    if (state == 3u) vq_stu(p.abort_code, site);
    if (state != 1u) return VC_Q_NONE;
    const uint32_t val = vq_ldu(reinterpret_cast<const uint32_t*>(s));         // the granule was written by ONE 8-byte store: the value stands behind its tag
    vq_stu64(s, 0u, 0u);                                                        // empty again
    return val;
}
// the value behind ticket t when it is there within a few polls, else VC_Q_NONE -- the ticket then stays the caller's to wait for later
__device__ __forceinline__ uint32_t vw_peek(const VcQueue& q, uint32_t t) {
    const uint32_t want = (t >> q.shift) + 1u;
    unsigned long long* s = &q.slots[t & q.mask];
    if (vq_poll_tag(s, want, 4u, 0u) != want) return VC_Q_NONE;
    const uint32_t val = vq_ldu(reinterpret_cast<const uint32_t*>(s));
    vq_stu64(s, 0u, 0u);
    return val;
}
// a ticket only if an item stands behind it (reserved by its producer, written at once)
__device__ __forceinline__ uint32_t vw_try(const VcPipe& p, const VcQueue& q, uint32_t site) {
    uint32_t got = VC_Q_NONE;
    for (int tries = 0; tries < 4 && got == VC_Q_NONE; ++tries) {
        const uint32_t h = vq_ldc(q.head), r = vq_ldc(q.res);
        if ((int)(r - h) <= 0) tries = 4;
        else if (vq_cas(q.head, h, h + 1u) == h) got = h;
    }
    return got == VC_Q_NONE ? VC_Q_NONE : vw_wait(p, q, got, site);
}

// everything this wave wrote becomes visible to the device
__device__ __forceinline__ void vp_release() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain: the write-back below covers what has ARRIVED in the L2, not stores still in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop its own wait behind the write-back: MI355X_MICROARCH.md)
}
__device__ __forceinline__ void vp_acquire() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // uniform loads may have gone through the scalar cache in an earlier round
}
// blocking take
__device__ __forceinline__ uint32_t vp_take(const VcPipe& p, const VcQueue& q, uint32_t site) {
    const uint32_t v = vw_wait(p, q, vq_inc(q.head), site);
    if (v != VC_Q_NONE) vp_acquire();
    return v;
}
// the window leaves the pipeline
__device__ __forceinline__ void vp_retire(const VcPipe& p) {
    if (vq_inc(p.done) + 1u == vq_ldu(p.n_active)) vq_stu(p.finished, 1u);
}

// windows with at least two layers enter at layer 1 (window.cpp:188-192: fewer than three sequences keep their backbone, k_init)
__global__ void k_pipe_seed(VcBatchDev b, VcPipe p, uint32_t w0, uint32_t nslots) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const uint32_t w = w0 + slot;
    const uint32_t ns = b.win_seq_off[w + 1] - b.win_seq_off[w];
    if (ns < 3 || b.status[w] != VC_WIN_OK) return;
    p.cur_layer[slot] = 1;
    atomicAdd(p.n_active, 1u);
    const uint32_t t = atomicAdd(p.fq.res, 1u);               // the queue is empty and nobody consumes yet: no wait
    p.fq.slots[t & p.fq.mask] = ((unsigned long long)((t >> p.fq.shift) + 1u) << 32) | (slot | (VC_QK_START << 24));
}

struct VcPipeFwdArgs {
    VcFwdArgs fa;                  // mode 0, group 1, band 1
    VcAddArgs aa;                  // make_rows 1
    VcPipe p;
    uint32_t* submask;             // Subgraph membership of partial-span layers (k_rows_sub)
    int force_fail_site;           // test hook: 0
};

// The forward wave.  dynamic LDS: max(kept-row ring, AddAlignment notes, row builders' scratch), all phases of one wave in turn
#ifndef VC_PIPE_FWD_OCC
#define VC_PIPE_FWD_OCC __attribute__((amdgpu_waves_per_eu(5, 8)))      // 96 registers.  (6 / 7 waves per SIMD -- 80 / 72 registers -- were measured:
                                                                         // one reload in the row loop and the forward pass of an item takes 1.77 instead of 0.75 ms)
#endif
#ifndef VC_PIPE_TRACE_OCC
#define VC_PIPE_TRACE_OCC __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
template <int CA, int CB, int RING, bool PACKED>
__global__ __launch_bounds__(64) VC_PIPE_FWD_OCC void k_pipe_fwd(VcPipeFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = vc_lane();
    if (vq_ldu(a.p.n_active) == 0 || vq_ldu(a.p.abort_code) != 0) return;
    const VcBatchDev& b = a.fa.b;
    unsigned long long t_wait = 0, t_add = 0, t_fwd = 0, t_hand = 0, n_items = 0;
    const unsigned long long t_born = wall_clock64();
    for (;;) {
        const unsigned long long c0 = wall_clock64();
        const uint32_t it = vp_take(a.p, a.p.fq, 1);
        if (it == VC_Q_NONE) break;
        const unsigned long long c1 = wall_clock64();
        t_wait += c1 - c0; n_items++;
        if (a.p.prof && lane == 0) atomicAdd(a.p.prof + VC_PP_N + min((uint32_t)((c1 - t_born) / VC_PP_TL_TICKS), (uint32_t)VC_PP_TL - 1u), c1 - c0);
        const uint32_t slot = it & VC_Q_SLOT_MASK, kind = it >> 24;
        if (a.p.pub_time && lane == 0 && kind != VC_QK_START) atomicAdd(a.p.prof + VC_PP_F_PICKUP, c1 - a.p.pub_time[slot]);
        const uint32_t w = a.fa.w0 + slot;
        const uint32_t s0 = b.win_seq_off[w], ns = b.win_seq_off[w + 1] - s0;
        const uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - b.seq_off[s0]);
        uint32_t layer = vq_ldu(&a.p.cur_layer[slot]);
        bool alive = b.status[w] == VC_WIN_OK;
        if (alive && kind == VC_QK_ADD) {
            vc_addaln_body(a.aa, smem, slot, layer, blockIdx.x);            // + the row records of layer + 1 when that one is full-span
            __syncthreads();
            layer++;
            alive = layer < ns && b.status[w] == VC_WIN_OK;
            if (alive) vq_stu(&a.p.cur_layer[slot], layer);
        }
        if (alive && kind != VC_QK_REDO && !vc_full_span(b.seq_begin[s0 + layer], b.seq_end[s0 + layer], L)) {
            // a partial-span layer aligns to Graph::Subgraph (graph.cpp:640-732)
            vc_rows_sub_body(b, a.aa.g, a.aa.dp, a.fa.w0, a.fa.nslots, a.fa.NC, a.fa.EC, (int)layer, a.aa.ring, a.submask, a.aa.kept, smem, slot);
            __syncthreads();
            alive = b.status[w] == VC_WIN_OK;
        }
        const unsigned long long c2 = wall_clock64();
        t_add += c2 - c1;
        if (!alive) {
            vp_retire(a.p);
            continue;
        }
        VcJob jb;
        jb.job = slot; jb.slot = slot; jb.k = layer; jb.redo = kind == VC_QK_REDO;
        uint32_t r;
        {
            uint32_t cls = CB;
            if (CA != CB) cls = vc_cpl_for((uint32_t)(b.seq_off[s0 + layer + 1] - b.seq_off[s0 + layer]));
            if (CA != CB && cls <= (uint32_t)CA) r = vc_fwd_body<CA, RING, true, PACKED, true, true>(a.fa, reinterpret_cast<uint32_t*>(smem), jb);
            else r = vc_fwd_body<CB, RING, true, PACKED, true, true>(a.fa, reinterpret_cast<uint32_t*>(smem), jb);
        }
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        __syncthreads();
        const unsigned long long c3 = wall_clock64();
        t_fwd += c3 - c2;
        if (r == VC_FWD_NONE) {
            // the packed-int16 pass declined the alignment (the host plans this pipeline only when no alignment can leave its
            // envelope): report it, never skip silently
            if (lane == 0) { if (b.status[w] == VC_WIN_OK) vc_fail(b, w, VC_WIN_INVALID, 30, layer); }
            vp_release();
            vp_retire(a.p);
            continue;
        }
        if (a.p.pub_time && lane == 0) a.p.pub_time[slot] = wall_clock64();
        vp_release();
        vw_push(a.p, a.p.tq, slot | ((r == VC_FWD_TIE ? VC_QK_TIE : jb.redo ? VC_QK_REDO : 0u) << 24));
        const unsigned long long c4 = wall_clock64();
        t_hand += c4 - c3;
        if (a.p.prof && lane == 0) atomicAdd(a.p.prof + VC_PP_N + VC_PP_TL + min((uint32_t)((c4 - t_born) / VC_PP_TL_TICKS), (uint32_t)VC_PP_TL - 1u), c4 - c1);
    }
    if (a.p.prof && lane == 0) {
        atomicAdd(a.p.prof + VC_PP_F_WAIT, t_wait); atomicAdd(a.p.prof + VC_PP_F_ADD, t_add); atomicAdd(a.p.prof + VC_PP_F_FWD, t_fwd);
        atomicAdd(a.p.prof + VC_PP_F_HAND, t_hand); atomicAdd(a.p.prof + VC_PP_F_ITEMS, n_items); atomicAdd(a.p.prof + VC_PP_F_WAVES, 1ull);
    }
}

struct VcPipeTraceArgs {
    VcTraceArgs ta;                // group 1, pair_group 1, shared_table 0
    VcPipe p;
    // end-cell ties (vc_resolve_one)
    VcGraph g; uint32_t STK;
    const uint16_t* tie_rows; const uint32_t* tie_cnt; const uint32_t* tie_over; uint32_t tie_over_stride; uint32_t* job_end;
    const uint32_t* submask; uint8_t* workspace; uint32_t ws_bytes; int force_dfs;
};

// end-cell ties of the windows a backtrack wave has just taken, one window at a time with the whole wave.  Not inlined: the exact
// resolver wants more registers than the walk, and inlined it pushed ninety of the walk's registers into scratch.
__device__ __attribute__((noinline)) uint32_t vc_pipe_ties(const VcPipeTraceArgs& a, uint8_t* smem, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t n = 0;
    for (int g2 = 0; g2 < VC_TG; ++g2) {
        const uint32_t x = g2 == 0 ? v0 : g2 == 1 ? v1 : g2 == 2 ? v2 : v3;
        if (x == VC_Q_NONE || (x >> 24) != VC_QK_TIE) continue;
        const uint32_t slot = x & VC_Q_SLOT_MASK;
        const uint32_t layer = vq_ldu(&a.p.cur_layer[slot]);
        __syncthreads();
        vc_resolve_one(smem, a.workspace + (size_t)blockIdx.x * a.ws_bytes, slot, a.ta.b, a.g, a.ta.dp, a.ta.w0, a.ta.nslots, a.ta.NC, a.ta.EC, a.STK,
                       a.tie_rows, a.tie_cnt, a.tie_over, a.tie_over_stride, a.job_end, a.submask, (int)layer, a.force_dfs);
        __syncthreads();
        n++;
    }
    return n;
}

// The backtrack wave: up to VC_TG windows per round, whatever the queue holds
__global__ __launch_bounds__(64) VC_PIPE_TRACE_OCC void k_pipe_trace(VcPipeTraceArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = vc_lane();
    const uint32_t grp = (uint32_t)lane / VC_TL;
    if (vq_ldu(a.p.n_active) == 0 || vq_ldu(a.p.abort_code) != 0) return;
    unsigned long long t_wait = 0, t_walk = 0, t_hand = 0, t_tie = 0, n_rounds = 0, n_items = 0, n_ties = 0;
    const unsigned long long t_born = wall_clock64();
    uint32_t owe[VC_TG - 1] = {VC_Q_NONE, VC_Q_NONE, VC_Q_NONE};       // tickets this wave drew whose items had not arrived when it looked
    for (;;) {
        const unsigned long long c0 = wall_clock64();
        // The first item: a blocking wait -- the wave holds nothing, so waiting cannot keep a window from retiring.  Tickets drawn in
        // an earlier round whose items had not arrived yet ("owed") come first.
        uint32_t tk0;
        if (owe[0] != VC_Q_NONE) { tk0 = owe[0]; owe[0] = owe[1]; owe[1] = owe[2]; owe[2] = VC_Q_NONE; }
        else tk0 = vq_inc(a.p.tq.head);
        uint32_t vv[VC_TG] = {vw_wait(a.p, a.p.tq, tk0, 2), VC_Q_NONE, VC_Q_NONE, VC_Q_NONE};
        if (vv[0] != VC_Q_NONE) {
            // The other groups: only what stands in the queue NOW.  A wave that holds an item never blocks on a further ticket
            // (many waves that see a backlog at once can together draw more tickets than there are items; at the end of a chunk
            // nothing would fill the surplus, the windows held would never retire, and every wave would wait out its patience):
            // a ticket whose item is not there yet stays owed and is waited for when the wave's hands are empty.
            uint32_t n = 1, no = 0;
            uint32_t keep[VC_TG - 1] = {VC_Q_NONE, VC_Q_NONE, VC_Q_NONE};
#pragma unroll
            for (int i = 0; i < VC_TG - 1; ++i) {
                if (owe[i] == VC_Q_NONE) continue;
                const uint32_t x = vw_peek(a.p.tq, owe[i]);
                if (x != VC_Q_NONE) {
#pragma unroll
                    for (int j = 1; j < VC_TG; ++j) if (n == (uint32_t)j) vv[j] = x;
                    n++;
                } else {
#pragma unroll
                    for (int j = 0; j < VC_TG - 1; ++j) if (no == (uint32_t)j) keep[j] = owe[i];
                    no++;
                }
            }
#pragma unroll
            for (int i = 0; i < VC_TG - 1; ++i) owe[i] = keep[i];
            const uint32_t room = (uint32_t)VC_TG - n - no;      // held + owed never exceed the groups of the wave
            if (room) {
                // With a backlog the tickets are simply drawn (a compare-and-swap on a counter that a thousand waves advance every
                // half microsecond never succeeds: the first version took 1.4 windows per round while 12 000 were queued);
                // near-empty, one at a time and only behind an item that stands there
                const int avail = (int)(vq_ldc(a.p.tq.res) - vq_ldc(a.p.tq.head));
                if (avail >= 64) {
                    const uint32_t t1 = vq_add(a.p.tq.head, room);
#pragma unroll
                    for (uint32_t r = 0; r < (uint32_t)VC_TG - 1; ++r) {
                        if (r >= room) continue;
                        const uint32_t x = vw_peek(a.p.tq, t1 + r);
                        if (x != VC_Q_NONE) {
#pragma unroll
                            for (int j = 1; j < VC_TG; ++j) if (n == (uint32_t)j) vv[j] = x;
                            n++;
                        } else {
#pragma unroll
                            for (int j = 0; j < VC_TG - 1; ++j) if (no == (uint32_t)j) owe[j] = t1 + r;
                            no++;
                        }
                    }
                } else if (avail > 0) {
                    bool more = true;
#pragma unroll
                    for (uint32_t r = 0; r < (uint32_t)VC_TG - 1; ++r) {
                        if (r >= room || !more) continue;
                        const uint32_t x = vw_try(a.p, a.p.tq, 3);
                        if (x == VC_Q_NONE) { more = false; continue; }
#pragma unroll
                        for (int j = 1; j < VC_TG; ++j) if (n == (uint32_t)j) vv[j] = x;
                        n++;
                    }
                }
            }
        }
        const uint32_t v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = vv[3];
        if (v0 == VC_Q_NONE) break;
        vp_acquire();
        const unsigned long long c1 = wall_clock64();
        t_wait += c1 - c0; n_rounds++;
        if (a.p.prof && lane == 0) {
            const uint32_t bk = min((uint32_t)((c1 - t_born) / VC_PP_TL_TICKS), (uint32_t)VC_PP_TL - 1u);
            atomicAdd(a.p.prof + VC_PP_N + 2 * VC_PP_TL + bk, 1ull);
            atomicAdd(a.p.prof + VC_PP_N + 3 * VC_PP_TL + bk, (unsigned long long)((v0 != VC_Q_NONE) + (v1 != VC_Q_NONE) + (v2 != VC_Q_NONE) + (v3 != VC_Q_NONE)));
        }
        // ties first, one window at a time with the whole wave (the LDS of the first-in-edge tables is free until the walk starts)
        {
            const unsigned long long ct0 = wall_clock64();
            const uint32_t nt = vc_pipe_ties(a, smem, v0, v1, v2, v3);
            n_ties += nt; n_items += (v0 != VC_Q_NONE) + (v1 != VC_Q_NONE) + (v2 != VC_Q_NONE) + (v3 != VC_Q_NONE);
            if (nt) t_tie += wall_clock64() - ct0;
        }
        const uint32_t mine = grp == 0 ? v0 : grp == 1 ? v1 : grp == 2 ? v2 : v3;
        const bool valid = mine != VC_Q_NONE;
        const uint32_t slot = valid ? (mine & VC_Q_SLOT_MASK) : 0u;
        const bool redo = valid && ((mine >> 24) == VC_QK_REDO);
        const uint32_t k = vq_ldv(&a.p.cur_layer[slot]);          // (slot 0 for an idle group: a valid address)
        if (a.p.pub_time && valid && (lane % VC_TL) == 0) atomicAdd(a.p.prof + VC_PP_T_PICKUP, c1 - a.p.pub_time[slot]);
        const bool gredo = vc_tracew_body<true>(a.ta, smem, slot, slot, k, (uint64_t)slot, valid, redo);
        __syncthreads();
        const unsigned long long c2 = wall_clock64();
        t_walk += c2 - c1;
        if (a.p.pub_time && valid && (lane % VC_TL) == 0) a.p.pub_time[slot] = c2;
        vp_release();
        // (a window whose walk failed carries its status; the forward wave retires it)
        const uint32_t back = valid ? (slot | ((gredo && !redo ? VC_QK_REDO : VC_QK_ADD) << 24)) : VC_Q_NONE;
#pragma unroll
        for (int g2 = 0; g2 < VC_TG; ++g2) {
            const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)back, g2 * VC_TL);
            if (x != VC_Q_NONE) vw_push(a.p, a.p.fq, x);
        }
        t_hand += wall_clock64() - c2;
    }
    if (a.p.prof && lane == 0) {
        atomicAdd(a.p.prof + VC_PP_T_WAIT, t_wait); atomicAdd(a.p.prof + VC_PP_T_WALK, t_walk); atomicAdd(a.p.prof + VC_PP_T_HAND, t_hand);
        atomicAdd(a.p.prof + VC_PP_T_ROUNDS, n_rounds); atomicAdd(a.p.prof + VC_PP_T_ITEMS, n_items); atomicAdd(a.p.prof + VC_PP_T_WAVES, 1ull);
        atomicAdd(a.p.prof + VC_PP_T_TIES, n_ties); atomicAdd(a.p.prof + VC_PP_T_TIETIME, t_tie);
    }
}
