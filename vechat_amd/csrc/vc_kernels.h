// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for VeChat's per-window hot path.
// Written for this target only: DPP wave scans, LDS row ring, 16-byte coalesced HBM traffic.
//
// Kernel            parallelism            restates (reference file:line)
//   k_avg           wave / window          window.cpp:216-236,283,292-309 (average_weight, fp64, additions in order)
//   k_init          wave / window          window.cpp:188-201 + graph.cpp:109-130,182-212 (backbone chain)
//   k_rows          wave / window          row records from the incrementally kept order (full-span layers)
//   k_rows_sub      wave / window          graph.cpp:640-732 Subgraph membership + row records (partial-span layers)
//   k_topo          wave / window          graph.cpp:301-371 TopologicalSort (exact DFS) + row records, after a prune
//   k_fwd<CA,CB>    wave / alignment       sisd_alignment_engine.cpp:118-254 Initialize, :292-360 Linear (NW/SW)
//   k_resolve       wave / tied alignment  sisd :353-355 "first sink in rank order" among equal end scores
//   k_tracew        16 lanes / alignment   sisd_alignment_engine.cpp:362-459 (backtrack), speculative and cooperative
//   k_trace         thread / alignment     the same backtrack, plain (cross-check, VC_TRACE_THREAD=1)
//   k_addaln        wave / window          graph.cpp:182-299 AddAlignment (+ :94-107 AddEdge) + order maintenance
//   k_prune_lcc     wave / window          graph.cpp:811-982 PruneGraph, :984-1102 DfsUtil/LargestSubgraph
//   k_addw          wave / window          graph.cpp:1104-1165 AddWeights (+ window.cpp:351-372 weights)
//   k_finish        wave / window          graph.cpp:1167-1179 GenerateCorrectedSequence
//   k_consensus     wave / window          graph.cpp:450-638 GenerateConsensus (+ window.cpp:141-171 trim), racon-linear overload
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "vc_device.h"
#include "vechat_hip.h"

#define VC_INT_MIN (-2147483647 - 1)
// linkage of the kernels: external in vc_api.hip; `static` in a second translation unit that includes this header for a few templates only
// (vc_fwdn.hip), so that the host stubs of the non-template kernels are not defined twice
#ifndef VC_KL
#define VC_KL
#endif
#define VC_RING_PRUNED_N 4     // rows of the plain ring on pruned graphs (vc_api.hip: VC_RING_PRUNED; vc_fwdn.hip instantiates k_fwdn with it)
// The latency-bound single-lane kernels of one chunk run next to the throughput-bound k_fwd of another
// chunk (separate streams).  The CU issues the oldest ready wave first, which starves them; raising the
// wave priority lets their few instructions through at once and costs k_fwd almost nothing.
#define VC_LATENCY_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)

__device__ __forceinline__ int vc_lane() { return (int)(threadIdx.x & 63); }

// record why a window left the OK state (site id, detail) -- diagnostics only
__device__ __forceinline__ void vc_fail(const VcBatchDev& b, uint32_t w, int status, uint32_t site, uint32_t detail) {
    b.status[w] = (uint8_t)status;
    b.errinfo[w] = (site << 16) | (detail & 0xFFFF);
}

// exclusive prefix sum over the 64 lanes of the wave; total returned through `total`
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t x, uint32_t& total) {
    uint32_t v = x;
    const int lane = vc_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    total = __shfl(v, 63, 64);
    return v - x;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

__device__ __forceinline__ uint32_t vc_weight(const VcBatchDev& b, uint64_t off, uint32_t q, bool has_qual) {
    return has_qual ? b.lut_w[b.quals[off + q]] : 1u;
}

// columns-per-lane class of a sequence: the smallest instantiated k_fwd width that holds it
__host__ __device__ inline uint32_t vc_cpl_for(uint32_t len) {
    const uint32_t opts[11] = {4, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    for (int i = 0; i < 11; ++i) if (64 * opts[i] >= len) return opts[i];
    return 0;
}

// Does a linear-gap alignment of `len` columns against `nrows` rows stay inside int16 in the form k_fwd computes it?  The reference
// asks this of the plain matrix H, whose worst case is g * (len + nrows) (simd_alignment_engine_implementation.hpp:699-706), and
// switches to 32-bit lanes beyond it; that decides its speed, not its result.  k_fwd works on the TILTED matrix T = H - j*g: a
// horizontal step leaves T unchanged, so along any path a cell falls only by the rows it crosses -- T >= -rows * max(|g|, |n - g|)
// for a global alignment (column 0 is H[i][0] >= g * i, the virtual row is 0), T >= 0 for a local one -- and rises by at most
// m - g per column.  1 024 of headroom below, as before.  (Round 4: 3 kb reads against a 7 000-row graph fit; they used to go
// to the int32 kernel on H's account.)
__host__ __device__ inline bool vc_int16_ok(long long m, long long n, long long g, long long nrows, long long cpl, bool nw) {
    if (g >= 0 || (!nw && n >= 0)) return false;
    long long dec = -g;
    if (-(n - g) > dec) dec = -(n - g);
    const long long lo = nw ? -(nrows + 8) * dec : 0;
    return lo >= -31744 && (m - g) * (64 * cpl + 1) < 32767;
}

// floor(x / d) == umulhi(x, vc_magic(d)) for x < 65536 and 2 <= d <= 64
__host__ __device__ inline uint32_t vc_magic(uint32_t d) { return (uint32_t)((0x100000000ull + d - 1) / d); }

__device__ __forceinline__ bool vc_full_span(uint32_t begin, uint32_t end, uint32_t L) {
    uint32_t offset = (uint32_t)(0.01 * (double)L);          // window.cpp:212
    return begin < offset && end > L - offset;              // window.cpp:253-254
}

// the backtrack's first-in-edge entry of a row record (VcDp::fie)
__device__ __forceinline__ uint8_t vc_fie_entry(uint32_t rec_x, uint32_t rec_y) {
    const uint32_t d0 = rec_y & 0xFFFFu;
    return (uint8_t)((((rec_x >> 8) & VC_RF_OVF) || d0 > 15u) ? 0u : d0);
}
#define VC_FIE_STRIDE(NC) ((NC) + 4u)

// The forward kernel's view of a row (16 B): byte0 code, byte1 flags, byte2 number of listed predecessors,
// byte3 base index (0..3 = ACGT, 4 = other), then up to 6 x u16 row distances of the predecessors OTHER than
// the row directly above (that one is flagged VC_RF_PREV and taken from registers).  The order of the
// in-edges does not matter to the forward pass.  VC_RF_SLOW marks rows whose list needs the general path:
// the virtual row 0, a row older than the LDS ring, or more predecessors than fit (list in VcDp::ovf).
__device__ __forceinline__ uint4 vc_make_frec(uint32_t code, uint32_t fl, uint32_t np, const uint16_t (&dl)[VC_INLINE_PRED],
                                              bool is_ovf, bool hasprev, uint32_t r, uint32_t ring) {
    uint32_t f = fl & (VC_RF_SINK | VC_RF_OVF);
    uint16_t out[VC_INLINE_PRED];
#pragma unroll
    for (int k = 0; k < VC_INLINE_PRED; ++k) out[k] = 0;
    uint32_t nq = 0;
    bool slow = is_ovf;
    if (hasprev) f |= VC_RF_PREV;
    if (is_ovf) {
        out[0] = dl[0]; out[1] = dl[1];                      // offset into VcDp::ovf
        out[2] = dl[2]; out[3] = dl[3];                      // number of entries there
        nq = min(np, 255u);
    } else {
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) {
            if ((uint32_t)k < np && !(hasprev && dl[k] == 1)) {
                const uint32_t d = dl[k];
#pragma unroll
                for (int t = 0; t < VC_INLINE_PRED; ++t) if (nq == (uint32_t)t) out[t] = (uint16_t)d;
                nq++;
                if (d > ring || d == r + 1) slow = true;
            }
        }
    }
    if (slow) f |= VC_RF_SLOW;
    if (hasprev && nq == 0 && !slow) f |= VC_RF_PLAIN;
    const uint32_t bi = code == 'A' ? 0u : code == 'C' ? 1u : code == 'G' ? 2u : code == 'T' ? 3u : 4u;
    uint4 o;
    o.x = code | (f << 8) | (nq << 16) | (bi << 24);
    o.y = out[0] | ((uint32_t)out[1] << 16);
    o.z = out[2] | ((uint32_t)out[3] << 16);
    o.w = out[4] | ((uint32_t)out[5] << 16);
    return o;
}

// ------------------------------------------------------------------------------------------------
// k_avg: average_weight per window -- a strictly ordered fp64 sum (window.cpp:225-236,283,292-309)
// ------------------------------------------------------------------------------------------------
// One wave per window.  The sum is strictly sequential in the reference (a double accumulated base by base
// in rank order, window.cpp:225-236,283,292-296), so the ORDER of the additions is kept; what is parallel is
// only the fetch: 64 lanes load 64 qualities and their table values at once, then every lane performs the
// same 64 dependent additions on values broadcast with v_readlane.
VC_KL __global__ __launch_bounds__(64) void k_avg(VcBatchDev b, uint32_t w0, uint32_t nw) {
    VC_LATENCY_KERNEL_PRIO();
    const uint32_t t = blockIdx.x;
    if (t >= nw) return;
    const int lane = vc_lane();
    const uint32_t w = w0 + t;
    const uint32_t s0 = b.win_seq_off[w], s1 = b.win_seq_off[w + 1];
    const uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - b.seq_off[s0]);
    const bool fasta = b.win_fasta[w] != 0;
    double total = 0.0;
    auto add_quals = [&](uint64_t o, uint32_t len) __attribute__((always_inline)) {
        for (uint32_t q0 = 0; q0 < len; q0 += 64) {
            const uint32_t cnt = min(64u, len - q0);
            const double v = (uint32_t)lane < cnt ? b.lut_d[b.quals[o + q0 + lane]] : 0.0;
            const int lo = __double2loint(v), hi = __double2hiint(v);
            for (uint32_t k = 0; k < cnt; ++k)
                total += __hiloint2double(__builtin_amdgcn_readlane(hi, (int)k), __builtin_amdgcn_readlane(lo, (int)k));
        }
    };
    if (fasta) total += (double)L;
    else add_quals(b.seq_off[s0], L);
    for (uint32_t s = s0 + 1; s < s1; ++s) {
        const uint64_t o = b.seq_off[s];
        const uint32_t len = (uint32_t)(b.seq_off[s + 1] - o);
        if (!b.seq_has_qual[s]) total += (double)len;
        else add_quals(o, len);
    }
    const uint16_t wl = (uint16_t)L;                          // window.cpp:216
    const double avg = fasta ? 2.0 * total / wl : 2.0 * total / wl * 1000;
    if (lane == 0) b.win_avg[w] = avg;
}

__device__ __forceinline__ void vc_rows_full(const VcBatchDev& b, const VcGraph& g, const VcDp& dp, uint32_t slot, uint32_t w,
                                             uint32_t NC, uint32_t EC, int next_layer, uint32_t ring, uint32_t N, uint32_t kept, uint8_t* lds);

// ------------------------------------------------------------------------------------------------
// k_init: backbone chain graph (AddAlignment with an empty alignment, graph.cpp:207-212)
// ------------------------------------------------------------------------------------------------
VC_KL __global__ __launch_bounds__(64) void k_init(VcBatchDev b, VcGraph g, VcDp dp, uint32_t w0, uint32_t nslots,
                                             uint32_t NC, uint32_t EC, uint32_t ring, uint32_t kept, uint32_t* cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];          // kept != 0: vc_kept_lds_bytes(NC)
    uint32_t slot = blockIdx.x;
    if (slot >= nslots) return;
    if (cursor && threadIdx.x == 0) cursor[slot] = 1u;                       // the build loop starts at layer 1
    uint32_t w = w0 + slot;
    const int lane = vc_lane();
    uint32_t s0 = b.win_seq_off[w], ns = b.win_seq_off[w + 1] - s0;
    uint64_t o0 = b.seq_off[s0];
    uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - o0);
    if (ns < 3) {                                             // window.cpp:188-192
        uint32_t n = L < b.cons_cap ? L : b.cons_cap;
        for (uint32_t i = lane; i < n; i += 64) b.cons[(uint64_t)w * b.cons_cap + i] = b.bases[o0 + i];
        if (lane == 0) { b.cons_len[w] = n; b.status[w] = L <= b.cons_cap ? VC_WIN_UNPOLISHED : VC_WIN_OVERFLOW; }
        return;
    }
    if (b.status[w] != VC_WIN_OK) return;                    // marked at submit (outside the envelope)
    if (L > NC || L > EC + 1 || L >= 0xFFFF) { if (lane == 0) vc_fail(b, w, VC_WIN_OVERFLOW, 1, L); return; }
    uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    for (uint32_t i = lane; i < L; i += 64) {
        g.code[nb + i] = b.bases[o0 + i];
        uint16_t ein = i > 0 ? (uint16_t)(i - 1) : VC_NONE16;
        uint16_t eout = i + 1 < L ? (uint16_t)i : VC_NONE16;
        g.in_first[nb + i] = ein; g.in_last[nb + i] = ein;
        g.out_first[nb + i] = eout; g.out_last[nb + i] = eout;
        g.al_cnt[nb + i] = 0;
        g.ord[nb + i] = (uint16_t)i; g.pos[nb + i] = (uint16_t)i;
        g.visits[nb + i] = L >= 2 ? 1 : 0;
        g.nrec[nb + i] = make_uint4((uint32_t)b.bases[o0 + i] | (i > 0 ? 1u << 16 : 0u), i > 0 ? i - 1 : 0u, 0u, 0u);
        if (i + 1 < L) {
            uint32_t wgt = b.lut_w[b.quals[o0 + i]] + b.lut_w[b.quals[o0 + i + 1]];
            g.e_tn[eb + i] = i | ((uint32_t)VC_NONE16 << 16);
            g.e_hn[eb + i] = (i + 1) | ((uint32_t)VC_NONE16 << 16);
            g.e_w[eb + i] = wgt;
        }
    }
    if (lane == 0) { g.n_nodes[slot] = L; g.n_edges[slot] = L - 1; }
    __syncthreads();
    vc_rows_full(b, g, dp, slot, w, NC, EC, 1, ring, L, kept, smem);          // rows of the first layer's alignment
}

// ------------------------------------------------------------------------------------------------
// k_topo: exact TopologicalSort (graph.cpp:301-371) of the graph -- or of the Subgraph the next
// partial-span layer will be aligned to (graph.cpp:640-732, expressed as a node mask) -- followed
// by the construction of the row records k_fwd streams.  The order-defining DFS is inherently
// serial; it runs on lane 0 out of LDS, everything around it is wave-parallel.
//   next_layer >= 0 : build phase, prepare for sequence index `next_layer` of each window
//   next_layer <  0 : whole graph (pruned graphs)
// LDS carve (bytes): in_first 2*NC | etn 4*EC | al 8*NC | flag NC | stack 2*STK | rank 2*NC
// ------------------------------------------------------------------------------------------------
#define TF_MARK  0x03
#define TF_IGN   0x04
#define TF_SUB   0x08

struct VcTopoLds {
    uint16_t* in_first; uint32_t* etn; uint16_t* al; uint8_t* flag; uint8_t* alc; uint16_t* stack; uint16_t* rank;
    uint32_t ma;                  // entries per aligned list (VcGraph::ma)
};

__device__ __forceinline__ VcTopoLds vc_topo_carve(uint8_t* smem, uint32_t NC, uint32_t EC, uint32_t STK, uint32_t MA) {
    VcTopoLds t;
    t.ma = MA;
    t.in_first = (uint16_t*)smem;
    t.etn = (uint32_t*)(smem + ((2 * NC + 15) & ~15u));
    t.al = (uint16_t*)((uint8_t*)t.etn + 4 * EC);
    t.flag = (uint8_t*)t.al + 2 * MA * NC;
    t.alc = t.flag + ((NC + 15) & ~15u);
    t.stack = (uint16_t*)(t.alc + ((NC + 15) & ~15u));
    t.rank = t.stack + STK;
    return t;
}

// cooperative copy of the order-defining part of the graph (in-lists, aligned lists) into LDS
__device__ __forceinline__ void vc_topo_load(const VcGraph& g, uint64_t nb, uint64_t eb, uint32_t N, uint32_t E,
                                             const VcTopoLds& t, int lane) {
    for (uint32_t i = lane; i < N; i += 64) {
        t.in_first[i] = g.in_first[nb + i];
        t.flag[i] = 0;
        t.alc[i] = g.al_cnt[nb + i];
    }
    for (uint32_t i = lane; i < E; i += 64) t.etn[i] = g.e_tn[eb + i];
    const uint32_t* src = (const uint32_t*)(g.al + nb * g.ma);            // ma is even: whole dwords
    uint32_t* dst = (uint32_t*)t.al;
    for (uint32_t i = lane; i < N * (g.ma / 2); i += 64) dst[i] = src[i];
}

// The serial, order-defining part: ExtractSubgraph flood (when masked) + TopologicalSort DFS.
// Runs on ONE lane; returns 0 or a VC_WIN_* status, the number of emitted rows through nrows_out.
__device__ int vc_topo_dfs(const VcTopoLds& ls, uint32_t N, uint32_t STK, bool masked, uint32_t mb, uint32_t me,
                           uint32_t* nrows_out) {
    int err = 0;
    uint32_t nr = 0;
    if (masked) {
        // ExtractSubgraph(nodes_[end], nodes_[begin]), graph.cpp:640-666
        if (me >= N || mb >= N) { err = VC_WIN_INVALID; }
        else {
            uint32_t sp = 0;
            ls.stack[sp++] = (uint16_t)me;
            while (sp && !err) {
                uint32_t c = ls.stack[--sp];
                if (!(ls.flag[c] & TF_SUB) && c >= mb) {
                    for (uint32_t e = ls.in_first[c]; e != VC_NONE16; ) {
                        uint32_t tn = ls.etn[e];
                        if (sp >= STK) { err = VC_WIN_OVERFLOW; break; }
                        ls.stack[sp++] = (uint16_t)(tn & 0xFFFF);
                        e = tn >> 16;
                    }
                    uint32_t cnt = ls.alc[c];
                    for (uint32_t k = 0; k < cnt; ++k) {
                        if (sp >= STK) { err = VC_WIN_OVERFLOW; break; }
                        ls.stack[sp++] = ls.al[c * ls.ma + k];
                    }
                    ls.flag[c] |= TF_SUB;
                }
            }
        }
    }
    const uint8_t need = masked ? TF_SUB : 0;
    for (uint32_t s = 0; s < N && !err; ++s) {
        uint8_t fs = ls.flag[s];
        if ((fs & need) != need) continue;
        if ((fs & TF_MARK) != 0) continue;
        uint32_t sp = 0;
        ls.stack[sp++] = (uint16_t)s;
        while (sp) {
            uint32_t c = ls.stack[sp - 1];
            uint8_t fc = ls.flag[c];
            bool valid = true;
            if ((fc & TF_MARK) != 2) {
                for (uint32_t e = ls.in_first[c]; e != VC_NONE16; ) {
                    uint32_t tn = ls.etn[e];
                    uint32_t t = tn & 0xFFFF;
                    e = tn >> 16;
                    uint8_t ft = ls.flag[t];
                    if ((ft & need) != need) continue;
                    if ((ft & TF_MARK) != 2) {
                        if (sp >= STK) { err = VC_WIN_OVERFLOW; break; }
                        ls.stack[sp++] = (uint16_t)t;
                        valid = false;
                    }
                }
                if (err) break;
                uint32_t cnt = ls.alc[c];
                if (!(fc & TF_IGN)) {
                    for (uint32_t k = 0; k < cnt; ++k) {
                        uint32_t a = ls.al[c * ls.ma + k];
                        uint8_t fa = ls.flag[a];
                        if ((fa & need) != need) continue;
                        if ((fa & TF_MARK) != 2) {
                            if (sp >= STK) { err = VC_WIN_OVERFLOW; break; }
                            ls.stack[sp++] = (uint16_t)a;
                            ls.flag[a] = fa | TF_IGN;
                            valid = false;
                        }
                    }
                    if (err) break;
                }
                fc = ls.flag[c];
                if (valid) {
                    ls.flag[c] = (fc & ~TF_MARK) | 2;
                    if (!(fc & TF_IGN)) {
                        ls.rank[nr++] = (uint16_t)c;
                        for (uint32_t k = 0; k < cnt; ++k) {
                            uint32_t a = ls.al[c * ls.ma + k];
                            if ((ls.flag[a] & need) != need) continue;
                            ls.rank[nr++] = (uint16_t)a;
                        }
                    }
                } else {
                    if ((fc & TF_MARK) == 1) { err = VC_WIN_INVALID; break; }   // not a DAG
                    ls.flag[c] = (fc & ~TF_MARK) | 1;
                }
            }
            if (valid) sp--;
        }
    }
    *nrows_out = nr;
    return err;
}

VC_KL __global__ __launch_bounds__(64) void k_topo(VcBatchDev b, VcGraph g, VcDp dp, uint32_t w0, uint32_t nslots,
                                             uint32_t NC, uint32_t EC, uint32_t STK, int next_layer, int only_masked, uint32_t ring,
                                             uint32_t NCl, uint32_t ECl, uint8_t* ws, uint32_t ws_stride, int ws_only) {
    // ws != nullptr: the graph image does not fit the 160 KB LDS; work from this workgroup's HBM workspace
    // instead (same layout, same code; slower, but the window is computed rather than rejected).
    // NC/EC: strides of the graph arrays in HBM; NCl/ECl: capacity the LDS image was sized for (the host
    // passes the known maximum after a prune round, so small graphs do not reserve 60 KB per window)
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t slot = blockIdx.x;
    if (slot >= nslots) return;
    uint32_t w = w0 + slot;
    if (b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    uint32_t s0 = b.win_seq_off[w], ns = b.win_seq_off[w + 1] - s0;
    bool masked = false;
    uint32_t mb = 0, me = 0;
    if (next_layer >= 0) {
        if ((uint32_t)next_layer >= ns) return;
        uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - b.seq_off[s0]);
        mb = b.seq_begin[s0 + next_layer]; me = b.seq_end[s0 + next_layer];
        masked = !vc_full_span(mb, me, L);
        if (only_masked && !masked) return;      // full-span layers take the incremental order (k_rows)
    }
    const uint32_t N = g.n_nodes[slot], E = g.n_edges[slot];
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;

    // The LDS image is sized optimistically (NCl / ECl: what the host expects of a pruned graph); a graph that outgrows it
    // works from this workgroup's HBM workspace instead, laid out for the full capacities -- decided per window, so that
    // the common small graph does not reserve (and block for others) the LDS of the rare large one.
    const bool big = N > NCl || E > ECl;
    if (big && (!ws || N > NC || E > EC)) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 24, N); return; }
    if (big) { NCl = NC; ECl = EC; }
    const VcTopoLds t = vc_topo_carve((big || ws_only) ? ws + (size_t)blockIdx.x * ws_stride : smem, NCl, ECl, STK, g.ma);
    uint16_t* s_in_first = t.in_first; uint32_t* s_etn = t.etn; uint16_t* s_al = t.al;
    uint8_t* s_flag = t.flag; uint16_t* s_rank = t.rank;
    vc_topo_load(g, nb, eb, N, E, t, lane);
    __syncthreads();

    __shared__ uint32_t s_nrows;
    __shared__ int s_err;
    if (lane == 0) {
        uint32_t nr = 0;
        s_err = vc_topo_dfs(t, N, STK, masked, mb, me, &nr);
        s_nrows = nr;
    }
    __syncthreads();
    if (s_err) { if (lane == 0) vc_fail(b, w, s_err, 2, s_nrows); return; }
    const uint32_t nrows = s_nrows;

    // ---- row records (wave-parallel).  s_al is dead now: alias node->rank and two byte maps into it.
    uint16_t* s_noderank = s_al;                       // [NC]
    uint8_t*  s_hasout = (uint8_t*)(s_al + NCl);       // [NC] by node
    for (uint32_t i = lane; i < N; i += 64) s_hasout[i] = 0;
    __syncthreads();
    for (uint32_t r = lane; r < nrows; r += 64) {
        uint32_t v = s_rank[r];
        s_noderank[v] = (uint16_t)r;
        dp.rank2node[nb + r] = (uint16_t)v;
    }
    __syncthreads();
    const uint8_t need = masked ? TF_SUB : 0;
    // pass 1: out-degree
    for (uint32_t r = lane; r < nrows; r += 64) {
        uint32_t v = s_rank[r];
        for (uint32_t e = s_in_first[v]; e != VC_NONE16; ) {
            uint32_t tn = s_etn[e];
            uint32_t t = tn & 0xFFFF;
            e = tn >> 16;
            if ((s_flag[t] & need) != need) continue;
            s_hasout[t] = 1;
        }
    }
    __syncthreads();
    // pass 2: records, overflow lists
    uint32_t ovf_base = 0;
    int bad = 0;
    for (uint32_t r0 = 0; r0 < nrows; r0 += 64) {
        uint32_t r = r0 + lane;
        bool act = r < nrows;
        uint32_t v = act ? s_rank[r] : 0;
        uint32_t np = 0;
        bool hasprev = false;
        uint16_t dl[VC_INLINE_PRED];
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) dl[k] = 0;
        if (act) {
            for (uint32_t e = s_in_first[v]; e != VC_NONE16; ) {
                uint32_t tn = s_etn[e];
                uint32_t t = tn & 0xFFFF;
                e = tn >> 16;
                if ((s_flag[t] & need) != need) continue;
                uint32_t delta = r - s_noderank[t];
#pragma unroll
                for (int k = 0; k < VC_INLINE_PRED; ++k) if (np == (uint32_t)k) dl[k] = (uint16_t)delta;
                np++;
                hasprev |= delta == 1;
            }
        }
        bool is_ovf = np > VC_INLINE_PRED;
        uint32_t tot_ovf;
        uint32_t my_ovf = wave_excl_sum(is_ovf ? np : 0u, tot_ovf) + ovf_base;
        if (act) {
            if (np == 0) { np = 1; dl[0] = (uint16_t)(r + 1); }   // virtual row 0 is `row` rows above
            if (is_ovf) {
                if (my_ovf + np > EC) bad = 1;
                else {
                    uint32_t k = 0;
                    for (uint32_t e = s_in_first[v]; e != VC_NONE16; ) {
                        uint32_t tn = s_etn[e];
                        uint32_t t = tn & 0xFFFF;
                        e = tn >> 16;
                        if ((s_flag[t] & need) != need) continue;
                        uint32_t delta = r - s_noderank[t];
                        dp.ovf[eb + my_ovf + k] = (uint16_t)delta;
                        k++;
                    }
                }
                dl[0] = (uint16_t)(my_ovf & 0xFFFF); dl[1] = (uint16_t)(my_ovf >> 16);
                dl[2] = (uint16_t)(np & 0xFFFF); dl[3] = (uint16_t)(np >> 16);      // the full count (the header field saturates at 255)
            }
            uint32_t fl = (s_hasout[v] ? 0u : VC_RF_SINK) | (is_ovf ? VC_RF_OVF : 0u) | (hasprev ? VC_RF_PREV : 0u);
            uint4 rec;
            rec.x = (uint32_t)g.code[nb + v] | (fl << 8) | (min(np, 255u) << 16);
            rec.y = dl[0] | ((uint32_t)dl[1] << 16);
            rec.z = dl[2] | ((uint32_t)dl[3] << 16);
            rec.w = dl[4] | ((uint32_t)dl[5] << 16);
            dp.rec[nb + r] = rec;
            dp.fie[(uint64_t)slot * VC_FIE_STRIDE(NC) + r + 1] = vc_fie_entry(rec.x, rec.y);
            if (r == 0) dp.fie[(uint64_t)slot * VC_FIE_STRIDE(NC)] = 0;
            dp.frec[nb + r] = vc_make_frec(rec.x & 0xFF, fl, np, dl, is_ovf, hasprev, r, ring);
        }
        ovf_base += tot_ovf;
    }
    bad = __any(bad);
    // banded matrix store (vc_band_start): rows that a later row reads back from the stored matrix are written whole --
    // the predecessors beyond the LDS ring and everything on a long list (the forward pass reads those as the records say)
    __syncthreads();
    for (uint32_t r = lane; r < nrows && !bad; r += 64) {
        const uint4 q = dp.frec[nb + r];
        const uint32_t fl = (q.x >> 8) & 0xFF;
        if (!(fl & VC_RF_SLOW)) continue;
        if (fl & VC_RF_OVF) {
            const uint32_t o0 = q.y, cnt = q.z;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t d = dp.ovf[eb + o0 + k];
                if (d > ring && d <= r) atomicOr(reinterpret_cast<uint32_t*>(&dp.frec[nb + r - d]), VC_RF_FULL << 8);
            }
        } else {
            const uint32_t nq = (q.x >> 16) & 0xFF;
            const uint32_t dl[VC_INLINE_PRED] = {q.y & 0xFFFF, q.y >> 16, q.z & 0xFFFF, q.z >> 16, q.w & 0xFFFF, q.w >> 16};
#pragma unroll
            for (int k = 0; k < VC_INLINE_PRED; ++k)
                if ((uint32_t)k < nq && dl[k] > ring && dl[k] <= r) atomicOr(reinterpret_cast<uint32_t*>(&dp.frec[nb + r - dl[k]]), VC_RF_FULL << 8);
        }
    }
    if (lane == 0) {
        dp.nrows[slot] = nrows;
        dp.flags[slot] = bad ? 1u : 0u;
        if (bad) b.errinfo[w] = (11u << 16) | (ovf_base & 0xFFFF);
    }
}

// ------------------------------------------------------------------------------------------------
// vc_rows_full (was k_rows): row records for a FULL-SPAN next layer from the incrementally maintained order VcGraph::ord
// (no DFS).  H and the backtrack do not depend on which topological order the rows are visited in
// (sisd :315-360 only needs predecessors first; ties in the backtrack follow in-edge LIST order),
// so any valid order gives the reference's matrix.  The one place the reference's rank matters --
// "first sink in rank order" among equal end scores (sisd :353-355) -- is settled by k_resolve,
// which runs the exact DFS only for the ~1-2 % of alignments that actually tie.
// ------------------------------------------------------------------------------------------------
// Row records of a full-span layer.  The rows of such an alignment are the nodes in the kept order VcGraph::ord, so the record
// of row r follows from the node's own record (code, in-degree, tails of the first in-edges in list order -- VcGraph::nrec,
// kept by k_init / k_addaln), the tails' positions and the node's out-list head: three dependent loads per row instead of a
// walk along the in-edge chain.  Writes the backtrack's view (VcDp::rec), the forward view (VcDp::frec) and rank2node.  A
// list beyond VC_INLINE_PRED entries is walked through the in-edge chain into VcDp::ovf (rare).
struct VcOtf {
    const uint16_t* ord; const uint16_t* pos; const uint4* nrec; const uint16_t* out_first; const uint16_t* in_first; const uint32_t* e_tn;
    uint4* rec; uint4* frec; uint16_t* rank2node; uint16_t* ovf; uint8_t* fie;
    uint32_t N, EC;
};
template <int U>
__device__ __forceinline__ void vc_otf_rows(const VcOtf& o, uint32_t r0, uint32_t ring, uint32_t& ovf_base, int& bad, bool plain_frec,
                                            uint32_t* s_keep = nullptr) {
    // s_keep != nullptr (kept-row ring): mark, bit per row in LDS, the rows some row reads back at distance 2..64 -- the first pass
    // of vc_frec_kept, done here where the distances are in registers anyway (that pass was a global load per row on a chain)
    // U blocks of 64 rows at once, level by level: every load of a level is issued before the first result is used, so a lane
    // has U (then 6 U) independent loads in flight instead of one chain
    const uint32_t lane = (uint32_t)vc_lane();
    uint32_t r[U], v[U], of[U], np[U];
    bool act[U];
    uint4 nr[U];
    uint32_t pp[U][VC_INLINE_PRED];
    // The lanes walk the NODES (ids r0 ...), not the rows: a node's row is pos[node], which comes in the same round trip as the
    // node's record and out-list head (all three indexed by the id: coalesced), where "row -> ord[row] -> the node's record" was a
    // level more.  The records land where they belong (rec[pos[node]]); nothing below depends on the order the rows are made in.
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = r0 + 64 * u + lane; act[u] = v[u] < o.N; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        nr[u] = make_uint4(0, 0, 0, 0); of[u] = 0; r[u] = 0;
        if (act[u]) { r[u] = o.pos[v[u]]; nr[u] = o.nrec[v[u]]; of[u] = o.out_first[v[u]]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        np[u] = nr[u].x >> 16;
        const uint32_t pid[VC_INLINE_PRED] = {nr[u].y & 0xFFFF, nr[u].y >> 16, nr[u].z & 0xFFFF, nr[u].z >> 16, nr[u].w & 0xFFFF, nr[u].w >> 16};
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) pp[u][k] = (act[u] && (uint32_t)k < np[u]) ? (uint32_t)o.pos[pid[k]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t code = nr[u].x & 0xFF;
        uint32_t npu = np[u];
        uint16_t dl[VC_INLINE_PRED];
        bool hasprev = false;
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) {
            dl[k] = 0;
            if (act[u] && (uint32_t)k < npu) {
                if (pp[u][k] >= r[u]) bad |= 2;                   // the kept order must stay topological
                const uint32_t delta = r[u] - pp[u][k];
                dl[k] = (uint16_t)delta;
                hasprev |= delta == 1;
                if (s_keep && npu <= VC_INLINE_PRED && delta >= 2 && delta <= 64 && delta <= r[u]) atomicOr(&s_keep[(r[u] - delta) >> 5], 1u << ((r[u] - delta) & 31));
            }
        }
        const bool is_ovf = npu > VC_INLINE_PRED;
        if (__any(is_ovf)) {
            uint32_t tot_ovf;
            const uint32_t my_ovf = wave_excl_sum(is_ovf ? npu : 0u, tot_ovf) + ovf_base;
            if (is_ovf) {
                if (my_ovf + npu > o.EC) bad |= 1;
                else {
                    uint32_t k = 0;
                    for (uint32_t e = o.in_first[v[u]]; e != VC_NONE16; ) {
                        const uint32_t tn = o.e_tn[e];
                        e = tn >> 16;
                        const uint32_t delta = r[u] - o.pos[tn & 0xFFFF];
                        o.ovf[my_ovf + k] = (uint16_t)delta;
                        hasprev |= delta == 1;
                        if (s_keep && delta >= 2 && delta <= 64 && delta <= r[u]) atomicOr(&s_keep[(r[u] - delta) >> 5], 1u << ((r[u] - delta) & 31));
                        k++;
                    }
                }
                dl[0] = (uint16_t)(my_ovf & 0xFFFF); dl[1] = (uint16_t)(my_ovf >> 16);
                dl[2] = (uint16_t)(npu & 0xFFFF); dl[3] = (uint16_t)(npu >> 16);
            }
            ovf_base += tot_ovf;
        }
        if (act[u]) {
            if (npu == 0) { npu = 1; dl[0] = (uint16_t)(r[u] + 1); }       // the virtual row 0 is `row` rows above
            const uint32_t fl = (of[u] == VC_NONE16 ? VC_RF_SINK : 0u) | (is_ovf ? VC_RF_OVF : 0u) | (hasprev ? VC_RF_PREV : 0u);
            uint4 rec;
            rec.x = code | (fl << 8) | (min(npu, 255u) << 16);
            rec.y = dl[0] | ((uint32_t)dl[1] << 16);
            rec.z = dl[2] | ((uint32_t)dl[3] << 16);
            rec.w = dl[4] | ((uint32_t)dl[5] << 16);
            o.rec[r[u]] = rec;
            o.fie[r[u] + 1] = vc_fie_entry(rec.x, rec.y);
            if (r[u] == 0) o.fie[0] = 0;
            o.rank2node[r[u]] = (uint16_t)v[u];
            if (plain_frec) o.frec[r[u]] = vc_make_frec(code, fl, npu, dl, is_ovf, hasprev, r[u], ring);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Kept-row ring (build phase).  In a grown graph a row has ~1.5 predecessors that are not the row above, up to ~12 rows
// back, but only ~70 % of the rows are ever read back that way.  So k_fwd keeps in LDS only the rows some later row will ask
// for, in a ring of K SLOTS: K = 6 slots hold what a plain ring needs 8 rows for (96 % of such reads at the last layers),
// and the smaller ring is what lets a fifth / sixth wave onto each SIMD.  The forward records are made here, after the
// backtrack's records of all rows exist: mark the rows that are read back (distance 2..64, inline lists), count them
// (kix[r] = kept rows before row r), and give every such predecessor its slot kix[p] % K -- it is still there when the
// reader comes iff kix[reader] - kix[p] <= K.  A forward entry (u16) is then 0x8000 | slot << 8 | distance for a row in the
// ring, or the plain distance (< 32768) for one that has to come back from the stored matrix; the word with the flags
// carries VC_RF_KEEP and the row's own slot in bits 27..29.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t vc_kept_lds_bytes(uint32_t NC) { return 8u * (NC / 32 + 2) + 2u * (NC + 2) + 16u; }

// the two bitmaps of vc_frec_kept (keep, full) at the head of its LDS scratch: cleared by whoever marks the keep bits
__device__ __forceinline__ uint32_t* vc_frec_kept_clear(uint32_t nrows, uint8_t* lds) {
    uint32_t* s_keep = reinterpret_cast<uint32_t*>(lds);
    for (uint32_t i = (uint32_t)vc_lane(); i < 2 * (nrows / 32 + 2); i += 64) s_keep[i] = 0;
    __syncthreads();
    return s_keep;
}
// marked: the keep bits stand already (vc_otf_rows set them while it made the records)
__device__ __forceinline__ void vc_frec_kept(const uint4* rec, uint4* frec, uint16_t* ovf, uint32_t nrows, uint32_t K, uint8_t* lds, bool marked = false) {
    const uint32_t lane = (uint32_t)vc_lane();
    uint32_t* s_keep = reinterpret_cast<uint32_t*>(lds);                           // bit per row
    uint32_t* s_full = s_keep + (nrows / 32 + 2);                                   // bit per row: read back from the stored matrix (VC_RF_FULL)
    uint16_t* s_kix = reinterpret_cast<uint16_t*>(s_full + (nrows / 32 + 2));       // [nrows + 1] kept rows before row r
    if (!marked) (void)vc_frec_kept_clear(nrows, lds);
    // which rows are read back
    for (uint32_t r = lane; r < nrows && !marked; r += 64) {
        const uint4 q = rec[r];
        const uint32_t fl = (q.x >> 8) & 0xFF, np = (q.x >> 16) & 0xFF;
        if (fl & VC_RF_OVF) {                                                      // a long list (VcDp::ovf): its rows are read back all the same
            const uint32_t o0 = q.y, cnt = q.z;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t d = ovf[o0 + k];
                if (d >= 2 && d <= 64 && d <= r) atomicOr(&s_keep[(r - d) >> 5], 1u << ((r - d) & 31));
            }
            continue;
        }
        const uint32_t dl[VC_INLINE_PRED] = {q.y & 0xFFFF, q.y >> 16, q.z & 0xFFFF, q.z >> 16, q.w & 0xFFFF, q.w >> 16};
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k)
            if ((uint32_t)k < np && dl[k] >= 2 && dl[k] <= 64 && dl[k] <= r) atomicOr(&s_keep[(r - dl[k]) >> 5], 1u << ((r - dl[k]) & 31));
    }
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t r0 = 0; r0 < nrows; r0 += 64) {
        const uint32_t r = r0 + lane;
        const bool kp = r < nrows && ((s_keep[r >> 5] >> (r & 31)) & 1u);
        const unsigned long long m = __ballot(kp);
        if (r <= nrows) s_kix[r] = (uint16_t)(base + __popcll(m & ((1ull << lane) - 1ull)));
        base += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_kix[nrows] = (uint16_t)base;
    __syncthreads();
    // (four rows per lane per trip: the records of a trip are requested together -- one load per row on a chain was 35 round trips)
    for (uint32_t rb = lane; rb < nrows; rb += 256) {
      uint4 qq[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) qq[u] = rb + 64 * u < nrows ? rec[rb + 64 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t r = rb + 64 * u;
        if (r >= nrows) continue;
        const uint4 q = qq[u];
        const uint32_t code = q.x & 0xFF, fl = (q.x >> 8) & 0xFF, np = (q.x >> 16) & 0xFF;
        const bool is_ovf = (fl & VC_RF_OVF) != 0, hasprev = (fl & VC_RF_PREV) != 0;
        const uint32_t dl[VC_INLINE_PRED] = {q.y & 0xFFFF, q.y >> 16, q.z & 0xFFFF, q.z >> 16, q.w & 0xFFFF, q.w >> 16};
        uint32_t f = fl & (VC_RF_SINK | VC_RF_OVF | VC_RF_PREV);
        uint32_t out[VC_INLINE_PRED];
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) out[k] = 0;
        uint32_t nq = 0;
        bool slow = is_ovf;
        const uint32_t kr = s_kix[r];
        if (is_ovf) {
            out[0] = dl[0]; out[1] = dl[1]; out[2] = dl[2]; out[3] = dl[3];           // offset into VcDp::ovf, number of entries there
            nq = min(np, 255u);
            // the list itself takes the forward form (the backtrack reads distances from it: it masks the mark off)
            const uint32_t o0 = q.y, cnt = q.z;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t d = ovf[o0 + k] & 0x7FFFu;
                bool hit = false;
                if (d >= 2 && d <= 64 && d <= r) {
                    const uint32_t kp = s_kix[r - d];
                    if (kr - kp <= K) { hit = true; ovf[o0 + k] = (uint16_t)(0x8000u | ((kp % K) << 8) | d); }
                }
                if (!hit && d >= 2 && d <= r) atomicOr(&s_full[(r - d) >> 5], 1u << ((r - d) & 31));
            }
        } else {
#pragma unroll
            for (int k = 0; k < VC_INLINE_PRED; ++k) {
                if ((uint32_t)k < np && !(hasprev && dl[k] == 1)) {
                    const uint32_t d = dl[k];
                    uint32_t e = d;
                    bool hit = false;
                    if (d >= 2 && d <= 64 && d <= r) {
                        const uint32_t kp = s_kix[r - d];
                        if (kr - kp <= K) { hit = true; e = 0x8000u | ((kp % K) << 8) | d; }
                    }
                    if (!hit) {
                        slow = true;                                                 // (the host uses this form below 32768 rows only: d < 0x8000)
                        if (d <= r) atomicOr(&s_full[(r - d) >> 5], 1u << ((r - d) & 31));     // that row comes back from the stored matrix
                    }
#pragma unroll
                    for (int t = 0; t < VC_INLINE_PRED; ++t) if (nq == (uint32_t)t) out[t] = e;
                    nq++;
                }
            }
        }
        if (slow) f |= VC_RF_SLOW;
        if (hasprev && nq == 0 && !slow) f |= VC_RF_PLAIN;
        const bool keep = (s_keep[r >> 5] >> (r & 31)) & 1u;
        if (keep) f |= VC_RF_KEEP;
        const uint32_t bi = code == 'A' ? 0u : code == 'C' ? 1u : code == 'G' ? 2u : code == 'T' ? 3u : 4u;
        uint4 o;
        o.x = code | (f << 8) | (nq << 16) | (bi << 24) | ((keep ? kr % K : 0u) << 27);
        o.y = out[0] | (out[1] << 16);
        o.z = out[2] | (out[3] << 16);
        o.w = out[4] | (out[5] << 16);
        frec[r] = o;
      }
    }
    __syncthreads();
    for (uint32_t r = lane; r < nrows; r += 64)
        if ((s_full[r >> 5] >> (r & 31)) & 1u) frec[r].x |= VC_RF_FULL << 8;
}

// Runs at the tail of the kernel that last changed the graph (k_init for layer 1, k_addaln of layer j for layer j + 1): the wave
// that has just written the graph prepares the rows of its next alignment, instead of a kernel of its own between k_addaln and
// k_fwd (one launch and one queueing delay less per layer, and the graph arrays are still warm in the cache).
#ifndef VC_ROWS_U
#define VC_ROWS_U 4
#endif
// kept != 0: forward records for k_fwd's kept-row ring of `kept` slots (vc_frec_kept; `lds` = vc_kept_lds_bytes(NC) of scratch)
__device__ __forceinline__ void vc_rows_full(const VcBatchDev& b, const VcGraph& g, const VcDp& dp, uint32_t slot, uint32_t w,
                                             uint32_t NC, uint32_t EC, int next_layer, uint32_t ring, uint32_t N, uint32_t kept, uint8_t* lds) {
    const int lane = vc_lane();
    const uint32_t s0 = b.win_seq_off[w], ns = b.win_seq_off[w + 1] - s0;
    if ((uint32_t)next_layer >= ns) return;
    {
        const uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - b.seq_off[s0]);
        if (!vc_full_span(b.seq_begin[s0 + next_layer], b.seq_end[s0 + next_layer], L)) return;   // k_rows_sub's job
    }
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    VcOtf ot;
    ot.ord = g.ord + nb; ot.pos = g.pos + nb; ot.nrec = g.nrec + nb; ot.out_first = g.out_first + nb; ot.in_first = g.in_first + nb;
    ot.e_tn = g.e_tn + eb; ot.rec = dp.rec + nb; ot.frec = dp.frec + nb; ot.rank2node = dp.rank2node + nb; ot.ovf = dp.ovf + eb; ot.fie = dp.fie + (uint64_t)slot * VC_FIE_STRIDE(NC);
    ot.N = N; ot.EC = EC;
    uint32_t ovf_base = 0;
    int bad = 0;
    // VC_ROWS_U blocks of 64 rows per iteration: more independent load chains in flight per lane
    uint32_t* const s_keep = kept ? vc_frec_kept_clear(N, lds) : nullptr;
    for (uint32_t r0 = 0; r0 < N; r0 += 64 * VC_ROWS_U) vc_otf_rows<VC_ROWS_U>(ot, r0, ring, ovf_base, bad, kept == 0, s_keep);
    if (kept) { __syncthreads(); vc_frec_kept(ot.rec, ot.frec, ot.ovf, N, kept, lds, true); }
    bad = __any(bad & 1) | (__any(bad & 2) ? 2 : 0);
    if (bad & 2) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 12, 0); return; }   // order invariant violated
    if (lane == 0) {
        dp.nrows[slot] = N;
        dp.flags[slot] = ((bad & 1) ? 1u : 0u) | 2u;
        if (bad & 1) b.errinfo[w] = (13u << 16) | (ovf_base & 0xFFFF);
    }
}

// ------------------------------------------------------------------------------------------------
// k_rows_sub: row records for a PARTIAL-SPAN next layer without the DFS.  The reference aligns such a
// layer to Graph::Subgraph(begin, end) (graph.cpp:640-732): everything that reaches nodes_[end]
// backwards over in-edges and aligned links without passing a node id < begin.  VcGraph::ord restricted
// to that set is a valid DP order of the subgraph (a restriction of a topological order, aligned groups
// still contiguous), so only the membership is needed: a reverse sweep over ord in 64-position blocks,
// every lane pulling from its out-neighbours / aligned mates, each block iterated to its fixed point
// with ballots; repeated until nothing changes (groups straddling a block boundary).
// LDS carve: member bits 8*nW | rowidx 2*NC | node-id bitmap 4*(NC/32+1)
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t vc_rows_sub_lds_bytes(uint32_t NC, uint32_t kept) {
    const uint32_t a = 8 * ((NC + 63) / 64) + 2 * NC + ((NC + 15) & ~15u) + 4 * (NC / 32 + 1) + 64, b = kept ? vc_kept_lds_bytes(NC) : 0u;
    return a > b ? a : b;
}
__device__ __forceinline__ void vc_rows_sub_body(const VcBatchDev& b, const VcGraph& g, const VcDp& dp, uint32_t w0, uint32_t nslots,
                                                 uint32_t NC, uint32_t EC, int next_layer, uint32_t ring, uint32_t* submask, uint32_t kept,
                                                 uint8_t* smem, const uint32_t slot) {
    if (slot >= nslots) return;
    const uint32_t w = w0 + slot;
    if (b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint32_t s0 = b.win_seq_off[w], ns = b.win_seq_off[w + 1] - s0;
    if ((uint32_t)next_layer >= ns) return;
    const uint32_t L = (uint32_t)(b.seq_off[s0 + 1] - b.seq_off[s0]);
    const uint32_t mb = b.seq_begin[s0 + next_layer], me = b.seq_end[s0 + next_layer];
    if (vc_full_span(mb, me, L)) return;                       // k_rows' job
    const uint32_t N = g.n_nodes[slot];
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    if (me >= N || mb >= N) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 19, me); return; }
    const uint32_t nW = (NC + 63) / 64;
    unsigned long long* s_mem = (unsigned long long*)smem;                 // [nW] member bit per POSITION
    uint16_t* s_rowidx = (uint16_t*)(s_mem + nW);                          // [NC] row of a member position
    uint32_t* s_sub = (uint32_t*)(s_rowidx + ((NC + 1) & ~1u));            // [NC/32+1] member bit per NODE id
    for (uint32_t i = lane; i < nW; i += 64) s_mem[i] = 0;
    for (uint32_t i = lane; i < NC / 32 + 1; i += 64) s_sub[i] = 0;
    __syncthreads();
    auto is_mem = [&](uint32_t p) -> bool { return (s_mem[p >> 6] >> (p & 63)) & 1ull; };
    // the sweep starts at the block holding the last position of end's aligned group
    uint32_t ptop = g.pos[nb + me];
    {
        const uint32_t cnt = g.al_cnt[nb + me];
        for (uint32_t t = 0; t < cnt; ++t) ptop = max(ptop, (uint32_t)g.pos[nb + g.al[(nb + me) * g.ma + t]]);
    }
    int guard = 0;
    for (;;) {
        int changed = 0;
        for (int B = (int)(ptop >> 6); B >= 0; --B) {
            const uint32_t p = (uint32_t)B * 64 + lane;
            const bool act = p < N && p <= ptop;
            const uint32_t v = act ? g.ord[nb + p] : 0;
            const bool elig = act && v >= mb;
            unsigned long long pull = 0;
            bool ext = act && v == me;
            if (elig) {
                for (uint32_t e = g.out_first[nb + v]; e != VC_NONE16; ) {
                    const uint32_t hn = g.e_hn[eb + e];
                    e = hn >> 16;
                    const uint32_t ph = g.pos[nb + (hn & 0xFFFF)];
                    if ((ph >> 6) == (uint32_t)B) pull |= 1ull << (ph & 63);
                    else if (ph <= ptop && is_mem(ph)) ext = true;
                }
                const uint32_t cnt = g.al_cnt[nb + v];
                for (uint32_t t = 0; t < cnt; ++t) {
                    const uint32_t pa = g.pos[nb + g.al[(nb + v) * g.ma + t]];
                    if ((pa >> 6) == (uint32_t)B) pull |= 1ull << (pa & 63);
                    else if (pa <= ptop && is_mem(pa)) ext = true;
                }
            }
            const unsigned long long cur = s_mem[B];
            unsigned long long M = cur;
            for (;;) {
                const unsigned long long M2 = __ballot(elig && (ext || (pull & M) != 0ull)) | M;
                if (M2 == M) break;
                M = M2;
            }
            if (M != cur) { if (lane == 0) s_mem[B] = M; changed = 1; }
            __syncthreads();
        }
        if (!changed) break;
        if (++guard > 64) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 20, 0); return; }
    }
    // compact: row index of every member position
    uint32_t nrows = 0;
    for (uint32_t B = 0; B <= (ptop >> 6); ++B) {
        const unsigned long long M = s_mem[B];
        const uint32_t p = B * 64 + lane;
        if ((M >> lane) & 1ull) {
            s_rowidx[p] = (uint16_t)(nrows + __popcll(M & ((1ull << lane) - 1ull)));
            const uint32_t v = g.ord[nb + p];
            atomicOr(&s_sub[v >> 5], 1u << (v & 31));
        }
        nrows += __popcll(M);
    }
    __syncthreads();
    if (nrows == 0) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 21, 0); return; }
    int bad = 0, broken = 0;
    // records, block by block in position order
    uint32_t ovf_base = 0;
    for (uint32_t B = 0; B <= (ptop >> 6); ++B) {
        const uint32_t p = B * 64 + lane;
        const bool act = p < N && p <= ptop && is_mem(p);
        const uint32_t v = act ? g.ord[nb + p] : 0, r = act ? s_rowidx[p] : 0;
        uint32_t np = 0;
        bool hasprev = false;
        uint16_t dl[VC_INLINE_PRED];
#pragma unroll
        for (int k = 0; k < VC_INLINE_PRED; ++k) dl[k] = 0;
        if (act) {
            for (uint32_t e = g.in_first[nb + v]; e != VC_NONE16; ) {
                const uint32_t tn = g.e_tn[eb + e];
                e = tn >> 16;
                const uint32_t pt = g.pos[nb + (tn & 0xFFFF)];
                if (pt > ptop || !is_mem(pt)) continue;
                if (s_rowidx[pt] >= r) broken = 1;               // the kept order must stay topological
                const uint32_t delta = r - s_rowidx[pt];
#pragma unroll
                for (int k = 0; k < VC_INLINE_PRED; ++k) if (np == (uint32_t)k) dl[k] = (uint16_t)delta;
                np++;
                hasprev |= delta == 1;
            }
        }
        const bool is_ovf = np > VC_INLINE_PRED;
        uint32_t tot_ovf;
        const uint32_t my_ovf = wave_excl_sum(is_ovf ? np : 0u, tot_ovf) + ovf_base;
        if (act) {
            if (np == 0) { np = 1; dl[0] = (uint16_t)(r + 1); }
            if (is_ovf) {
                if (my_ovf + np > EC) bad = 1;
                else {
                    uint32_t k = 0;
                    for (uint32_t e = g.in_first[nb + v]; e != VC_NONE16; ) {
                        const uint32_t tn = g.e_tn[eb + e];
                        e = tn >> 16;
                        const uint32_t pt = g.pos[nb + (tn & 0xFFFF)];
                        if (pt > ptop || !is_mem(pt)) continue;
                        dp.ovf[eb + my_ovf + k] = (uint16_t)(r - s_rowidx[pt]);
                        k++;
                    }
                }
                dl[0] = (uint16_t)(my_ovf & 0xFFFF); dl[1] = (uint16_t)(my_ovf >> 16);
                dl[2] = (uint16_t)(np & 0xFFFF); dl[3] = (uint16_t)(np >> 16);      // the full count (the header field saturates at 255)
            }
            bool hasout = false;                                // a sink of the subgraph has no member successor
            for (uint32_t e = g.out_first[nb + v]; e != VC_NONE16 && !hasout; ) {
                const uint32_t hn = g.e_hn[eb + e];
                e = hn >> 16;
                const uint32_t ph = g.pos[nb + (hn & 0xFFFF)];
                hasout = ph <= ptop && is_mem(ph);
            }
            const uint32_t fl = (hasout ? 0u : VC_RF_SINK) | (is_ovf ? VC_RF_OVF : 0u) | (hasprev ? VC_RF_PREV : 0u);
            uint4 rec;
            rec.x = (uint32_t)g.code[nb + v] | (fl << 8) | (min(np, 255u) << 16);
            rec.y = dl[0] | ((uint32_t)dl[1] << 16);
            rec.z = dl[2] | ((uint32_t)dl[3] << 16);
            rec.w = dl[4] | ((uint32_t)dl[5] << 16);
            dp.rec[nb + r] = rec;
            dp.fie[(uint64_t)slot * VC_FIE_STRIDE(NC) + r + 1] = vc_fie_entry(rec.x, rec.y);
            if (r == 0) dp.fie[(uint64_t)slot * VC_FIE_STRIDE(NC)] = 0;
            dp.frec[nb + r] = vc_make_frec(rec.x & 0xFF, fl, np, dl, is_ovf, hasprev, r, ring);
            dp.rank2node[nb + r] = (uint16_t)v;
        }
        ovf_base += tot_ovf;
    }
    for (uint32_t i = lane; i < NC / 32 + 1; i += 64) submask[(uint64_t)slot * (NC / 32 + 1) + i] = s_sub[i];
    bad = __any(bad);
    broken = __any(broken);
    if (broken) { if (lane == 0) vc_fail(b, w, VC_WIN_INVALID, 22, 0); return; }
    if (kept) {                                               // forward records for the kept-row ring (the LDS of the sweep is free now)
        __syncthreads();
        vc_frec_kept(dp.rec + nb, dp.frec + nb, dp.ovf + eb, nrows, kept, smem);
    }
    if (lane == 0) {
        dp.nrows[slot] = nrows;
        dp.flags[slot] = (bad ? 1u : 0u) | 2u | 4u;            // incremental order, masked
        if (bad) b.errinfo[w] = (23u << 16);
    }
}

VC_KL __global__ __launch_bounds__(64) void k_rows_sub(VcBatchDev b, VcGraph g, VcDp dp, uint32_t w0, uint32_t nslots,
                                                 uint32_t NC, uint32_t EC, int next_layer, uint32_t ring, uint32_t* submask, uint32_t kept,
                                                 const uint32_t* cursor) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (cursor) {                                             // every window at its own layer; one that repeats its layer has its rows already
        if (blockIdx.x >= nslots) return;
        const uint32_t cv = cursor[blockIdx.x];
        if (cv >> 31) return;
        next_layer = (int)(cv & 0xFFFFu);
    }
    vc_rows_sub_body(b, g, dp, w0, nslots, NC, EC, next_layer, ring, submask, kept, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_resolve: the reference picks, among sinks with the same best end score, the first in ITS rank
// order (sisd :353-355 with `<`).  For alignments done on the incremental order and ending in such a
// tie, run the exact TopologicalSort DFS now and choose the tied sink with the smallest rank.
// ------------------------------------------------------------------------------------------------
__device__ void vc_resolve_one(uint8_t* smem, uint8_t* gws, uint32_t slot, const VcBatchDev& b, const VcGraph& g, const VcDp& dp,
                               uint32_t w0, uint32_t nslots, uint32_t NC, uint32_t EC, uint32_t STK,
                               const uint16_t* tie_rows, const uint32_t* tie_cnt, const uint32_t* tie_over, uint32_t tie_over_stride, uint32_t* job_end,
                               const uint32_t* submask, int layer, int force_dfs) {
    if (slot >= nslots) return;
    const uint32_t w = w0 + slot;
    const uint32_t nt = tie_cnt[slot];
    if (nt < 2) return;
    if (b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint32_t N = g.n_nodes[slot], E = g.n_edges[slot];
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    // alignment to a Subgraph: every rule below applies to the induced subgraph (its own DFS, roots in id order)
    const bool masked = (dp.flags[slot] & 4u) != 0;
    const uint32_t* sub = submask + (uint64_t)slot * (NC / 32 + 1);
    auto in_sub = [&](uint32_t v) -> bool { return !masked || ((sub[v >> 5] >> (v & 31)) & 1u); };

    // ---- shortcut that settles most ties without the DFS.  The reference's DFS takes roots in id order
    // and, from a root, reaches everything upstream of it through in-edges and aligned links; an aligned
    // group is emitted as a block [leader, leader's aligned list] the first time any member is reached.
    // Hence group A precedes group B if the smallest id that can reach A (forward closure of A over
    // out-edges and aligned links, F(A)) is smaller than that of B; and when min F(A) is itself a member
    // of A, that member is a root and therefore A's leader.  Anything else falls through to the DFS.
    if (nt <= VC_MAXTIE) {
        uint32_t* s_vis = (uint32_t*)smem;                        // bitmap [N]
        uint16_t* s_stk = (uint16_t*)(smem + 4 * ((NC + 31) / 32 + 1));   // [256]
        __shared__ uint32_t s_fast;                                // winning row, 0 = undecided
        for (uint32_t i = lane; i < (N + 31) / 32; i += 64) s_vis[i] = 0;
        __syncthreads();
        if (lane == 0) {
            uint32_t row[VC_MAXTIE], node[VC_MAXTIE], gid[VC_MAXTIE], rmin[VC_MAXTIE];
            bool lead[VC_MAXTIE];
            bool ok = true;
            for (uint32_t k = 0; k < nt; ++k) {
                row[k] = tie_rows[(uint64_t)slot * VC_MAXTIE + k];
                node[k] = dp.rank2node[nb + row[k] - 1];
                uint32_t gm = node[k];
                const uint32_t cnt = g.al_cnt[nb + node[k]];
                for (uint32_t t2 = 0; t2 < cnt; ++t2) {
                    const uint32_t mnode = g.al[(nb + node[k]) * g.ma + t2];
                    if (in_sub(mnode)) gm = min(gm, mnode);
                }
                gid[k] = gm; rmin[k] = 0xFFFFFFFFu; lead[k] = false;
            }
            for (uint32_t k = 0; k < nt && ok; ++k) {
                bool seen = false;
                for (uint32_t q = 0; q < k; ++q) if (gid[q] == gid[k]) { rmin[k] = rmin[q]; lead[k] = lead[q]; seen = true; break; }
                if (seen) continue;
                // forward closure of the group
                uint32_t sp = 0, visited = 0, mn = 0xFFFFFFFFu;
                auto push = [&](uint32_t v) {
                    if (!in_sub(v)) return;
                    if (s_vis[v >> 5] & (1u << (v & 31))) return;
                    s_vis[v >> 5] |= 1u << (v & 31);
                    if (sp < 256) s_stk[sp++] = (uint16_t)v; else ok = false;
                };
                push(node[k]);
                while (sp && ok) {
                    const uint32_t v = s_stk[--sp];
                    mn = min(mn, v);
                    if (++visited > 512) { ok = false; break; }
                    for (uint32_t e = g.out_first[nb + v]; e != VC_NONE16; ) {
                        const uint32_t hn = g.e_hn[eb + e];
                        push(hn & 0xFFFF);
                        e = hn >> 16;
                    }
                    const uint32_t cnt = g.al_cnt[nb + v];
                    for (uint32_t t2 = 0; t2 < cnt; ++t2) push(g.al[(nb + v) * g.ma + t2]);
                }
                rmin[k] = mn;
                // is the smallest id a member of the group?
                bool member = mn == node[k];
                const uint32_t cnt = g.al_cnt[nb + node[k]];
                for (uint32_t t2 = 0; t2 < cnt; ++t2) member = member || mn == g.al[(nb + node[k]) * g.ma + t2];   // mn is in the subgraph
                lead[k] = member;
                // groups are disjoint and closures of different tied groups must not share the bitmap
                for (uint32_t i2 = 0; i2 < (N + 31) / 32; ++i2) s_vis[i2] = 0;
            }
            uint32_t win = 0;
            if (ok) {
                uint32_t best = 0xFFFFFFFFu, bestg = 0xFFFFFFFFu;
                bool amb = false;
                for (uint32_t k = 0; k < nt; ++k) {
                    if (rmin[k] < best) { best = rmin[k]; bestg = gid[k]; amb = false; }
                    else if (rmin[k] == best && gid[k] != bestg) amb = true;
                }
                if (!amb) {
                    uint32_t cnt_in = 0, only = 0;
                    bool ld = false;
                    for (uint32_t k = 0; k < nt; ++k) if (gid[k] == bestg) { cnt_in++; only = k; ld = lead[k]; }
                    if (cnt_in == 1) win = row[only];
                    else if (ld) {
                        const uint32_t L = best;                     // the leader
                        for (uint32_t k = 0; k < nt && !win; ++k) if (gid[k] == bestg && node[k] == L) win = row[k];
                        const uint32_t cnt = g.al_cnt[nb + L];
                        for (uint32_t t2 = 0; t2 < cnt && !win; ++t2) {
                            const uint32_t mnode = g.al[(nb + L) * g.ma + t2];
                            for (uint32_t k = 0; k < nt; ++k) if (gid[k] == bestg && node[k] == mnode) { win = row[k]; break; }
                        }
                    }
                }
            }
            s_fast = win;
            if (win) job_end[slot] = (win << 16) | (job_end[slot] & 0xFFFF);
        }
        __syncthreads();
        if (s_fast && !force_dfs) return;
    }

    // exact DFS for what the shortcut left undecided.  It is rare (none in the benchmark workload), so it
    // works out of this workgroup's HBM workspace: the kernel then needs ~1 KB of LDS and can start next
    // to the forward kernel of another chunk instead of waiting for 60 KB to drain.
    const VcTopoLds t = vc_topo_carve(gws, NC, EC, STK, g.ma);
    vc_topo_load(g, nb, eb, N, E, t, lane);
    __threadfence_block();
    __syncthreads();
    __shared__ uint32_t s_nrows;
    __shared__ int s_err;
    if (lane == 0) {
        uint32_t nr = 0;
        uint32_t mb = 0, me = 0;
        if (masked) { const uint32_t s0 = b.win_seq_off[w]; mb = b.seq_begin[s0 + layer]; me = b.seq_end[s0 + layer]; }
        s_err = vc_topo_dfs(t, N, STK, masked, mb, me, &nr);
        s_nrows = nr;
    }
    __threadfence_block();
    __syncthreads();
    if (s_err) { if (lane == 0) vc_fail(b, w, s_err, 14, s_nrows); return; }
    // exact rank of each tied row's node: the smallest wins (64 ties per sweep; ties beyond VC_MAXTIE sit in tie_over)
    uint32_t best_rank = 0xFFFFFFFFu, row = 0;
    for (uint32_t base = 0; base < nt; base += 64) {
        const uint32_t kk = base + lane;
        uint32_t myrank = 0xFFFFFFFFu, myrow = 0;
        if (kk < nt) {
            myrow = kk < VC_MAXTIE ? (uint32_t)tie_rows[(uint64_t)slot * VC_MAXTIE + kk] : tie_over[(uint64_t)slot * tie_over_stride + kk];
            const uint32_t node = dp.rank2node[nb + myrow - 1];
            for (uint32_t r = 0; r < s_nrows; ++r) if (t.rank[r] == node) { myrank = r; break; }
        }
        const uint32_t best = wave_min_u32(myrank);
        if (best < best_rank) {
            const unsigned long long m = __ballot(myrank == best && kk < nt);
            const int src = __ffsll((long long)m) - 1;
            row = (uint32_t)__shfl((int)myrow, src, 64);
            best_rank = best;
        }
    }
    if (lane == 0) job_end[slot] = (row << 16) | (job_end[slot] & 0xFFFF);
}

// a few resident workgroups walk the list of windows whose alignment ended in a tie (appended by k_fwd),
// so the thousands of untied windows cost nothing and nobody parks 60 KB of LDS per window
// (at most 72 VGPRs, the rest in scratch: nearly every launch finds an empty list, and a 127-register wave could not even be
// placed beside k_fwd's waves to find that out -- 77 -> 37 ms per 32 768 windows spent in these launches)
#ifndef VC_RESOLVE_OCC
#define VC_RESOLVE_OCC __attribute__((amdgpu_waves_per_eu(7, 8)))
#endif
VC_KL __global__ __launch_bounds__(64) VC_RESOLVE_OCC void k_resolve(VcBatchDev b, VcGraph g, VcDp dp, uint32_t w0, uint32_t nslots,
                                                uint32_t NC, uint32_t EC, uint32_t STK,
                                                const uint16_t* tie_rows, const uint32_t* tie_cnt, const uint32_t* tie_over, uint32_t tie_over_stride, uint32_t* job_end,
                                                const uint32_t* tie_list, const uint32_t* tie_n,
                                                const uint32_t* submask, int layer, uint8_t* workspace, uint32_t ws_bytes, int force_dfs,
                                                const uint32_t* cursor) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];     // visit bitmap [NC/32 + 1] words + stack [256] u16
    const uint32_t n = *tie_n;
    for (uint32_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
        __syncthreads();
        const uint32_t slot = tie_list[idx];
        vc_resolve_one(smem, workspace + (size_t)blockIdx.x * ws_bytes, slot, b, g, dp, w0, nslots, NC, EC, STK, tie_rows, tie_cnt, tie_over, tie_over_stride,
                       job_end, submask, (cursor && slot < nslots) ? (int)(cursor[slot] & 0xFFFFu) : layer, force_dfs);
    }
}

// ------------------------------------------------------------------------------------------------
// k_fwd: linear-gap graph DP, one alignment per wavefront (64-thread workgroup = 1 wave).
//   columns: lane l owns columns l*CPL+1 .. l*CPL+CPL (contiguous); column 0 is a per-row scalar
//   scores:  int16, two cells per 32-bit lane register, packed math (v_pk_add / v_pk_max_i16) -- the
//            same width the reference's SIMD engine picks (simd impl:699-754), no saturation needed
//            inside the envelope checked below
//   rows:    the previous row stays in registers; the last RING rows live in LDS; rows that a successor
//            more than RING rows away needs are re-read from the H matrix in HBM
//   output:  the H matrix (2 B/cell, coalesced 256-B stores), column 0 per row, the end cell; the
//            backtrack (k_trace) re-derives every move from H exactly as sisd:362-459 does
// mode: 0 build (NW), 1 re-alignment (NW for backbone/full-span else SW), 2 final SW of the backbone
// ------------------------------------------------------------------------------------------------
// Banded matrix store.  The backtrack of a global alignment stays close to the rank diagonal -- row i meets column
// i * len / rows -- so k_fwd writes, per row, only the VC_BAND_LANES lanes around it (192 instead of 768 bytes at 10 cells
// per lane: the stored matrix is 62 % of all the bytes this path moves through HBM, and HBM is what bounds it).  Rows that a
// later row reads back from the stored matrix (VC_RF_FULL, set by the row builders) are also written whole, as before.  A
// backtrack that needs a cell outside the band gives up and puts its alignment on a redo list: k_fwd runs again for those
// (whole rows), and the backtrack walks them from there.
// Layout: [row][VC_BAND_LANES][NDS dwords]; the band lanes store under their own exec mask, worked out on the scalar side
// (no per-row vector arithmetic for the band).  (A TILED layout for the reader -- one 128-byte line holding ONE lane's stored form for ten
// consecutive rows -- was built and measured twice, rounds 3 and 4: the backtrack gains 3 %, k_fwd loses 5 - 12 % to the sixteen partial lines
// a row then writes; the code is gone, NOTES.md has the numbers.)
// Both kernels take the band of a row from the same number per alignment (VcFwdArgs::band_par): the slope of the diagonal.
__device__ __forceinline__ uint32_t vc_band_slope(uint32_t len, uint32_t nrows, uint32_t cpl) {   // lanes per row, 16.16 fixed point
    return min((uint32_t)((((unsigned long long)len << 16) / nrows) / cpl), 0xFFFFFFu);
}
// Width of the band, in lanes, for a width class: 80 columns, at least 8 lanes.  Round 6 (profiles/r6_ab_band_width.txt): rounds 3-5 kept 16
// lanes and only ever tried MORE (24: - 5 %); fewer are faster -- config C 37.5 k (16) / 38.5 (12) / 39.2 (10) / 40.0 (8) / 38.2 (6) / 29.0 k
// (4 lanes) windows/s: a row of 8 lanes is 96 bytes instead of 192 (k_fwd stores half, the backtrack finds 1.3 rows per 128-byte line
// instead of 0.7), and the alignments that leave the band go from 0.14 % to 0.19 % (at 6 lanes: 0.6 %, and the catch-up rounds eat the
// gain).  What the walk needs is a width in COLUMNS: config E (20 columns per lane) leaves 8 lanes as rarely as 16.
// `chain` (development, -DVC_BAND_CHAIN_COLS=n: a narrower band for the re-alignment rounds, whose graphs are little more than a chain of the
// backbone's length): n columns, at least 4 lanes.  The forward kernels take it from their KEPT parameter (with the band on, KEPT <=> build
// phase), the backtrack from VcTraceArgs::band_chain.  Measured and left off: the exact DFS order of a pruned graph puts rows of distant
// backbone coordinates side by side often enough that 40 columns send 10 % of the re-alignments through the redo pass.
#ifndef VC_BAND_CHAIN_COLS
#define VC_BAND_CHAIN_COLS 0       // 0 = off (the product): 40 columns lose 10 % of the re-alignments to the redo pass, profiles/r6_ab_chain_band.txt
#endif
__host__ __device__ constexpr uint32_t vc_band_lanes(uint32_t cpl, bool chain = false) {
    const uint32_t full = cpl >= 10u ? 8u : (cpl >= 8u ? 10u : (cpl >= 6u ? 14u : (uint32_t)VC_BAND_LANES));
    if (!chain || VC_BAND_CHAIN_COLS == 0) return full;
    const uint32_t c = ((uint32_t)VC_BAND_CHAIN_COLS + cpl - 1u) / cpl;
    return c < 4u ? 4u : (c > full ? full : c);
}
__device__ __forceinline__ uint32_t vc_band_start(uint32_t i, uint32_t ql, uint32_t bl) {
    // lane of the diagonal at row i (i <= rows, so i * ql < 2^22 * ... fits 32 bits; the 24-bit multiply is the fast one).  Any
    // function would do as long as k_fwd and the backtrack use the same: a band that misses the path only costs a redo
    const uint32_t t = __umul24(i, ql) >> 16;
    return min(max(t, bl / 2u - 1u) - (bl / 2u - 1u), 64u - bl);
}
#define VC_BAND_JOB_PAD_DWORDS (VC_BAND_LANES * 128u / 4u)
__host__ __device__ inline uint64_t vc_band_job_dwords(uint64_t hstride) {       // a quarter of the whole rows (room for 16 of 64 lanes), + padding
    return hstride / 4 + VC_BAND_JOB_PAD_DWORDS;
}
// The band MOVES only every VC_BAND_ROWS rows (a power of two) -- rows (b * VC_BAND_ROWS + 1 ...) share the lanes of the diagonal at the
// block's middle.  The diagonal advances ~0.04 lanes per row, so a block of 16 rows shifts the band by two thirds of a lane at most, and
// k_fwd works out a band (scalar multiply, clamps, lane offsets) once per block instead of once per row.
#ifndef VC_BAND_ROWS
#define VC_BAND_ROWS 16     // (8 until the end of round 6; 16: + 0.5 % on config C, + 0.2 ... 1.9 % on the other shapes of the bench line, 4 % more alignments redone: profiles/r6_ab_band_rows_kept_ring.txt)
#endif
static_assert(VC_BAND_ROWS >= 1 && (VC_BAND_ROWS & (VC_BAND_ROWS - 1)) == 0, "rows per band block: a power of two");
__device__ __forceinline__ uint32_t vc_band_row_start(uint32_t r1, uint32_t ql, uint32_t bl) {        // r1 = row - 1
    return vc_band_start((r1 & ~(uint32_t)(VC_BAND_ROWS - 1)) + 1u + VC_BAND_ROWS / 2u, ql, bl);
}

struct VcFwdArgs {
    VcBatchDev b;
    VcDp dp;
    uint32_t w0, nslots, NC, EC;
    uint32_t group, k0;            // jobs per slot in this launch, first sequence index
    int mode;
    int do_init;                   // first width class of a launch group resets the per-job outputs
    int m, n, g;                   // NW scores
    int sm, sn, sg;                // SW scores
    uint32_t* hmat;                // [jobs * hstride] stored rows: byte-packed [64 lanes][NDS] or raw int16 pairs [CPL/2][64 lanes] dwords
    uint64_t  hstride;             // dwords per job
    int16_t*  c0;                  // [jobs * NC] H[i][0]
    uint32_t* job_end;             // [jobs] (row << 16) | col ; 0 = empty alignment
    uint8_t*  job_type;            // [jobs] 0 SW, 1 NW, 255 skipped
    uint16_t* tie_rows;            // [jobs * VC_MAXTIE] NW: sink rows sharing the best end score (incremental order only)
    uint32_t* tie_cnt;             // [jobs]
    uint32_t* tie_over;            // build phase: [jobs * tie_over_stride] ties beyond VC_MAXTIE (the job's pair list, not yet in use); else null
    uint32_t  tie_over_stride;
    uint32_t* tie_list;            // [jobs] windows whose alignment ended in a tie (build phase)
    uint32_t* tie_n;               // [1]
    unsigned long long* stat;      // [4] cells, rows, -, far-row reads
    uint32_t wcols;                // != 0: k_fwd_wide follows this launch and takes what the packed-int16 kernel declines
    uint32_t kept;                 // build phase: slots of the kept-row ring the forward records were made for (0: plain ring)
    uint32_t* bmat;                // band matrix of a job: bmat + job * vc_band_job_dwords(hstride) (see vc_band_start)
    uint32_t* band_par;            // [jobs] slope (lanes per row, 16.16) of the job's band
    int band;                      // 1: global alignments store the band (+ whole rows where VC_RF_FULL asks for them)
    const uint32_t* redo_list;     // != nullptr: this launch re-runs the listed jobs with whole rows (the backtrack left the band)
    const uint32_t* redo_n;
    uint32_t fold;                 // 1: the launch is built for the two widest classes of the batch and takes every narrower sequence in the lower one
    uint32_t lean;                 // classes of 32+ columns per lane build the row's match / mismatch profile on the fly (v_perm_b32 through a 4-entry table)
                                   //   instead of holding four profiles in 4 x CPL / 2 registers: possible when the batch holds A / C / G / T only and
                                   //   mismatch - gap == -1 (the selector's 0xFF constant).  bit 0: global alignments may, bit 1: local ones; otherwise
                                   //   such sequences go to k_fwd_wide
    const uint32_t* cursor;        // build phase, != nullptr: [nslots] the layer every window is at (bits 0..15) and, in bit 31, "its last backtrack
                                   //   left the band: this layer again, with whole rows" (see Plan::build_layer); k0 is then unused
#ifdef VC_LAB
    uint32_t dbg;                  // development (tools/gpu_fwd_lab.py): parts of the row loop switched off, timing only
#endif
};
#ifdef VC_LAB
#define VC_LABF(f) ((a.dbg & (f)) != 0)
#else
#define VC_LABF(f) false
#endif

#define VC_DPP_SHR(v, old, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xF, false)

typedef short vc_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(vc_s2, a), __builtin_bit_cast(vc_s2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (vc_s2)(__builtin_bit_cast(vc_s2, a) + __builtin_bit_cast(vc_s2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (vc_s2)(__builtin_bit_cast(vc_s2, a) - __builtin_bit_cast(vc_s2, b)));
}
// d.lo = a.lo ; d.hi = max(a.hi, a.lo)
__device__ __forceinline__ uint32_t pk_max_hi_with_lo(uint32_t a) {
    uint32_t d;
    asm("v_pk_max_i16 %0, %1, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
// d.lo = max(a.lo, b.hi) ; d.hi = max(a.hi, b.hi)
__device__ __forceinline__ uint32_t pk_max_bcast_hi(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_dup(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }
__device__ __forceinline__ int pk_lo(uint32_t a) { return (int)(short)(a & 0xFFFF); }
__device__ __forceinline__ int pk_hi(uint32_t a) { return (int)a >> 16; }

// Stored form of a DP row.  Along a row the tilted scores never decrease and two neighbours differ by at
// most max(m, n) - 2g (diagonal vs the vertical-then-horizontal detour), so inside a lane's CPL cells a
// score stays within (CPL-1)*(max(m,n)-2g) of the lane's first cell.  When that fits a byte (99 for the
// scores VeChat uses at 10 cells per lane) only the LOW BYTE of every cell is stored, followed by the
// first cell as a full int16: bytes 0..CPL-1 = cells, bytes CPL, CPL+1 = anchor.  A reader rebuilds
// T_c = anchor + ((byte_c - anchor) & 0xFF).  That is CPL+2 bytes per lane per row instead of 2*CPL, and
// packing costs one v_perm_b32 per four cells.  Other score sets keep raw packed int16 pairs.
__host__ __device__ constexpr int vc_nds(int cpl) { return (cpl + 2 + 3) / 4; }
// work counters: VC_STAT_SLOTS sets of 8, picked by block, summed by vc_get_stats
#define VC_STAT_SLOTS 64
__device__ __forceinline__ unsigned long long* vc_stat_slot(unsigned long long* stat) { return stat + (blockIdx.x % VC_STAT_SLOTS) * 8; }
// behind the counter sets: [VC_STAT_SLOTS][2] shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) the forward waves spent in their
// row loops -- their ratio is the shader clock the chip sustained under the job (bench.py: roofline.sclk_mhz)
#define VC_STAT_WORDS (8 * VC_STAT_SLOTS + 2 * VC_STAT_SLOTS)
__device__ __forceinline__ unsigned long long* vc_clk_slot(unsigned long long* stat) { return stat + 8 * VC_STAT_SLOTS + (blockIdx.x % VC_STAT_SLOTS) * 2; }

__host__ __device__ inline bool vc_row_packed(int m, int n, int g, int cpl) {
    return g < 0 && (cpl - 1) * ((m > n ? m : n) - 2 * g) <= 255;
}

template <int ND, int NDS>
__device__ __forceinline__ void vc_pack_row(const uint32_t (&T)[ND], uint32_t (&w)[NDS]) {
#pragma unroll
    for (int t = 0; t < ND / 2; ++t) w[t] = __builtin_amdgcn_perm(T[2 * t + 1], T[2 * t], 0x06040200u);
    if (ND & 1) w[ND / 2] = __builtin_amdgcn_perm(T[0], T[ND - 1], 0x05040200u);     // two cells + the anchor
    else        w[ND / 2] = T[0];                                                    // anchor in the low half
}
template <int ND, int NDS>
__device__ __forceinline__ void vc_unpack_row(const uint32_t (&w)[NDS], uint32_t (&T)[ND]) {
    const uint32_t aw = (ND & 1) ? (w[ND / 2] >> 16) : (w[ND / 2] & 0xFFFFu);
    const uint32_t a2 = aw * 0x10001u, alo = (aw & 0xFFu) * 0x10001u;
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        // bytes 2q, 2q+1 of the cell string -> low bytes of the two halves
        const uint32_t src = w[q / 2];
        const uint32_t two = (q & 1) ? __builtin_amdgcn_perm(0u, src, 0x0C030C02u) : __builtin_amdgcn_perm(0u, src, 0x0C010C00u);
        T[q] = pk_add(a2, pk_sub(two, alo) & 0x00FF00FFu);
    }
}
// one cell of a packed row (backtrack): words of the lane that owns the cell, cell index cc inside the lane
__device__ __forceinline__ int vc_packed_cell(const uint32_t* w, uint32_t cc, uint32_t cpl) {
    const uint32_t wc = w[cc >> 2], wa = w[cpl >> 2];                 // both unconditional: one round trip
    const uint32_t an = (wa >> ((cpl & 2) * 8)) & 0xFFFFu;
    const uint32_t b = (wc >> ((cc & 3) * 8)) & 0xFFu;
    return (int)(short)an + (int)((b - an) & 0xFFu);
}

// Rows written by k_fwd_dt (vc_fwd_dt.h) are DOUBLY tilted: T''[i][j] = H[i][j] - (i + j) * g, kept as unsigned 16-bit numbers (job_type bit
// VC_JOB_DT).  A reader turns a cell it has rebuilt with vc_packed_cell back into the singly tilted T the backtrack compares.
#define VC_JOB_DT 8u
__device__ __forceinline__ int vc_dt_cell(int v, uint32_t row, int g) { return (v & 0xFFFF) + (int)row * g; }

// The forward DP works on the TILTED matrix T[i][j] = H[i][j] - j*g.  In that domain the horizontal
// pass of sisd :347-349 is a plain prefix maximum (T[i][j] = max(T[i][j], T[i][j-1])), the vertical move
// adds g, and the diagonal move adds (P[c][j] - g) -- folded into the profile once.  Since the row's
// profile and g do not depend on the predecessor,
//     max_p (H[p][j-1] + P[j]) = (max_p H[p][j-1]) + P[j],   max_p (H[p][j] + g) = (max_p H[p][j]) + g,
// so each additional in-edge costs one packed max per register instead of a full relaxation, and the
// order of the in-edges is irrelevant here (it matters only to the backtrack, which follows it).
// The alignment a forward wave works on.  Lock-step launches derive it from the workgroup index (vc_fwd_pick); the waves of the
// persistent build pipeline (vc_pipe.h) take it from their work queue.
struct VcJob {
    uint32_t job;       // index of the per-alignment buffers (stored matrix, column 0, end cell, ties)
    uint32_t slot;      // window of the chunk
    uint32_t k;         // sequence of the window
    bool redo;          // second pass with whole rows (the backtrack left the band)
};
// what vc_fwd_body did with its job
#define VC_FWD_NONE 0u      // nothing to do here (another width class, no such sequence, window not OK, outside the envelope)
#define VC_FWD_DONE 1u
#define VC_FWD_TIE  2u      // done, and the end cell is tied between sinks on a non-reference order: the resolver decides

template <int CPL, int RING, bool NWT, bool PACKED, bool KEPT, bool PIPE = false>
__device__ __forceinline__ uint32_t vc_fwd_body(const VcFwdArgs& a, uint32_t* ring_raw, const VcJob& jb) {
    // KEPT: the LDS ring holds RING SLOTS for the rows a later row reads back (vc_frec_kept gave every such row its slot);
    // otherwise the last RING rows, slot = row % RING
    static_assert(KEPT || (RING & (RING - 1)) == 0, "ring slots are taken with a mask");
    constexpr int ND = CPL / 2;              // packed int16 dwords per lane per row
    constexpr int NDS = vc_nds(CPL);         // dwords per lane per row in the packed stored form
    uint32_t (*ring)[ND][64] = reinterpret_cast<uint32_t (*)[ND][64]>(ring_raw);
    const int lane = vc_lane();
    const bool redo = jb.redo;                                // second pass over the alignments whose backtrack left the band
    const uint32_t job = jb.job, slot = jb.slot, k = jb.k;
    const uint32_t w = a.w0 + slot;
    if (a.do_init && lane == 0 && !redo) { a.job_type[job] = 255; a.job_end[job] = 0; a.tie_cnt[job] = 0; }
    if (a.b.status[w] != VC_WIN_OK) return VC_FWD_NONE;
    const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
    if (k >= ns) return VC_FWD_NONE;
    const uint64_t so = a.b.seq_off[s0 + k];
    const uint32_t len = (uint32_t)(a.b.seq_off[s0 + k + 1] - so);
    // another width class handles this sequence.  (A folded launch -- the persistent pipeline always, a lock-step launch over more
    // than two classes -- is built for the two widest classes of a batch and takes everything narrower in the lower of them: a lane
    // simply owns more columns than the sequence needs, the matrix is the same -- its backtrack reads the rows in the same class,
    // VcTraceArgs::cpl_lo.)
    if ((PIPE || a.fold) ? len > 64u * CPL : vc_cpl_for(len) != (uint32_t)CPL) return VC_FWD_NONE;
    const uint32_t L = (uint32_t)(a.b.seq_off[s0 + 1] - a.b.seq_off[s0]);
    // NW or SW is fixed per instantiation (the caller looked at the layer, window.cpp:336-349): the row loop
    // then carries no alignment-type branches
    constexpr bool nw = NWT;
    {
        bool want = true;
        if (a.mode == 2) want = false;
        else if (a.mode == 1) want = (k == 0) || vc_full_span(a.b.seq_begin[s0 + k], a.b.seq_end[s0 + k], L);
        if (want != nw) return VC_FWD_NONE;
    }
    const int m = nw ? a.m : a.sm, n = nw ? a.n : a.sn, g = nw ? a.g : a.sg;
    const uint32_t nrows = a.dp.nrows[slot];
    const uint64_t nb = (uint64_t)slot * a.NC;
    // envelope: int16 on the tilted matrix (vc_int16_ok), the sequence inside this instantiation's columns.  The SW end-cell rule
    // below also relies on negative mismatch / gap scores (cells past the sequence end can then never strictly exceed the best
    // real cell).
    {
        const bool ok = vc_int16_ok(m, n, g, nrows, CPL, nw) && len <= 64u * CPL && len > 0 && nrows > 0 && !(a.dp.flags[slot] & 1u) &&
                        (CPL < 32 || ((a.lean >> (nw ? 0 : 1)) & 1u));                 // (the widest classes exist in the lean form only)
        if (!ok) {
            // outside the packed-int16 envelope: not an error -- the job keeps type 255 and k_fwd_wide (int32 lanes, any
            // length, like the reference's fallback to 32-bit lanes, simd impl:699-706) takes it
            if (len == 0 || nrows == 0 || (a.dp.flags[slot] & 1u)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 3, (a.dp.flags[slot] & 1u) ? 1 : 2); }
            else if (a.wcols == 0) { if (lane == 0) vc_fail(a.b, w, VC_WIN_OVERFLOW, 27, nrows); }    // the host planned no k_fwd_wide: say so, do not skip silently
            return VC_FWD_NONE;
        }
    }
    if (lane == 0 && !redo) {
        a.job_type[job] = nw ? 1 : 0;
        unsigned long long* st = vc_stat_slot(a.stat);
        atomicAdd(st + 0, (unsigned long long)nrows * len);
        atomicAdd(st + 1, (unsigned long long)nrows);
    }
    if (lane == 0 && redo) atomicAdd(vc_stat_slot(a.stat) + 7, 1ull);

    // tilted match/mismatch profile (score - g) of my columns for the four usual bases (packed pairs);
    // other row bytes are compared on the fly.  Columns beyond the sequence end never match.
    // LEAN (32 and more columns per lane): four profiles would be 4 x ND registers -- 279 to 380 VGPRs with up to 156 spilled, one wave per
    // SIMD.  Instead one selector per register pair: byte pairs (2c, 2c + 1) pick entry c of a four-entry int16 table {A, C, G, T} that the
    // row sets up on the scalar side (entry = m - g for the row's base, n - g for the others); columns past the sequence end take the
    // selector's constant 0xFF = -1, which is n - g for the scores this form is used with (VcFwdArgs::lean).  One v_perm_b32 more per
    // register pair and row, a quarter of the registers.
    constexpr bool LEAN = CPL >= 32;
    uint32_t pfA[LEAN ? 1 : ND], pfC[LEAN ? 1 : ND], pfG[LEAN ? 1 : ND], pfT[LEAN ? 1 : ND], sbp[LEAN ? 1 : ND], sel[LEAN ? ND : 1];
    const int mt = m - g, nt = n - g;
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        const uint32_t i0 = lane * CPL + 2 * q, i1 = i0 + 1;
        const uint32_t b0 = i0 < len ? a.b.bases[so + i0] : 0xFFu;
        const uint32_t b1 = i1 < len ? a.b.bases[so + i1] : 0xFFu;
        if constexpr (LEAN) {
            auto pick = [](uint32_t b) -> uint32_t {
                const uint32_t c2 = b == 'A' ? 0u : b == 'C' ? 2u : b == 'G' ? 4u : b == 'T' ? 6u : 0xFFu;
                return c2 == 0xFFu ? 0x0D0Du : (c2 | ((c2 + 1u) << 8));
            };
            sel[q] = pick(b0) | (pick(b1) << 16);
        } else {
            sbp[q] = b0 | (b1 << 16);
            auto sc = [&](uint32_t x) { return ((uint32_t)((b0 == x) ? mt : nt) & 0xFFFFu) | ((uint32_t)((b1 == x) ? mt : nt) << 16); };
            pfA[q] = sc('A'); pfC[q] = sc('C'); pfG[q] = sc('G'); pfT[q] = sc('T');
        }
    }
    const uint32_t gg = pk_dup(g);
    uint32_t njg[ND];                         // -(j*g) of my columns: the tilted image of H == 0 (SW floor, SW row 0)
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        const int j0 = lane * CPL + 2 * q + 1;
        njg[q] = ((uint32_t)(-j0 * g) & 0xFFFFu) | ((uint32_t)(-(j0 + 1) * g) << 16);
    }

    uint32_t* const hrow0 = a.hmat + (uint64_t)job * a.hstride;
    constexpr bool packed = PACKED;           // the stored row form is a property of the launch (host: both score sets fit the byte bound)
    // banded store: global alignments only (a local alignment may end and start anywhere), byte-packed rows only
    // (round 6, -DVC_EXPERIMENTS builds with VC_BAND_RAW=1: raw int16 rows too -- the widest classes, and scores whose rows do not fit the byte
    // form: [row][band lane][ND dwords]; bit-identical and 6 % slower on 3 kb windows, profiles/r6_ab_raw_band.txt)
#ifdef VC_EXPERIMENTS
    constexpr bool RAW_BAND = true;
#else
    constexpr bool RAW_BAND = false;
#endif
    const bool band = NWT && (PACKED || RAW_BAND) && a.band && !redo;
    const char* const brow0 = reinterpret_cast<const char*>(a.bmat + (uint64_t)job * vc_band_job_dwords(a.hstride));
    __amdgpu_buffer_rsrc_t brs;                                // the job's band rows behind a buffer descriptor (wave-uniform by construction)
    {
        const uintptr_t bb = reinterpret_cast<uintptr_t>(brow0);
        const uint64_t bjb = vc_band_job_dwords(a.hstride) * 4ull;
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bb), hi_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bb >> 32));
        brs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi_ << 32) | lo_), 0,
                                                (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bjb < 0xFFFFF000ull ? bjb : 0xFFFFF000ull)), 0x00020000);
    }
    const uint32_t band_ql = (uint32_t)__builtin_amdgcn_readfirstlane((int)vc_band_slope(len, nrows, CPL));   // a scalar: the band of a row is worked out on the scalar side
    if (band && lane == 0) a.band_par[job] = band_ql;
    constexpr uint32_t BL = vc_band_lanes(CPL, !KEPT);          // lanes of a band row in this width class and phase
    constexpr uint32_t TLB = (PACKED ? NDS : ND) * 4u, TBB = BL * TLB;   // a lane's bytes in a band row, a band row
    uint32_t t_off = 0u - TBB;                                 // byte offset of the band row in work (scalar)
    uint32_t b_rin = 0;                                        // rows left in the current band block (vc_band_row_start)
    unsigned long long t_mask = 0;                             // lanes of the block's band
    uint32_t t_lane = 0;                                       // my byte offset inside a block: (lane - first band lane) * TLB
    const uint32_t lane_tlb = (uint32_t)lane * TLB;
    int16_t* const c0p_out = a.c0 + (uint64_t)job * a.NC;
    const uint16_t* const ovfp = a.dp.ovf + (uint64_t)slot * a.EC;

    // end-cell tracking
    int best = nw ? VC_INT_MIN : 0;          // NW: uniform (tilted value of the last column); SW: per lane (real H)
    uint32_t best_row = 0, ntie = 0;
    const uint32_t lane_e = (len - 1) / CPL, c_e = (len - 1) % CPL;
    uint32_t far_reads = 0;

    uint32_t acc[ND];                         // between iterations: T of the row just finished
    int c0prev = 0;
    int c0vec = 0;                            // lane t: column 0 of the latest row r with (r - 1) % 64 == t
#pragma unroll
    for (int q = 0; q < ND; ++q) acc[q] = 0;

    // row records (the forward view, VcDp::frec): lane t holds the record of row (block*64 + t); the next
    // block is fetched a block ahead
    uint4 myrec = make_uint4(0, 0, 0, 0), nextrec = make_uint4(0, 0, 0, 0);
    if ((uint32_t)lane < nrows) nextrec = a.dp.frec[nb + lane];
    constexpr uint32_t rowdw = PACKED ? NDS * 64 : ND * 64;     // dwords per stored row
    const uint32_t loff = (PACKED ? lane * NDS : lane) * 4u;    // my BYTE offset inside a stored row
    uint32_t srow = 0;                                          // byte offset of the current row (scalar; < 4 GB per job)

    // a row of the LDS ring merged into the running maximum; column 0 of the last 64 rows lives in c0vec
    auto ring_slot_merge = [&](uint32_t slot, uint32_t c0lane, int& c0m) __attribute__((always_inline)) {
        const uint32_t* rp = ring_raw + slot * (ND * 64) + lane;
        uint32_t hp[ND];
#pragma unroll
        for (int q = 0; q < ND; ++q) hp[q] = rp[q * 64];
        const int c0p = __builtin_amdgcn_readlane(c0vec, c0lane);
        // in place (tied operand): at the join behind the predecessor handling acc is then ONE value on every path, and the
        // compiler has no copies to insert there (it did: ten v_mov per row that is not "plain")
#pragma unroll
        for (int q = 0; q < ND; ++q) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(acc[q]) : "v"(hp[q]));
        c0m = max(c0m, c0p);
    };
    // a listed predecessor known to sit in the ring: `e` is its forward entry (plain ring: the distance; kept-row ring:
    // 0x8000 | slot << 8 | distance)
    auto ring_merge = [&](uint32_t i, uint32_t e, int& c0m) __attribute__((always_inline)) {
        if (KEPT) ring_slot_merge((e >> 8) & 7u, (i - (e & 0x7Fu) - 1u) & 63u, c0m);
        else ring_slot_merge((i - e) & (RING - 1), (i - e - 1u) & 63u, c0m);
    };

    // the DP of one row once the predecessor maximum stands in acc / c0m (two call sites: common rows, general-path rows)
    auto row_tail = [&](const uint32_t r0, const int c0m, const uint32_t i, const uint32_t ri) __attribute__((always_inline)) {
        // ---- this row: diagonal (cell j-1 of the maximum: shift right by one int16; the hole is filled by
        // the left lane's last cell, lane 0 takes column 0) and vertical candidates
        const uint32_t left = (uint32_t)VC_DPP_SHR((int)acc[ND - 1], (int)((uint32_t)c0m << 16), 0x138, 0xF);
        uint32_t P[ND];
#pragma unroll
        for (int q = 0; q < ND; ++q) P[q] = __builtin_amdgcn_alignbit(acc[q], q == 0 ? left : acc[q - 1], 16);
        const uint32_t bi = (r0 >> 24) & 7u;
        if constexpr (LEAN) {
            // the row's table on the scalar side: entries A, C in the low dword, G, T in the high one (a row byte outside A/C/G/T cannot
            // occur in a batch that passed VcFwdArgs::lean; it would score as a mismatch everywhere)
            const uint32_t x = (uint32_t)(mt ^ nt) & 0xFFFFu, nt2 = pk_dup(nt);
            const uint32_t lo = nt2 ^ (bi == 0 ? x : bi == 1 ? x << 16 : 0u), hi = nt2 ^ (bi == 2 ? x : bi == 3 ? x << 16 : 0u);
#pragma unroll
            for (int q = 0; q < ND; ++q) P[q] = pk_add(P[q], __builtin_amdgcn_perm(hi, lo, sel[q]));
        } else
        if (bi < 2) {
            if (bi == 0) {
#pragma unroll
                for (int q = 0; q < ND; ++q) P[q] = pk_add(P[q], pfA[q]);
            } else {
#pragma unroll
                for (int q = 0; q < ND; ++q) P[q] = pk_add(P[q], pfC[q]);
            }
        } else if (bi == 2) {
#pragma unroll
            for (int q = 0; q < ND; ++q) P[q] = pk_add(P[q], pfG[q]);
        } else if (bi == 3) {
#pragma unroll
            for (int q = 0; q < ND; ++q) P[q] = pk_add(P[q], pfT[q]);
        } else {
            const uint32_t x = r0 & 0xFF;
#pragma unroll
            for (int q = 0; q < ND; ++q)
                P[q] = pk_add(P[q], ((uint32_t)(((sbp[q] & 0xFFFFu) == x) ? mt : nt) & 0xFFFFu) | ((uint32_t)(((sbp[q] >> 16) == x) ? mt : nt) << 16));
        }
#pragma unroll
        for (int q = 0; q < ND; ++q) P[q] = pk_max(P[q], pk_add(acc[q], gg));
        // column 0: NW max over predecessors + g (Initialize, sisd :210-222); SW 0
        const int col0 = nw ? c0m + g : 0;
        if (!nw) {                                         // SW floor H >= 0 (sisd :350-352)
#pragma unroll
            for (int q = 0; q < ND; ++q) P[q] = pk_max(P[q], njg[q]);
        }
#ifdef VC_PAD_VALU
        {   // development: VC_PAD_VALU independent vector instructions per row (is the job bound by VALU issue?)
            uint32_t pad = (uint32_t)lane;
#pragma unroll
            for (int t = 0; t < VC_PAD_VALU; ++t) asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad) : "v"(pfA[t % ND]));
            asm volatile("" :: "v"(pad));
        }
#endif
        // ---- horizontal pass (sisd :347-349): prefix maximum, in-lane then across lanes
        P[0] = pk_max_hi_with_lo(P[0]);
#pragma unroll
        for (int q = 1; q < ND; ++q) P[q] = pk_max_bcast_hi(pk_max_hi_with_lo(P[q]), P[q - 1]);
        // the scan runs on the raw dword of the lane's last pair: as an int32 it orders by its high half (the lane's running
        // maximum), the low half only decides between equal high halves -- and only the high half of the result is used
        int sc = (int)P[ND - 1];
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x111, 0xF));
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x112, 0xF));
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x114, 0xF));
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x118, 0xF));
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x142, 0xA));
        sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x143, 0xC));
        int carry = VC_DPP_SHR(sc, VC_INT_MIN, 0x138, 0xF);
        carry = max(carry, (int)((uint32_t)col0 << 16));   // column 0 enters as T[i][0] = H[i][0]
#pragma unroll
        for (int q = 0; q < ND; ++q) acc[q] = pk_max_bcast_hi(P[q], (uint32_t)carry);       // both halves against the carry's high half

        // ---- end cell
        if (nw) {
            if (r0 & (VC_RF_SINK << 8)) {                    // sisd :353-355 (same column: tilted compare is exact)
                uint32_t hv = acc[0];
#pragma unroll
                for (int q = 1; q < ND; ++q) hv = (c_e / 2 == (uint32_t)q) ? acc[q] : hv;
                int v = (c_e & 1) ? pk_hi(hv) : pk_lo(hv);
                v = __builtin_amdgcn_readlane(v, lane_e);
                if (v > best) {
                    best = v; best_row = i; ntie = 1;
                    if (lane == 0) a.tie_rows[(uint64_t)job * VC_MAXTIE] = (uint16_t)i;
                } else if (v == best) {
                    if (lane == 0) {
                        if (ntie < VC_MAXTIE) a.tie_rows[(uint64_t)job * VC_MAXTIE + ntie] = (uint16_t)i;
                        else if (a.tie_over && ntie < a.tie_over_stride) a.tie_over[(uint64_t)job * a.tie_over_stride + ntie] = i;
                    }
                    ntie++;
                }
            }
        } else {                                             // sisd :350-352 (first row with the best score)
            uint32_t rmx = pk_sub(acc[0], njg[0]);
#pragma unroll
            for (int q = 1; q < ND; ++q) rmx = pk_max(rmx, pk_sub(acc[q], njg[q]));
            const int rm = max(pk_lo(rmx), pk_hi(rmx));
            if (rm > best) { best = rm; best_row = i; }
        }

        // ---- keep the row: registers (acc), LDS ring, HBM
        c0prev = col0;
        // lane ri keeps this row's column 0.  v_writelane_b32 takes ONE scalar register (constant bus), and both the value and the lane are
        // scalars: the lane goes through m0, which the clobber list tells the compiler (this clang has no __builtin_amdgcn_writelane, and
        // warns that m0 is "reserved" -- it is, the compiler keeps nothing there across an asm that names it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(c0vec) : "s"(col0), "s"(ri) : "m0");
#pragma clang diagnostic pop
        // one wave per workgroup: LDS operations of a wave retire in order, so no s_barrier (and no
        // vmcnt(0) drain of the H stores) is needed -- only keep the compiler from reordering
        __builtin_amdgcn_wave_barrier();
        if (VC_LABF(2)) {
        } else if (!KEPT || (r0 & (VC_RF_KEEP << 8))) {             // kept-row ring: only rows a later row reads back, in the slot the row builder chose
            uint32_t* wp = ring_raw + (KEPT ? ((r0 >> 27) & 7u) : (i & (RING - 1))) * (ND * 64) + lane;
#pragma unroll
            for (int q = 0; q < ND; ++q) wp[q * 64] = acc[q];
        }
        // the band of this row: its byte offset (scalar) and, every VC_BAND_ROWS rows, the lanes of the next block
        auto band_advance = [&]() __attribute__((always_inline)) {
            t_off += TBB;
            const bool newblock = b_rin == 0;         // the band moves every VC_BAND_ROWS rows
            if (newblock) b_rin = VC_BAND_ROWS;
            b_rin--;
            if (newblock) {                           // next row block: its band, on the scalar side
                // vc_band_start of the block's middle row in scalar arithmetic (row and slope are uniform; the product stays below 2^23,
                // so the plain multiply equals the 24-bit one the backtrack uses).  As vector code, once per row -- v_mul_u32_u24, a
                // clamped subtract, a minimum, v_readfirstlane and a quarter-rate v_mul_lo_u32 for the lane offset -- this was 5 of a
                // row's ~59 vector instructions
                const uint32_t bt_ = ((i + VC_BAND_ROWS / 2u) * band_ql) >> 16;
                // (in assembly: left to itself the compiler clamps with v_med3_u32 / a saturating v_sub -- only the vector ALU has those --
                // and multiplies the lane offset with a quarter-rate v_mad_u64_u32)
                constexpr uint32_t BLO = BL / 2 - 1, BHI = BLO + 64u - BL;
                uint32_t bs, bso;
                asm("s_max_u32 %0, %2, %3\n\ts_min_u32 %0, %0, %4\n\ts_sub_u32 %0, %0, %3\n\ts_mul_i32 %1, %0, %5"
                    : "=&s"(bs), "=s"(bso) : "s"(bt_), "n"(BLO), "n"(BHI), "n"(TLB) : "scc");
                t_mask = (unsigned long long)((1u << BL) - 1u) << bs;
                t_lane = lane_tlb - bso;
            }
        };
        if (VC_LABF(1)) {                                    // development (tools/gpu_fwd_lab.py): time the row loop without its stores
        } else if (PACKED) {
            uint32_t wv[NDS];
            vc_pack_row<ND, NDS>(acc, wv);
            auto put = [&](uint32_t* hr) __attribute__((always_inline)) {
                if constexpr (NDS == 2) *reinterpret_cast<uint2*>(hr) = make_uint2(wv[0], wv[1]);
                else if constexpr (NDS == 4) *reinterpret_cast<uint4*>(hr) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                else if constexpr (NDS == 3) { struct __attribute__((packed, aligned(4))) u3 { uint32_t a, b, c; }; *reinterpret_cast<u3*>(hr) = u3{wv[0], wv[1], wv[2]}; }
                else {
#pragma unroll
                    for (int t = 0; t < NDS; ++t) hr[t] = wv[t];
                }
            };
            // whole row: always without the band; with it only where a later row reads the row back (VC_RF_FULL)
            if (!band || (r0 & (VC_RF_FULL << 8))) put(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hrow0) + srow + loff));
            if (band) {
                band_advance();
                // the band lanes store through a raw buffer descriptor over the job's band rows: a lane outside the band carries an offset beyond
                // every range and the hardware drops it (tools/buffer_probe.hip).  (Until round 6: exec-mask writes around a store inside an asm
                // block -- invisible to the compiler's hazard pass; a gfx950 store-data hazard there cost round 5 a day, NOTES.md.)
                const uint32_t soff = t_off;
                const uint32_t voff = ((t_mask >> lane) & 1ull) ? t_lane : 0x80000000u;
                typedef uint32_t vc_u2 __attribute__((ext_vector_type(2)));
                typedef uint32_t vc_u3 __attribute__((ext_vector_type(3)));
                typedef uint32_t vc_u4 __attribute__((ext_vector_type(4)));
                if constexpr (NDS == 2) { const vc_u2 d = {wv[0], wv[1]}; __builtin_amdgcn_raw_buffer_store_b64(d, brs, voff, soff, 0); }
                else if constexpr (NDS == 3) { const vc_u3 d = {wv[0], wv[1], wv[2]}; __builtin_amdgcn_raw_buffer_store_b96(d, brs, voff, soff, 0); }
                else {
                    const vc_u4 d = {wv[0], wv[1], wv[2], wv[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(d, brs, voff, soff, 0);
#pragma unroll
                    for (int t = 4; t < NDS; ++t) __builtin_amdgcn_raw_buffer_store_b32(wv[t], brs, voff + 4u * t, soff, 0);
                }
            }
        } else {
            // full 256-B rows on purpose: masking the lanes past the sequence end was measured SLOWER
            // (partial cache-line writes), although it would save 20 % of the bytes
            // (with the band: only the rows a later row reads back, VC_RF_FULL -- as for the byte-packed rows)
            if (!band || (r0 & (VC_RF_FULL << 8))) {
                uint32_t* hr = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hrow0) + srow + loff);
#pragma unroll
                for (int q = 0; q < ND; ++q) hr[q * 64] = acc[q];
            }
            if (band) {
                // raw band row: [band lane][ND dwords], a lane's cells side by side; lanes outside the band are dropped by the descriptor's range check
                band_advance();
                const uint32_t soff = t_off;
                const uint32_t voff = ((t_mask >> lane) & 1ull) ? t_lane : 0x80000000u;
                typedef uint32_t vc_u2 __attribute__((ext_vector_type(2)));
                typedef uint32_t vc_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int q = 0; q + 4 <= ND; q += 4) { const vc_u4 d = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]}; __builtin_amdgcn_raw_buffer_store_b128(d, brs, voff + 4u * q, soff, 0); }
                if constexpr ((ND & 3) >= 2) { const vc_u2 d = {acc[ND & ~3], acc[(ND & ~3) + 1]}; __builtin_amdgcn_raw_buffer_store_b64(d, brs, voff + 4u * (ND & ~3), soff, 0); }
                if constexpr (ND & 1) __builtin_amdgcn_raw_buffer_store_b32(acc[ND - 1], brs, voff + 4u * (ND - 1), soff, 0);
            }
        }
        srow += rowdw * 4u;
        __builtin_amdgcn_wave_barrier();
    };

    const unsigned long long clk_w0 = wall_clock64(), clk_c0 = clock64();
    for (uint32_t i0 = 1; i0 <= nrows; i0 += 64) {              // blocks of 64 rows: one record fetch, one column-0 flush
      myrec = nextrec;
      {
          const uint32_t r = i0 - 1 + 64 + lane;
          if (r < nrows) nextrec = a.dp.frec[nb + r];
      }
      const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, nrows - i0 + 1));
      for (uint32_t ri = 0; ri < cnt; ++ri) {
        const uint32_t i = i0 + ri;
        const uint32_t r0 = __builtin_amdgcn_readlane(myrec.x, ri);

        // ---- element-wise maximum over the predecessor rows (and over their column 0).  The commonest row has the row
        // above as its only predecessor: acc and c0prev are that maximum already
        int c0m = c0prev;
        if (r0 & (VC_RF_SLOW << 8)) {
            // ---- rows whose list needs the general path (~3 %) take their own copy of the row body: no join with the
            // common path in front of the DP, so the compiler keeps acc in place there (it used to shuffle it through
            // ten copies per row)
            if (!(r0 & (VC_RF_PREV << 8))) {
                c0m = VC_INT_MIN;
#pragma unroll
                for (int q = 0; q < ND; ++q) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[q]) : "s"(0x80008000u));
            }
            const uint32_t nq = (r0 >> 16) & 0xFF;
            {
                // general path: the virtual row 0 analytically, a recent row from the LDS ring, an older one
                // back from the stored matrix in HBM; long lists come from VcDp::ovf
                const uint32_t fl = (r0 >> 8) & 0xFF;
                const uint32_t r1 = __builtin_amdgcn_readlane(myrec.y, ri);
                const uint32_t r2 = __builtin_amdgcn_readlane(myrec.z, ri);
                const uint32_t r3 = __builtin_amdgcn_readlane(myrec.w, ri);
                const uint32_t nlist = (fl & VC_RF_OVF) ? r2 : nq;
                for (uint32_t p = 0; p < nlist; ++p) {
                    uint32_t delta;
                    if (fl & VC_RF_OVF) {
                        delta = ovfp[r1 + p];
                        if ((fl & VC_RF_PREV) && delta == 1) continue;
                        if (KEPT && (delta & 0x8000u)) { ring_merge(i, delta, c0m); continue; }
                    } else {
                        const uint32_t wsel = p < 2 ? r1 : (p < 4 ? r2 : r3);
                        delta = (p & 1) ? (wsel >> 16) : (wsel & 0xFFFF);
                        if (KEPT && (delta & 0x8000u)) { ring_merge(i, delta, c0m); continue; }      // this one is in the ring
                    }
                    const uint32_t pr = i - delta;
                    if (pr == 0) {                                                   // H[0][j] = j*g (NW) / 0 (SW); column 0: 0
#pragma unroll
                        for (int q = 0; q < ND; ++q) { const uint32_t z = nw ? 0u : njg[q]; asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(acc[q]) : "v"(z)); }
                        c0m = max(c0m, 0);
                    } else if (!KEPT && delta <= (uint32_t)RING) {
                        ring_merge(i, delta, c0m);
                    } else {
                        uint32_t hA[ND];
                        __threadfence_block();                                        // my own earlier stores must have landed
                        if (PACKED) {
                            const uint32_t* hr = hrow0 + (uint64_t)(pr - 1) * (NDS * 64) + lane * NDS;
                            uint32_t wv[NDS];
#pragma unroll
                            for (int t = 0; t < NDS; ++t) wv[t] = hr[t];
                            vc_unpack_row<ND, NDS>(wv, hA);
                        } else {
                            const uint32_t* hr = hrow0 + (uint64_t)(pr - 1) * (ND * 64);
#pragma unroll
                            for (int q = 0; q < ND; ++q) hA[q] = hr[q * 64 + lane];
                        }
                        far_reads++;
                        int cA;
                        if (delta <= 64) cA = __builtin_amdgcn_readlane(c0vec, (pr - 1) & 63);
                        else cA = (int)__builtin_amdgcn_readfirstlane((int)c0p_out[pr - 1]);
#pragma unroll
                        for (int q = 0; q < ND; ++q) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(acc[q]) : "v"(hA[q]));
                        c0m = max(c0m, cA);
                    }
                }
            }
            row_tail(r0, c0m, i, ri);
            continue;
        }
        if (!(r0 & (VC_RF_PLAIN << 8))) {
            if (!(r0 & (VC_RF_PREV << 8))) {
                c0m = VC_INT_MIN;
#pragma unroll
                for (int q = 0; q < ND; ++q) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[q]) : "s"(0x80008000u));
            }
            const uint32_t nq = (r0 >> 16) & 0xFF;
            {
                // every listed predecessor sits in the LDS ring: straight-line, one test per further in-edge
                if (nq) {
                    const uint32_t r1 = __builtin_amdgcn_readlane(myrec.y, ri);
                    ring_merge(i, r1 & 0xFFFF, c0m);
                    if (nq > 1) {
                        ring_merge(i, r1 >> 16, c0m);
                        if (nq > 2) {
                            const uint32_t r2 = __builtin_amdgcn_readlane(myrec.z, ri);
                            ring_merge(i, r2 & 0xFFFF, c0m);
                            if (nq > 3) {
                                ring_merge(i, r2 >> 16, c0m);
                                if (nq > 4) {
                                    const uint32_t r3 = __builtin_amdgcn_readlane(myrec.w, ri);
                                    ring_merge(i, r3 & 0xFFFF, c0m);
                                    if (nq > 5) ring_merge(i, r3 >> 16, c0m);
                                }
                            }
                        }
                    }
                }
            }
        }
        row_tail(r0, c0m, i, ri);
      }
      if ((uint32_t)lane < cnt) c0p_out[i0 - 1 + lane] = (int16_t)c0vec;   // column 0 of the block just completed
      __threadfence_block();
    }
    if (lane == 0 && far_reads && !redo) atomicAdd(vc_stat_slot(a.stat) + 3, (unsigned long long)far_reads);
    if (!PIPE) {
        const unsigned long long dc = clock64() - clk_c0, dw = wall_clock64() - clk_w0;
        if (lane == 0) { unsigned long long* ck = vc_clk_slot(a.stat); atomicAdd(ck, dc); atomicAdd(ck + 1, dw); }
    }

    if (redo) return VC_FWD_DONE;                           // the end cell, ties and counters stand from the first pass
    // publish the end cell
    uint32_t end = 0;
    uint32_t outcome = VC_FWD_DONE;
    if (nw) {
        end = (best_row << 16) | len;
        if (ntie > 1 && (a.dp.flags[slot] & 2u)) {          // tie on a non-reference order: k_resolve decides
            if (lane == 0) { a.tie_cnt[job] = ntie; if (!PIPE) a.tie_list[atomicAdd(a.tie_n, 1u)] = slot; }
            outcome = VC_FWD_TIE;                           // (the persistent pipeline hands the window to its resolver wave instead of the list)
        }
    } else {
        const int gmax = wave_max_i32(best);
        if (gmax > 0) {
            const uint32_t rowc = (best == gmax) ? best_row : 0xFFFFFFFFu;
            const uint32_t rstar = wave_min_u32(rowc);
            // first column of that row holding the best score: re-read my part of the row
            __threadfence_block();
            uint32_t trow[ND];
            if (packed) {
                const uint32_t* hr = hrow0 + (uint64_t)(rstar - 1) * (NDS * 64) + lane * NDS;
                uint32_t wv[NDS];
#pragma unroll
                for (int t = 0; t < NDS; ++t) wv[t] = hr[t];
                vc_unpack_row<ND, NDS>(wv, trow);
            } else {
                const uint32_t* hr = hrow0 + (uint64_t)(rstar - 1) * (ND * 64);
#pragma unroll
                for (int q = 0; q < ND; ++q) trow[q] = hr[q * 64 + lane];
            }
            uint32_t firstc = 0xFFFFFFFFu;
#pragma unroll
            for (int q = ND - 1; q >= 0; --q) {
                const uint32_t hv = pk_sub(trow[q], njg[q]);
                const uint32_t c1 = lane * CPL + 2 * q + 1, c0i = c1 - 1;
                if (c1 < len && pk_hi(hv) == gmax) firstc = c1;
                if (c0i < len && pk_lo(hv) == gmax) firstc = c0i;
            }
            const uint32_t cstar = wave_min_u32(firstc);
            end = (rstar << 16) | (cstar + 1);
        }
    }
    if (lane == 0) a.job_end[job] = end;
    return outcome;
}

// the job of this workgroup in a lock-step launch: workgroup index, or the entry of the redo list
__device__ __forceinline__ bool vc_fwd_pick(const VcFwdArgs& a, VcJob& jb) {
    jb.redo = a.redo_list != nullptr;
    if (jb.redo && blockIdx.x >= *a.redo_n) return false;
    jb.job = jb.redo ? a.redo_list[blockIdx.x] : blockIdx.x;
    jb.slot = jb.job / a.group;
    if (jb.slot >= a.nslots) return false;
    jb.k = a.k0 + jb.job % a.group;
    if (a.cursor) {                                         // every window at its own layer
        const uint32_t cv = a.cursor[jb.slot];
        jb.k = cv & 0xFFFFu;
        jb.redo = jb.redo || (cv >> 31) != 0;
    }
    return true;
}

// NWONLY: the launch holds global alignments only (every build-phase launch; re-alignment launches whose layers are all
// full-span -- the host knows).  The kernel then carries no local-alignment body: 66 instead of 83 VGPRs at 10 cells per lane,
// which is what lets a backtrack wave (96) sit beside five forward waves on a SIMD.
template <int CPL, int RING, bool PACKED, bool KEPT, bool NWONLY>
__device__ __forceinline__ void vc_fwd_any(const VcFwdArgs& a, uint32_t* ring_raw, const VcJob& jb) {
    // alignment type of this job (uniform per wave)
    bool nw = a.mode == 0;
    const uint32_t w = a.w0 + jb.slot;
    if (a.mode == 1) {
        const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
        if (jb.k < ns) {
            const uint32_t L = (uint32_t)(a.b.seq_off[s0 + 1] - a.b.seq_off[s0]);
            nw = (jb.k == 0) || vc_full_span(a.b.seq_begin[s0 + jb.k], a.b.seq_end[s0 + jb.k], L);
        } else if (NWONLY) nw = true;                         // no such sequence in this window: the body resets the job's outputs and leaves
    }
    if (nw) (void)vc_fwd_body<CPL, RING, true, PACKED, KEPT>(a, ring_raw, jb);
    else if (!NWONLY) (void)vc_fwd_body<CPL, RING, false, PACKED, KEPT>(a, ring_raw, jb);
    else if (vc_lane() == 0) vc_fail(a.b, w, VC_WIN_INVALID, 28, a.k0);     // the host promised global alignments only: say so, do not skip
}

// CA <= CB: the two adjacent width classes of a batch share one launch (register and LDS budget of the
// wider one); each alignment takes the narrowest body that holds its sequence.  CA == CB: single class.
#ifndef VC_FWD_OCC
#define VC_FWD_OCC            // development: e.g. -DVC_FWD_OCC='__attribute__((amdgpu_waves_per_eu(4,4)))' caps the forward kernel's waves per SIMD
#endif
template <int CA, int CB, int RING, bool PACKED, bool KEPT, bool NWONLY>
VC_KL __global__ __launch_bounds__(64) VC_FWD_OCC void k_fwd(VcFwdArgs a) {
    __shared__ uint32_t ring_raw[RING * (CB / 2) * 64];
#ifdef VC_FWD_VGPR_PAD
    asm volatile("; keep the register allocation at 104: four forward waves per SIMD leave LDS and registers to the other kernels" ::: "v103");
#endif
    VcJob jb;
    if (!vc_fwd_pick(a, jb)) return;
    if (CA != CB) {
        // sequence length of this job decides the body (uniform per wave)
        const uint32_t w = a.w0 + jb.slot;
        const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
        uint32_t cls = CB;
        if (jb.k < ns) cls = vc_cpl_for((uint32_t)(a.b.seq_off[s0 + jb.k + 1] - a.b.seq_off[s0 + jb.k]));
        if ((cls == (uint32_t)CA || (a.fold && cls < (uint32_t)CA))) { vc_fwd_any<CA, RING, PACKED, KEPT, NWONLY>(a, ring_raw, jb); return; }
    }
    vc_fwd_any<CB, RING, PACKED, KEPT, NWONLY>(a, ring_raw, jb);
}

#include "vc_fwd_dt.h"

// ------------------------------------------------------------------------------------------------
// k_fwd_wide: the general forward pass for alignments the packed-int16 kernel declines -- score range beyond int16
// (the reference switches to 32-bit lanes there, simd_alignment_engine_implementation.hpp:699-706), sequences longer
// than 64 lanes x 32 columns, unusual score signs.  Same recurrence (sisd_alignment_engine.cpp:118-254, 292-360), int32,
// no assumption beyond what the reference makes: columns are processed in tiles of 512 (64 lanes x 8 cells), one full
// sweep over the rows per tile, predecessor rows re-read from the stored matrix (raw int32, tilted like k_fwd's:
// T[i][j] = H[i][j] - j*g), the row directly above kept in registers, row records fetched a block of 64 ahead, the lane scan
// on DPP.  It runs only for jobs k_fwd left untouched (job_type still 255).
// ------------------------------------------------------------------------------------------------
#define VC_WIDE_CPL 8
#define VC_WIDE_RING 8          // rows of the current tile kept in LDS (16 KB per wave; this kernel never fills a CU)
#define VC_WIDE_NEG (-(1 << 29))
VC_KL __global__ __launch_bounds__(64) void k_fwd_wide(VcFwdArgs a, int* wmat, uint64_t wstride, uint32_t wcols, int* c0w) {
    // the last VC_WIDE_RING rows of the tile in LDS (slot = row % VC_WIDE_RING): the predecessors that are not the row directly
    // above are nearly always among them, and a read back from the stored matrix costs a fence and a memory round trip
    __shared__ int wring[VC_WIDE_RING][VC_WIDE_CPL][64];
    const int lane = vc_lane();
    const uint32_t job = blockIdx.x;
    const uint32_t slot = job / a.group;
    if (slot >= a.nslots) return;
    const uint32_t k = a.cursor ? (a.cursor[slot] & 0xFFFFu) : a.k0 + job % a.group;
    const uint32_t w = a.w0 + slot;
    if (a.job_type[job] != 255) return;                      // k_fwd took it
    if (a.b.status[w] != VC_WIN_OK) return;
    const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
    if (k >= ns) return;
    const uint64_t so = a.b.seq_off[s0 + k];
    const uint32_t len = (uint32_t)(a.b.seq_off[s0 + k + 1] - so);
    const uint32_t L = (uint32_t)(a.b.seq_off[s0 + 1] - a.b.seq_off[s0]);
    bool nw = a.mode == 0;
    if (a.mode == 1) nw = (k == 0) || vc_full_span(a.b.seq_begin[s0 + k], a.b.seq_end[s0 + k], L);
    const int m = nw ? a.m : a.sm, n = nw ? a.n : a.sn, g = nw ? a.g : a.sg;
    const uint32_t nrows = a.dp.nrows[slot];
    if (len == 0 || nrows == 0 || (a.dp.flags[slot] & 1u)) return;                 // k_fwd reported it
    if (len > wcols || (uint64_t)nrows * wcols > wstride) { if (lane == 0) vc_fail(a.b, w, VC_WIN_OVERFLOW, 26, nrows); return; }
    const uint64_t nb = (uint64_t)slot * a.NC, eb = (uint64_t)slot * a.EC;
    if (lane == 0) {
        a.job_type[job] = nw ? 3 : 2;
        unsigned long long* st = vc_stat_slot(a.stat);
        atomicAdd(st + 0, (unsigned long long)nrows * len);
        atomicAdd(st + 1, (unsigned long long)nrows);
    }
    int* const T = wmat + (uint64_t)job * wstride;
    int* const c0out = c0w + (uint64_t)job * a.NC;
    constexpr int C = VC_WIDE_CPL;

    int best = nw ? VC_INT_MIN : 0;                            // NW: uniform; SW: per lane
    uint32_t best_row = 0, best_col = 0, ntie = 0;
    const uint32_t ntile = (len + 64 * C - 1) / (64 * C);
    for (uint32_t tile = 0; tile < ntile; ++tile) {
        const uint32_t cb = tile * 64 * C + lane * C;          // 0-based index of my first column (column j = cb + 1 + q)
        uint32_t sb[C];
#pragma unroll
        for (int q = 0; q < C; ++q) sb[q] = cb + q < len ? (uint32_t)a.b.bases[so + cb + q] : 0x100u;
        int prevT[C], prevLeft = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) prevT[q] = 0;
        int c0prev = 0, c0vec = 0;
        // row records a block of 64 ahead, lane t holding the record of row (block * 64 + t + 1), as in k_fwd: a load per row
        // would put a memory round trip on every row's critical path
        uint4 blkrec = make_uint4(0, 0, 0, 0), nextrec = make_uint4(0, 0, 0, 0);
        int blkedge = 0, prvedge = 0;                          // left edge of the tile for the rows of this block / of the block before
        if ((uint32_t)lane < nrows) nextrec = a.dp.rec[nb + lane];
        for (uint32_t i = 1; i <= nrows; ++i) {
            const uint32_t ri = (i - 1) & 63u;
            if (ri == 0) {
                blkrec = nextrec;
                const uint32_t r = i - 1 + 64 + (uint32_t)lane;
                if (r < nrows) nextrec = a.dp.rec[nb + r];
                // what enters this tile from the left, for the 64 rows of the block at once (the previous sweep wrote it)
                prvedge = blkedge;
                if (tile && i - 1 + (uint32_t)lane < nrows) blkedge = T[(uint64_t)(i - 1 + (uint32_t)lane) * wcols + tile * 64 * C - 1];
            }
            const uint4 rec = make_uint4((uint32_t)__builtin_amdgcn_readlane((int)blkrec.x, ri), (uint32_t)__builtin_amdgcn_readlane((int)blkrec.y, ri),
                                         (uint32_t)__builtin_amdgcn_readlane((int)blkrec.z, ri), (uint32_t)__builtin_amdgcn_readlane((int)blkrec.w, ri));
            const uint32_t x = rec.x & 0xFF, fl = (rec.x >> 8) & 0xFF;
            const bool isovf = (fl & VC_RF_OVF) != 0;
            const uint32_t np = isovf ? rec.z : ((rec.x >> 16) & 0xFF);
            int accd[C], accv[C];
#pragma unroll
            for (int q = 0; q < C; ++q) { accd[q] = VC_WIDE_NEG; accv[q] = VC_WIDE_NEG; }
            int c0m = VC_INT_MIN;
            for (uint32_t p = 0; p < np; ++p) {
                uint32_t delta;
                if (isovf) { delta = a.dp.ovf[eb + rec.y + p]; if (a.kept && (delta & 0x8000u)) delta &= 0x7Fu; }   // kept-row ring: forward form of the entry
                else { const uint32_t wsel = p < 2 ? rec.y : (p < 4 ? rec.z : rec.w); delta = (p & 1) ? (wsel >> 16) : (wsel & 0xFFFF); }
                const uint32_t pr = i - delta;
                int hv[C], hl, c0p;
                if (pr == 0) {                                 // virtual row: H[0][j] = j*g (NW) / 0 (SW)
#pragma unroll
                    for (int q = 0; q < C; ++q) hv[q] = nw ? 0 : -(int)(cb + 1 + q) * g;
                    hl = nw ? 0 : -(int)cb * g;
                    c0p = 0;
                } else if (delta == 1) {
#pragma unroll
                    for (int q = 0; q < C; ++q) hv[q] = prevT[q];
                    hl = prevLeft; c0p = c0prev;
                } else if (delta <= (uint32_t)VC_WIDE_RING) {
                    const int (*wr)[64] = wring[pr & (VC_WIDE_RING - 1)];
#pragma unroll
                    for (int q = 0; q < C; ++q) hv[q] = wr[q][lane];
                    c0p = __builtin_amdgcn_readlane(c0vec, (pr - 1) & 63);
                    int edge = nw ? c0p : 0;                   // column 0 of that row
                    if (tile) edge = (pr - 1 >= ((i - 1) & ~63u)) ? __builtin_amdgcn_readlane(blkedge, (pr - 1) & 63) : __builtin_amdgcn_readlane(prvedge, (pr - 1) & 63);
                    hl = VC_DPP_SHR(hv[C - 1], edge, 0x138, 0xF);
                } else {
                    __threadfence_block();                     // a row read back: my own earlier stores must have landed (only here, not per row)
                    const int* hr = T + (uint64_t)(pr - 1) * wcols + cb;
#pragma unroll
                    for (int q = 0; q < C; ++q) hv[q] = hr[q];
                    if (delta <= 64) c0p = __builtin_amdgcn_readlane(c0vec, (pr - 1) & 63);
                    else c0p = (int)__builtin_amdgcn_readfirstlane(c0out[pr - 1]);
                    int edge = nw ? c0p : 0;                   // column 0 of that row
                    if (tile) edge = (int)__builtin_amdgcn_readfirstlane(T[(uint64_t)(pr - 1) * wcols + tile * 64 * C - 1]);
                    hl = VC_DPP_SHR(hv[C - 1], edge, 0x138, 0xF);          // the left lane's last cell; lane 0 takes the edge
                }
#pragma unroll
                for (int q = 0; q < C; ++q) { accv[q] = max(accv[q], hv[q]); accd[q] = max(accd[q], q ? hv[q - 1] : hl); }
                c0m = max(c0m, c0p);
            }
            const int col0 = nw ? c0m + g : 0;
            int P[C];
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const int j = (int)(cb + 1 + q);
                int v = max(accd[q] + ((sb[q] == x ? m : n) - g), accv[q] + g);
                if (!nw) v = max(v, -j * g);                  // SW floor H >= 0
                P[q] = cb + q < len ? v : VC_WIDE_NEG;
            }
            // horizontal pass: prefix maximum with the value entering from the left of the tile
            const int carry_in = tile ? __builtin_amdgcn_readlane(blkedge, ri) : col0;
#pragma unroll
            for (int q = 1; q < C; ++q) P[q] = max(P[q], P[q - 1]);
            int sc = P[C - 1];
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x111, 0xF));
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x112, 0xF));
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x114, 0xF));
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x118, 0xF));
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x142, 0xA));
            sc = max(sc, VC_DPP_SHR(sc, VC_INT_MIN, 0x143, 0xC));
            const int carry = max(VC_DPP_SHR(sc, VC_INT_MIN, 0x138, 0xF), carry_in);     // lane 0: what enters from the left of the tile
            int cur[C];
#pragma unroll
            for (int q = 0; q < C; ++q) cur[q] = max(P[q], carry);
            {
                int* hr = T + (uint64_t)(i - 1) * wcols + cb;
#pragma unroll
                for (int q = 0; q < C; ++q) hr[q] = cur[q];
                // one wave per workgroup: its LDS operations retire in order, no barrier -- only keep the compiler from reordering
                __builtin_amdgcn_wave_barrier();
                int (*ww)[64] = wring[i & (VC_WIDE_RING - 1)];
#pragma unroll
                for (int q = 0; q < C; ++q) ww[q][lane] = cur[q];
                __builtin_amdgcn_wave_barrier();
            }
            // end cell
            if (nw) {
                if ((fl & VC_RF_SINK) && tile + 1 == ntile) {   // sisd :353-355
                    const uint32_t le = ((len - 1) % (64 * C)) / C, ce = (len - 1) % C;
                    int v = cur[0];
#pragma unroll
                    for (int q = 1; q < C; ++q) v = ce == (uint32_t)q ? cur[q] : v;
                    v = __shfl(v, (int)le, 64);
                    if (v > best) {
                        best = v; best_row = i; ntie = 1;
                        if (lane == 0) a.tie_rows[(uint64_t)job * VC_MAXTIE] = (uint16_t)i;
                    } else if (v == best) {
                        if (lane == 0) {
                            if (ntie < VC_MAXTIE) a.tie_rows[(uint64_t)job * VC_MAXTIE + ntie] = (uint16_t)i;
                            else if (a.tie_over && ntie < a.tie_over_stride) a.tie_over[(uint64_t)job * a.tie_over_stride + ntie] = i;
                        }
                        ntie++;
                    }
                }
            } else {
                // sisd :350-352: the first cell in (row, column) order among those with the best score
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    if (cb + q < len) {
                        const int h = cur[q] + (int)(cb + 1 + q) * g;
                        if (h > best || (h == best && h > 0 && i < best_row)) { best = h; best_row = i; best_col = cb + 1 + q; }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < C; ++q) prevT[q] = cur[q];
            prevLeft = VC_DPP_SHR(cur[C - 1], carry_in, 0x138, 0xF);
            // bookkeeping of column 0 like k_fwd: the last 64 rows in a register, the rest in memory
            c0prev = col0;
            c0vec = ((uint32_t)lane == ((i - 1) & 63)) ? col0 : c0vec;
            if ((i & 63) == 0 || i == nrows) {
                const uint32_t first = (i - 1) & ~63u;
                if (first + lane < i) c0out[first + lane] = c0vec;
            }
        }
        __threadfence_block();                                 // the next tile reads this tile's last column of every row
    }
    uint32_t end = 0;
    if (nw) {
        end = (best_row << 16) | len;
        if (ntie > 1 && (a.dp.flags[slot] & 2u) && lane == 0) { a.tie_cnt[job] = ntie; a.tie_list[atomicAdd(a.tie_n, 1u)] = slot; }
    } else {
        const int gmax = wave_max_i32(best);
        if (gmax > 0) {
            const uint32_t rowc = (best == gmax) ? best_row : 0xFFFFFFFFu;
            const uint32_t rstar = wave_min_u32(rowc);
            const uint32_t colc = (best == gmax && best_row == rstar) ? best_col : 0xFFFFFFFFu;
            // a lane keeps one (row, column) per value, the smallest row; the smallest column of THAT row may sit in
            // another lane or tile, so look the row up again
            uint32_t cstar = wave_min_u32(colc);
            for (uint32_t c = lane; c < len; c += 64) {
                const int h = T[(uint64_t)(rstar - 1) * wcols + c] + (int)(c + 1) * g;
                if (h == gmax) cstar = min(cstar, c + 1);
            }
            cstar = wave_min_u32(cstar);
            end = (rstar << 16) | cstar;
        }
    }
    if (lane == 0) a.job_end[job] = end;
}

// ------------------------------------------------------------------------------------------------
// k_trace: backtrack, one alignment per thread, straight from the stored matrix like sisd_alignment_engine.cpp:362-459:
// diagonal over the in-edges in list order, then vertical over the in-edges in list order, then
// horizontal.  Pairs are emitted tail-first as (row << 16) | column, 0 meaning "-1"; consumers read
// them back to front.
// ------------------------------------------------------------------------------------------------
struct VcTraceArgs {
    VcBatchDev b;
    VcDp dp;
    uint32_t w0, nslots, NC, EC, group, cpl;
    int m, n, g, sm, sn, sg;
    const uint32_t* hmat; uint64_t hstride;
    const int16_t* c0;
    const uint32_t* job_end;
    const uint8_t* job_type;
    uint32_t* pairs;          // [pair_jobs * PC]
    uint32_t* npairs;         // [pair_jobs]
    uint32_t PC;
    uint32_t pair_group, pair_k0;   // pairs index = slot*pair_group + (k - pair_k0)
    uint32_t k0;
    unsigned long long* stat;   // [VC_STAT_SLOTS][8], see vc_ctx::d_stat
    const int* wmat; uint64_t wstride; uint32_t wcols; const int* c0w;   // matrices of k_fwd_wide (job types 2, 3)
    int packed;                 // stored row form of this launch's k_fwd (byte-packed rows or raw int16 pairs)
    uint32_t kept;              // != 0: long predecessor lists (VcDp::ovf) are in the forward form of the kept-row ring (vc_frec_kept)
    const uint32_t* bmat; const uint32_t* band_par; int band;     // banded matrix store of this launch's k_fwd (see vc_band_start)
    const uint32_t* redo_list; const uint32_t* redo_n;            // != nullptr: walk the listed jobs (their matrices were stored whole again)
    uint32_t* redo_out; uint32_t* redo_out_n;                     // band != 0: jobs whose walk needed a cell outside the band
    int only_wide;              // k_trace: take only those jobs (k_tracew walked the rest)
    int shared_table;           // k_tracew: the VC_TG alignments of a wave share a window (group % VC_TG == 0)
    uint32_t tab_rows;          // k_tracew: rows the LDS table is sized for (>= every graph's height in this launch)
    uint32_t cpl_lo;            // narrowest width class the forward pass of this launch used (0: every sequence in its own class)
    int band_chain;             // the forward pass of this launch stored the narrow band of the re-alignment rounds (vc_band_lanes(cpl, true))
    uint32_t* cursor;           // build phase, != nullptr: [nslots] layer of every window | "left the band" << 31 (VcFwdArgs::cursor); the walk
                                //   sets / clears the flag, the pair list of a window is pairs + slot * PC
};

// One alignment per THREAD: the walk is a chain of dependent lookups, so the instruction cost is shared
// by 64 alignments per wave and the kernel is a latency chain that overlaps the forward kernel of another
// stream (wave-per-alignment and LDS-tiled variants were measured slower end to end: they take issue
// slots and CUs away from k_fwd).  Loads stop at the first matching move, like the reference's scan.
#define VC_TRACE_LANES 8     // alignments per wave: lanes walk in lockstep, so fewer per wave = less waiting on the slowest
VC_KL __global__ void k_trace(VcTraceArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    if (threadIdx.x >= VC_TRACE_LANES) return;
    const uint32_t job = blockIdx.x * VC_TRACE_LANES + threadIdx.x;
    if (job >= a.nslots * a.group) return;
    const uint32_t slot = job / a.group, k = a.cursor ? (a.cursor[slot] & 0xFFFFu) : a.k0 + job % a.group;
    const uint32_t w = a.w0 + slot;
    const uint64_t pj = a.cursor ? (uint64_t)slot : (uint64_t)slot * a.pair_group + (k - a.pair_k0);
    const uint8_t type_raw = a.job_type[job];
    if (type_raw == 255) return;
    const bool dtj = (type_raw & VC_JOB_DT) != 0;             // doubly tilted rows (k_fwd_dt): see vc_dt_cell
    const uint8_t type = (uint8_t)(type_raw & ~VC_JOB_DT);
    const bool wide = type == 2 || type == 3;                 // (VC_JOB_DT rows: k_tracew's, or converted below)
    if (a.only_wide && !wide) return;
    if (a.b.status[w] != VC_WIN_OK) return;
    uint32_t* out = a.pairs + pj * a.PC;
    const uint32_t end = a.job_end[job];
    uint32_t i = end >> 16, j = end & 0xFFFF;
    const uint64_t nb = (uint64_t)slot * a.NC, eb = (uint64_t)slot * a.EC;
    const bool nw = (type & 1) != 0;
    const int m = nw ? a.m : a.sm, n = nw ? a.n : a.sn, g = nw ? a.g : a.sg;
    const uint64_t so = a.b.seq_off[a.b.win_seq_off[w] + k];
    const int* wm = wide ? a.wmat + (uint64_t)job * a.wstride : nullptr;
    const int* wc0 = wide ? a.c0w + (uint64_t)job * a.NC : nullptr;
    const uint32_t* hm32 = a.hmat + (uint64_t)job * a.hstride;
    const uint16_t* hm = (const uint16_t*)hm32;
    const int16_t* c0 = a.c0 + (uint64_t)job * a.NC;
    const uint32_t cpl = max(vc_cpl_for((uint32_t)(a.b.seq_off[a.b.win_seq_off[w] + k + 1] - so)), a.cpl_lo), nd = cpl / 2, nds = (uint32_t)vc_nds((int)cpl);
    const bool packed = a.packed != 0;
    // k_fwd stores the tilted matrix T[r][col] = H[r][col] - col*g; the tests of sisd :392-448 become
    // diagonal T == T' + (score - g), vertical T == T' + g, horizontal T == T', SW stop T == -col*g
    auto Hat = [&](uint32_t r, uint32_t col) -> int {     // T[r][col] incl. the virtual row 0 / column 0
        if (r == 0) return nw ? 0 : -(int)col * g;
        if (wide) return col == 0 ? (nw ? wc0[r - 1] : 0) : wm[(uint64_t)(r - 1) * a.wcols + col - 1];
        if (col == 0) return nw ? (int)c0[r - 1] : 0;
        const uint32_t ci = col - 1, lc = ci / cpl, cc = ci % cpl;
        if (packed) { const int v = vc_packed_cell(hm32 + (uint64_t)(r - 1) * nds * 64 + lc * nds, cc, cpl); return dtj ? vc_dt_cell(v, r, g) : v; }
        return (int)(short)hm[((uint64_t)(r - 1) * nd * 64 + (cc >> 1) * 64 + lc) * 2 + (cc & 1)];
    };
    uint32_t nout = 0;
    bool ovf = false, broken = false;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    if (end != 0) {
        int Hij = Hat(i, j);
        uint4 rec = i ? a.dp.rec[nb + i - 1] : zero4;
        for (;;) {
            if (nw) { if (i == 0 && j == 0) break; }
            else if (Hij == -(int)j * g) break;
            uint32_t pi_ = 0, pj_ = 0;
            int hv = 0;
            uint4 nrec = zero4;
            bool found = false, have_nrec = false;
            if (i != 0) {
                const bool isovf = ((rec.x >> 8) & VC_RF_OVF) != 0;
                const uint32_t np = isovf ? rec.z : ((rec.x >> 16) & 0xFF);
                auto delta_of = [&](uint32_t p) -> uint32_t {
                    if (isovf) { const uint32_t e = a.dp.ovf[eb + rec.y + p]; return (a.kept && (e & 0x8000u)) ? (e & 0x7Fu) : e; }
                    const uint32_t wsel = p < 2 ? rec.y : (p < 4 ? rec.z : rec.w);
                    return (p & 1) ? (wsel >> 16) : (wsel & 0xFFFF);
                };
                if (j != 0) {
                    // by far the most frequent move is the diagonal through the first in-edge: issue its
                    // three loads (cell, base, next row's record) together -- one round trip per step
                    const uint32_t pr0 = i - (isovf ? delta_of(0) : (rec.y & 0xFFFF));
                    // unconditional, branch-free addresses so the three loads are in flight together
                    const uint32_t rr = pr0 ? pr0 : 1, cc1 = j > 1 ? j - 2 : 0, lc = cc1 / cpl, cw = cc1 % cpl;
                    const int v0raw = packed ? vc_packed_cell(hm32 + (uint64_t)(rr - 1) * nds * 64 + lc * nds, cw, cpl)
                                             : (int)(short)hm[((uint64_t)(rr - 1) * nd * 64 + (cw >> 1) * 64 + lc) * 2 + (cw & 1)];
                    const uint32_t bs = a.b.bases[so + j - 1];
                    const uint4 q0 = a.dp.rec[nb + rr - 1];
                    int v0 = (dtj && packed) ? vc_dt_cell(v0raw, rr, g) : v0raw;
                    if (pr0 == 0 || j == 1 || wide) v0 = Hat(pr0, j - 1);
                    const int sc = ((bs == (rec.x & 0xFF)) ? m : n) - g;
                    if (Hij == v0 + sc) { pi_ = pr0; pj_ = j - 1; hv = v0; nrec = pr0 ? q0 : zero4; have_nrec = true; found = true; }
                    for (uint32_t p = 1; p < np && !found; ++p) {
                        const uint32_t pr = i - delta_of(p);
                        const int v = Hat(pr, j - 1);
                        if (Hij == v + sc) { pi_ = pr; pj_ = j - 1; hv = v; found = true; }
                    }
                }
                if (!found) {
                    for (uint32_t p = 0; p < np; ++p) {
                        const uint32_t pr = i - delta_of(p);
                        const int v = Hat(pr, j);
                        if (Hij == v + g) { pi_ = pr; pj_ = j; hv = v; found = true; break; }
                    }
                }
            }
            if (!found && j != 0) {
                const int v = Hat(i, j - 1);
                if (Hij == v) { pi_ = i; pj_ = j - 1; hv = v; nrec = rec; have_nrec = true; found = true; }
            }
            if (!found) { broken = true; break; }
            if (nout >= a.PC) { ovf = true; break; }
            out[nout++] = ((i == pi_ ? 0u : i) << 16) | (j == pj_ ? 0u : j);
            if (!have_nrec) nrec = pi_ ? a.dp.rec[nb + pi_ - 1] : zero4;
            i = pi_; j = pj_; Hij = hv; rec = nrec;
        }
    }
    if (broken) { vc_fail(a.b, w, VC_WIN_INVALID, 17, i); nout = 0; }
    if (ovf) { vc_fail(a.b, w, VC_WIN_OVERFLOW, 5, nout); nout = 0; }
    a.npairs[pj] = nout;
}

// ------------------------------------------------------------------------------------------------
// k_tracew: the same backtrack, cooperative: VC_TG alignments per wave, VC_TL = 16 lanes each.  The walk
// is a chain of dependent HBM round trips, and ~85 % of its moves are "diagonal through the first
// in-edge".  Each round therefore
//   A. follows first in-edges for up to VC_SPECW positions using a per-graph table in LDS (no HBM; 4-bit entries, see
//      vc_tracew_tab_len),
//   B. lets lane k of the group fetch the diagonal cell (and the row record) of speculated position k --
//      one round trip for all of them,
//   C. accepts the longest prefix whose cells confirm the move (exactly the reference's first test at
//      each of those cells, so nothing is skipped), and
//   D. takes one fully general step at the first position that did not confirm: lanes 0..6 of the group
//      test the diagonal through in-edge p, lanes 8..14 the vertical one, lane 15 the horizontal move,
//      all in one round trip; ballots pick the first match in the reference's order (sisd :392-448).
// Four alignments share every instruction of the round.  Measured with the cycle counter: B's loads (HBM misses, one
// 128-B line per cell) are ~75 % of a round, D's mostly hit the lines B brought in (~7 %).  A step taken locally for
// single-in-edge rows (vertical / horizontal cells fetched in B) was tried and is slower: a wave still runs D when any of
// its four groups needs it, and the extra scattered loads lengthen B.
// ------------------------------------------------------------------------------------------------
#define VC_TG 4
#ifndef VC_SPECW
#define VC_SPECW 8         // positions speculated per round: 6..10 measured equal and 5 % better than 16 (fewer lines fetched for moves that get rejected)
#endif
#define VC_TL 16
// first-in-edge table of k_tracew: one 4-bit entry per row (distance to the row of the first in-edge; 0: do not speculate --
// also for distances beyond 15, which the general step then takes).  Half a byte instead of a byte per row: what the job is
// short of is LDS x time (k_fwd alone fills the LDS of every CU; a backtrack wave that waits on memory with 9 KB of tables
// keeps a forward wave out), and distances beyond 15 are rare
__host__ __device__ inline uint32_t vc_tracew_tab_len(uint32_t max_rows) { return (max_rows + 2 + 7) & ~7u; }     // entries per table
// Shared table (the alignments of a wave belong to one window: re-alignment rounds): the whole graph, 4 bits per row.  Otherwise (build phase:
// every alignment of a wave walks a window of its own) each group keeps a SLIDING window of VC_TW_WIN rows around its position -- 128 bytes
// instead of a table of the graph's height (1.1 KB at 2 240 rows; 9 KB per wave of eight alignments, which is what kept a second
// backtrack wave off a CU whose LDS five forward waves per SIMD fill to 10 KB): the walk only ever looks at the rows just below it.
#define VC_TW_WIN 256u       // rows of the sliding window (four blocks of 64; a power of two)
__host__ __device__ inline uint32_t vc_tracew_lds_bytes(uint32_t max_rows, bool shared_table, uint32_t tg = VC_TG) { return shared_table ? vc_tracew_tab_len(max_rows) / 2u : tg * (VC_TW_WIN / 2u); }
__device__ __forceinline__ int vc_row_shr1(int v, int first) {          // value of the lane to the left inside a 16-lane row
    return __builtin_amdgcn_update_dpp(first, v, 0x111, 0xF, 0xF, false);
}

// The walk of the VC_TG alignments of one wave.  Group g (lanes 16 g .. 16 g + 15) walks alignment `job` of window `slot`,
// sequence k, pair list pj; `redo`: its matrix was stored whole (no band).  Lock-step launches take these from the workgroup
// index (k_tracew), the persistent build pipeline (vc_pipe.h) from its work queue -- there the groups of a wave hold alignments
// of different windows AND different layers.  Returns (per lane of the group) whether the alignment left the band; PIPE: the
// caller puts it on its own redo queue instead of the launch's redo list.
template <bool PIPE, int TL = VC_TL>
__device__ __forceinline__ bool vc_tracew_body(const VcTraceArgs& a, uint8_t* smem, uint32_t job, const uint32_t slot, const uint32_t k,
                                               const uint64_t pj, bool valid, const bool redo) {
    // TL lanes per alignment, TG = 64 / TL alignments per wave.  16 x 4: the general step looks at seven in-edges x {diagonal,
    // vertical} + the horizontal move in one round trip.  8 x 8: three in-edges x {diagonal, vertical} + horizontal per round trip
    // (in-degrees beyond three take another pass), the eight speculated positions of a round fill the group exactly, and a
    // wave-instruction serves twice the alignments -- what the lock-step launches use.
    static_assert(TL == 16 || TL == 8, "group width");
    constexpr int TG = 64 / TL;
    constexpr uint32_t ND_ = TL == 16 ? 7u : 3u;             // in-edges per pass of the general step; vertical candidates sit ND_ + 1 lanes up
    const int lane = vc_lane();
    const uint32_t grp = (uint32_t)lane / TL, gl = (uint32_t)lane % TL, gbase = grp * TL;
    // first in-edge distance of row r (0: do not speculate).  In the re-alignment rounds the alignments of a
    // wave belong to one window (group % TG == 0) and share one table: a quarter of the LDS, more waves
    const bool shared_tab = a.shared_table != 0;
    uint8_t* tab = smem + (shared_tab ? 0u : grp * (VC_TW_WIN / 2u));                           // two entries per byte
    uint32_t* const tab32 = reinterpret_cast<uint32_t*>(tab);
    uint32_t wlo = 0;                                          // sliding window: its lowest row (a multiple of 64); rows [wlo, wlo + VC_TW_WIN) sit at row % VC_TW_WIN
    auto tab_at = [&](uint32_t r) __attribute__((always_inline)) -> uint32_t {
        if (shared_tab) return r <= a.tab_rows ? ((uint32_t)tab[r >> 1] >> ((r & 1u) * 4u)) & 15u : 0u;
        return r >= wlo ? (tab32[(r & (VC_TW_WIN - 1u)) >> 3] >> ((r & 7u) * 4u)) & 15u : 0u;      // below the window: not speculated (the window follows)
    };
    const uint32_t w = a.w0 + slot;
    const uint8_t type_raw = valid ? a.job_type[job] : (uint8_t)255;
    // VC_JOB_DT: the rows of this job are doubly tilted (k_fwd_dt, vc_fwd_dt.h) -- a cell read back is T'' = H - (row + col) * g as an unsigned
    // 16-bit number; Tat() below hands out T = T'' + row * g, so everything behind it is the same walk
    const bool dtj = type_raw != 255 && (type_raw & VC_JOB_DT) != 0;
    const uint8_t type = type_raw == 255 ? type_raw : (uint8_t)(type_raw & ~VC_JOB_DT);
    valid = valid && type < 2;                                // 255: nothing to walk; 2, 3: k_fwd_wide's, walked by k_trace
    if (valid && a.b.status[w] != VC_WIN_OK) valid = false;
    if (!__any(valid)) return false;
    uint32_t* out = a.pairs + pj * a.PC;
    const uint32_t end = valid ? a.job_end[job] : 0u;
    const uint64_t nb = (uint64_t)slot * a.NC, eb = (uint64_t)slot * a.EC;
    const bool nw = type == 1;
    const int m = nw ? a.m : a.sm, n = nw ? a.n : a.sn, g = nw ? a.g : a.sg;
    const uint32_t sq = a.b.win_seq_off[w] + (valid ? k : 0);
    const uint64_t so = a.b.seq_off[sq];
    const uint32_t* hm32 = a.hmat + (uint64_t)(valid ? job : 0) * a.hstride;
    const uint16_t* hm = (const uint16_t*)hm32;
    const int16_t* c0 = a.c0 + (uint64_t)(valid ? job : 0) * a.NC;
    const uint32_t cpl = max(vc_cpl_for((uint32_t)(a.b.seq_off[sq + 1] - so)), a.cpl_lo), nd = cpl / 2, nds = (uint32_t)vc_nds((int)cpl);
    const bool packed = a.packed != 0;
    const uint32_t band_lanes = vc_band_lanes(cpl, a.band_chain != 0);   // (the width class and the phase decide: 80 columns, at least 8 lanes; re-alignment rounds: 40, at least 4)
    // banded store: global alignments of a banded launch keep VC_BAND_LANES lanes per row around the rank diagonal
    const bool band = a.band != 0 && !redo && valid && type == 1;
    const uint32_t* bm32 = a.bmat + (uint64_t)(valid ? job : 0) * vc_band_job_dwords(a.hstride);
    const uint32_t band_ql = band ? a.band_par[job] : 0u;
    const uint32_t blk_dw = band_lanes * (a.packed != 0 ? nds : nd);   // dwords of a band row (byte-packed or raw int16 pairs)
    bool oob = false;                                          // this lane asked for a cell outside the band (its value is then meaningless)
    const uint32_t nrows = valid ? min(a.dp.nrows[slot], a.tab_rows) : 0;
    // stored matrix (tilted, see vc_fwd_body): diagonal T == T' + (score - g), vertical T == T' + g,
    // horizontal T == T', SW stop T == -col*g.  (The runtime division by cpl stays: a multiply in its place -- a multiply-high in
    // round 2, a 24-bit multiply and a shift in round 3 -- made the kernel 8-10 % SLOWER both times: 394 -> 427 ms per 32 768
    // windows; the compiler then orders the loads of a round differently.)
    auto Tat = [&](uint32_t r, uint32_t col) __attribute__((always_inline)) -> int {
        if (r == 0) return nw ? 0 : -(int)col * g;
        if (col == 0) return nw ? (int)c0[r - 1] : 0;
        const uint32_t ci = col - 1, lc = ci / cpl, cc = ci % cpl;
        if (band) {
            const uint32_t bl = lc - vc_band_row_start(r - 1, band_ql, band_lanes);
            if (bl >= band_lanes) { oob = true; return 0; }
            if (!packed) {                                       // raw band row: [band lane][nd dwords]
                const uint32_t wv = bm32[(r - 1) * blk_dw + bl * nd + (cc >> 1)];
                return (int)(short)((cc & 1u) ? (wv >> 16) : (wv & 0xFFFFu));
            }
            const int v = vc_packed_cell(bm32 + (r - 1) * blk_dw + bl * nds, cc, cpl);
            return dtj ? vc_dt_cell(v, r, g) : v;
        }
        if (packed) { const int v = vc_packed_cell(hm32 + (uint64_t)(r - 1) * nds * 64 + lc * nds, cc, cpl); return dtj ? vc_dt_cell(v, r, g) : v; }
        return (int)(short)hm[((uint64_t)(r - 1) * nd * 64 + (cc >> 1) * 64 + lc) * 2 + (cc & 1)];
    };
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto perm4 = [&](const uint4& v, uint32_t l) __attribute__((always_inline)) -> uint4 {
        return make_uint4((uint32_t)__shfl((int)v.x, (int)l, 64), (uint32_t)__shfl((int)v.y, (int)l, 64),
                          (uint32_t)__shfl((int)v.z, (int)l, 64), (uint32_t)__shfl((int)v.w, (int)l, 64));
    };
    auto gmask = [&](unsigned long long mm) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(mm >> gbase) & ((1u << TL) - 1u); };
    // value of the lane to the left inside the group; the group's first lane takes `first`
    auto shr1 = [&](int v, int first) __attribute__((always_inline)) -> int {
        const int x = vc_row_shr1(v, first);                  // (rows of 16 lanes: right for TL == 16, and for every lane but the first of an upper half-row)
        return (TL == 8 && gl == 0) ? first : x;
    };

    bool walking = valid && end != 0;
    bool gredo = false;                                       // this alignment goes on the redo list
    if (shared_tab) {
        // every lane of the wave has the same slot; the first valid lane's view of it is everybody's
        const unsigned long long vm = __ballot(valid);
        const int src = __ffsll((long long)vm) - 1;
        const uint32_t nr = (uint32_t)__shfl((int)nrows, src, 64);
        const uint32_t nb_lo = (uint32_t)__shfl((int)(uint32_t)nb, src, 64), nb_hi = (uint32_t)__shfl((int)(uint32_t)(nb >> 32), src, 64);
        const uint64_t nbs = ((uint64_t)nb_hi << 32) | nb_lo;
        // lane k packs the entries of rows 2k and 2k + 1 (row numbers; row 0 is the virtual row): one 16-bit load of the rows' entries
        // (VcDp::fie, written beside the row records) -- the 16-byte records themselves used to be read here, a quarter of the bytes the
        // backtrack fetched
        const uint8_t* fie = a.dp.fie + (uint64_t)(uint32_t)__shfl((int)slot, src, 64) * VC_FIE_STRIDE(a.NC);
        (void)nbs;
        auto pair = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t {
            if (2 * k > nr) return 0u;
            const uint32_t v = *reinterpret_cast<const uint16_t*>(fie + 2 * k);
            return (v & 0xFu) | ((2 * k + 1 <= nr ? (v >> 8) & 0xFu : 0u) << 4);
        };
        // (two blocks of entries per pass: loads in flight per lane instead of a chain)
        for (uint32_t k0 = lane; 2 * k0 <= nr; k0 += 2 * TG * TL) {
            const uint32_t k1 = k0 + TG * TL;
            const uint32_t p0 = pair(k0), p1 = pair(k1);
            tab[k0] = (uint8_t)p0;
            if (2 * k1 <= nr) tab[k1] = (uint8_t)p1;
        }
    }
    // sliding window (not shared): lane gl of the group fetches entries 8 gl .. 8 gl + 7 of a block of 64 rows (VcDp::fie, one byte per row
    // number) and squeezes them into eight nibbles = the block's dword gl
    // (the address is worked out where it is needed -- once per 64 rows walked: k_tracew sits at 110 VGPRs, and 112 is what fits beside five forward waves)
    auto blk_fetch = [&](uint32_t b, uint32_t& lo, uint32_t& hi) __attribute__((always_inline)) {
        const uint32_t r = 64u * b + 8u * gl;
        lo = 0; hi = 0;
        if (gl < 8u && r <= nrows) {                            // (eight lanes fill a block, whatever the group's width; the array is padded: the eight bytes of the last chunk exist)
            const uint8_t* const fp = a.dp.fie + (uint64_t)slot * VC_FIE_STRIDE(a.NC) + r;
            lo = *reinterpret_cast<const uint32_t*>(fp);
            hi = *reinterpret_cast<const uint32_t*>(fp + 4);
        }
    };
    auto blk_store = [&](uint32_t b, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
        auto sq = [](uint32_t x) { x = (x | (x >> 4)) & 0x00FF00FFu; return (x | (x >> 8)) & 0xFFFFu; };     // four bytes (each < 16) -> four nibbles
        const uint32_t r = 64u * b + 8u * gl;
        // rows beyond the graph: 0 (never looked up)
        uint32_t v = sq(lo) | (sq(hi) << 16);
        if (r + 7u > nrows) v &= (r > nrows) ? 0u : (0xFFFFFFFFu >> ((7u - (nrows - r)) * 4u));
        if (gl < 8u) tab32[(b & (VC_TW_WIN / 64u - 1u)) * 8u + gl] = v;
    };
    auto win_init = [&](uint32_t top) __attribute__((always_inline)) {           // the window ends at the block of row `top`
        const uint32_t bt = top >> 6, b0 = bt >= 3u ? bt - 3u : 0u;
        uint32_t lo[4], hi[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) { lo[u] = hi[u] = 0; if (b0 + u <= bt) blk_fetch(b0 + u, lo[u], hi[u]); }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) if (b0 + u <= bt) blk_store(b0 + u, lo[u], hi[u]);
        wlo = b0 * 64u;
    };
    if (!shared_tab && walking) win_init(end >> 16);
    __syncthreads();
    uint32_t gi = end >> 16, gj = end & 0xFFFF, gnout = 0, nspec_ok = 0, nrounds = 0;
    bool govf = false, gbroken = false;
    int gT = 0;
    uint4 grec = zero4;
    if (walking) {
        gT = Tat(gi, gj);
        if (gi) grec = a.dp.rec[nb + gi - 1];
        if (oob) { gredo = true; walking = false; }           // (cannot happen: the band ends on the end cell)
    }
    for (;;) {
        if (walking && (nw ? (gi == 0 && gj == 0) : (gT == -(int)gj * g))) walking = false;
        if (!__any(walking)) break;
        // sliding window of first-in-edge entries: a position below it (a general step may jump any distance) starts it over there; a position
        // in its lower half asks for the next block of 64 rows now -- the request travels with this round's loads and is stored behind them
        uint32_t pf_lo = 0, pf_hi = 0;                                  // (an entry is < 16: bit 31 of pf_hi says "a block is on its way")
        if (!shared_tab) {
            if (walking && gi < wlo) win_init(gi);
            if (walking && wlo > 0 && gi < wlo + VC_TW_WIN / 2u) { blk_fetch((wlo >> 6) - 1u, pf_lo, pf_hi); pf_hi |= 0x80000000u; }
        }
        // ---- A: speculated positions (row my_i, column gj - gl), next row my_in
        uint32_t my_i = 0, my_in = 0, nspec = 0;
        {
            uint32_t ci = gi;
            bool can = walking && gi != 0 && gj != 0;
#pragma unroll
            for (uint32_t t = 0; t < TL; t += 2) {                    // two links per iteration
                can = can && ci != 0 && gj > t && t < VC_SPECW;
                const uint32_t d1 = (can && ci <= a.tab_rows) ? tab_at(ci) : 0u;
                const uint32_t d2 = (can && d1 != 0 && ci - d1 != 0 && ci - d1 <= a.tab_rows) ? tab_at(ci - d1) : 0u;
                can = can && d1 != 0;
                const uint32_t c1 = ci - d1;
                const bool can2 = can && c1 != 0 && gj > t + 1 && d2 != 0;
                if (can) {
                    if (gl == t) { my_i = ci; my_in = c1; }
                    nspec = t + 1;
                }
                if (can2) {
                    if (gl == t + 1) { my_i = c1; my_in = c1 - d2; }
                    nspec = t + 2;
                }
                ci = c1 - d2;
                can = can2;
            }
        }
        // ---- B: one round trip for all speculated diagonal cells and the records behind them
        const uint32_t jk = gj - gl;
        const bool lb = walking && gl < nspec;
        int tv = 0;
        uint32_t bs = 0;
        uint4 rnext = zero4;
        oob = false;
        if (lb) {
            bs = a.b.bases[so + jk - 1];
            if (my_in) rnext = a.dp.rec[nb + my_in - 1];
            tv = Tat(my_in, jk - 1);
            nrounds += gl == 0;
        }
        const uint32_t codek = (uint32_t)shr1((int)rnext.x, (int)grec.x) & 0xFF;       // code of position k's row
        const int tprev = shr1(tv, gT);                                               // T at position k
        // a speculated cell outside the band confirms nothing: the walk stops in front of it and the general step decides
        bool ok = lb && !oob && tprev == tv + (((bs == codek) ? m : n) - g);
        if (!nw && tprev == -(int)jk * g) ok = false;                   // SW: the walk ends at this position
        // ---- C: longest confirmed prefix
        uint32_t f = 0;
        {
            const uint32_t gm = gmask(__ballot(ok));
            f = (uint32_t)__ffs((int)(~gm & ((2u << TL) - 1u))) - 1;
            if (!walking) f = 0;
            if (f && gnout + f > a.PC) { govf = true; walking = false; f = 0; }
        }
        if (gl < f) out[gnout + gl] = (my_i << 16) | jk;
        if (pf_hi >> 31) { blk_store((wlo >> 6) - 1u, pf_lo, pf_hi & 0x7FFFFFFFu); wlo -= 64u; }      // (this round's cells are in: the block asked for in front of them is, too)
        bool cont = false;
        {
            const uint32_t src = gbase + (f ? f - 1 : 0);
            const uint32_t ni = (uint32_t)__shfl((int)my_in, (int)src, 64);
            const int nT = __shfl(tv, (int)src, 64);
            const uint4 nr = perm4(rnext, src);
            if (f) {
                gnout += f; nspec_ok += f;
                gi = ni; gT = nT; grec = ni ? nr : zero4; gj -= f;
                cont = f == VC_SPECW;                                       // everything confirmed: speculate again
                if (!cont && (nw ? (gi == 0 && gj == 0) : (gT == -(int)gj * g))) walking = false;
            }
        }
        const bool need = walking && !cont;
        // ---- D: one general step at (gi, gj) for the groups that stopped short
        if (__any(need)) {
            uint32_t pi_ = 0, pj_ = 0;
            int hv = 0;
            uint4 nrec = zero4;
            bool found = false, have_v = false;
            uint32_t v_pi = 0; int v_hv = 0; uint4 v_rec = zero4;
            int hz = 0;                                                  // lane 15 of the group: T[gi][gj-1]
            oob = false;
            if (need && gl == TL - 1 && gj != 0) hz = Tat(gi, gj - 1);
            const bool isovf = ((grec.x >> 8) & VC_RF_OVF) != 0;
            const uint32_t np = (need && gi != 0) ? (isovf ? grec.z : ((grec.x >> 16) & 0xFF)) : 0u;
            int sc = 0;
            if (np && gj != 0) sc = ((a.b.bases[so + gj - 1] == (grec.x & 0xFF)) ? m : n) - g;
            const bool isd = gl < ND_, isv = gl >= ND_ + 1 && gl < 2 * ND_ + 1;
            for (uint32_t base = 0; __any(!found && base < np); base += ND_) {
                const uint32_t p = base + (isd ? gl : gl - (ND_ + 1));
                const bool act = !found && (isd || isv) && p < np && (!isd || gj != 0) && (isd || !have_v);
                uint32_t delta = 0;
                if (act) {
                    if (isovf) { delta = a.dp.ovf[eb + grec.y + p]; if (a.kept && (delta & 0x8000u)) delta &= 0x7Fu; }
                    else {
                        const uint32_t wsel = p < 2 ? grec.y : (p < 4 ? grec.z : grec.w);
                        delta = (p & 1) ? (wsel >> 16) : (wsel & 0xFFFF);
                    }
                }
                const uint32_t pr = gi - delta;
                int cv = 0;
                uint4 rr = zero4;
                if (act) {
                    if (pr) rr = a.dp.rec[nb + pr - 1];
                    cv = Tat(pr, isd ? gj - 1 : gj);
                }
                const bool match = act && gT == cv + (isd ? sc : g);
                const uint32_t gm = gmask(__ballot(match));
                const uint32_t dm = gm & ((1u << ND_) - 1u), vm = (gm >> (ND_ + 1)) & ((1u << ND_) - 1u);
                const bool take_d = !found && dm != 0, take_v = !found && dm == 0 && vm != 0 && !have_v;
                const uint32_t src = gbase + (dm ? (uint32_t)__ffs((int)dm) - 1 : (vm ? ND_ + 1 + (uint32_t)__ffs((int)vm) - 1 : 0u));
                const uint32_t s_pr = (uint32_t)__shfl((int)pr, (int)src, 64);
                const int s_cv = __shfl(cv, (int)src, 64);
                const uint4 s_rr = perm4(rr, src);
                if (take_d) { pi_ = s_pr; pj_ = gj - 1; hv = s_cv; nrec = s_rr; found = true; }
                else if (take_v) { v_pi = s_pr; v_hv = s_cv; v_rec = s_rr; have_v = true; }
            }
            if (need && !found && have_v) { pi_ = v_pi; pj_ = gj; hv = v_hv; nrec = v_rec; found = true; }
            if (a.band) {
                // a candidate of this step lay outside the band: the decision cannot be taken from what is stored -- give the
                // alignment up here; k_fwd stores its matrix whole in the redo pass and the walk is done again from there
                const bool goob = gmask(__ballot(oob && need)) != 0;
                if (need && goob) { gredo = true; walking = false; }
            }
            {
                const int v = __shfl(hz, (int)(gbase + TL - 1), 64);
                if (need && !found && gj != 0 && gT == v) { pi_ = gi; pj_ = gj - 1; hv = v; nrec = grec; found = true; }
            }
            if (need && !gredo) {
                if (!found) { gbroken = true; walking = false; }
                else if (gnout >= a.PC) { govf = true; walking = false; }
                else {
                    if (gl == 0) out[gnout] = ((gi == pi_ ? 0u : gi) << 16) | (gj == pj_ ? 0u : gj);
                    gnout++;
                    gi = pi_; gj = pj_; gT = hv; grec = pi_ ? nrec : zero4;
                }
            }
        }
    }
    if (valid && gl == 0) {
        if (gredo) { gnout = 0; if (!PIPE) a.redo_out[atomicAdd(a.redo_out_n, 1u)] = job; }
        else if (gbroken) { vc_fail(a.b, w, VC_WIN_INVALID, 17, gi); gnout = 0; }
        else if (govf) { vc_fail(a.b, w, VC_WIN_OVERFLOW, 5, gnout); gnout = 0; }
        a.npairs[pj] = gnout;
        // the window stays at this layer when the walk left the band (k_addaln passes it over, the next forward pass stores its rows
        // whole); a walk over whole rows cannot leave anything: the flag goes
        if (!PIPE && a.cursor) a.cursor[slot] = k | (gredo ? 0x80000000u : 0u);
    }
    {   // statistics: summed over the wave first, then one of VC_STAT_SLOTS counter sets (a single set serialises in the L2)
        uint32_t s0 = (valid && gl == 0) ? gnout : 0u, s1 = (valid && gl == 0) ? nspec_ok : 0u, s2 = (valid && gl == 0) ? nrounds : 0u;
#pragma unroll
        for (int o = TL; o < TG * TL; o <<= 1) { s0 += (uint32_t)__shfl_xor((int)s0, o, 64); s1 += (uint32_t)__shfl_xor((int)s1, o, 64); s2 += (uint32_t)__shfl_xor((int)s2, o, 64); }
        if (lane == 0) {
            unsigned long long* st = vc_stat_slot(a.stat);
            atomicAdd(st + 4, (unsigned long long)s0); atomicAdd(st + 5, (unsigned long long)s1); atomicAdd(st + 6, (unsigned long long)s2);
        }
    }
    return valid && gredo;
}

// (112 VGPRs: what is left on a SIMD beside five forward waves of 80)
template <int TL>
VC_KL __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(112))) void k_tracew(VcTraceArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int TG = 64 / TL;
    const uint32_t grp = (uint32_t)vc_lane() / TL;
    const uint32_t njobs = a.nslots * a.group;
    const bool redo = a.redo_list != nullptr;
    uint32_t job = blockIdx.x * TG + grp;
    bool valid = job < njobs;
    if (redo) {                                               // second pass: the jobs the first one gave up on
        const uint32_t nr = *a.redo_n;
        if (blockIdx.x * TG >= nr) return;
        valid = job < nr;
        job = valid ? a.redo_list[job] : 0u;
    }
    const uint32_t slot = valid ? job / a.group : 0;
    uint32_t k = valid ? a.k0 + job % a.group : 0;
    bool whole = redo;                                        // this alignment's matrix was stored whole
    if (a.cursor && valid) {
        const uint32_t cv = a.cursor[slot];
        k = cv & 0xFFFFu;
        whole = redo || (cv >> 31) != 0;
    }
    const uint64_t pj = a.cursor ? (uint64_t)slot : (uint64_t)slot * a.pair_group + (k - a.pair_k0);
    (void)vc_tracew_body<false, TL>(a, smem, job, slot, k, pj, valid, whole);
}

#ifdef VC_EXPERIMENTS
#include "vc_traceb.h"     // k_traceb: the backtrack walked out of LDS -- measured slower than k_tracew, kept as an experiment
#endif

// ------------------------------------------------------------------------------------------------
// k_addaln: Graph::AddAlignment (graph.cpp:182-299) for the layer just aligned, wave-parallel.
// The reference walks the alignment serially; every decision it takes depends only on the graph
// BEFORE the call (a path visits a node, and an aligned group, at most once), so node/edge ids are
// reproduced with prefix sums over the pair list: new ids are handed out in alignment order exactly
// as nodes_.size()/edges_.size() would grow.
// ------------------------------------------------------------------------------------------------
#ifndef VC_ADD_U
#define VC_ADD_U 2             // blocks of 64 alignment pairs that k_addaln carries through its lookup chains side by side (3: 90 VGPRs, a wave of it does not fit beside five k_fwd waves; 2: + 1.7 % on the job, k_addaln 650 -> 435 ms beside k_fwd; 1: + 0.6 %)
#endif
struct VcAddArgs {
    VcBatchDev b;
    VcGraph g;
    VcDp dp;
    uint32_t w0, nslots, NC, EC, layer;
    const uint32_t* pairs; const uint32_t* npairs; uint32_t PC;
    uint16_t* scratch;            // [CW * (4*PC + NC)]
    uint32_t ring;                // rows k_fwd keeps in LDS (the row records of the NEXT layer are made at the end of this kernel)
    int make_rows;                // 0: leave the row records of THIS layer in place (vc_debug_stop_after looks at them)
    uint32_t kept;                // != 0: slots of k_fwd's kept-row ring (the dynamic LDS then also covers vc_kept_lds_bytes(NC))
    uint32_t* tie_n; uint32_t* redo_n;   // the layer's tie-list and redo-list counters: this is the layer's last kernel, it clears them for the next layer
    uint32_t* cursor;             // != nullptr: [nslots] the layer every window is at (VcFwdArgs::cursor); `layer` is then unused.  A window whose
                                  //   backtrack left the band is passed over; one whose alignment was added moves on to its next layer
};

// AddAlignment of sequence `layer` of window `slot` (+ the row records of the next layer when it is full-span).  `scr`: which of the
// per-wave note blocks in VcAddArgs::scratch this wave uses (lock-step launches: the workgroup index; persistent pipeline: the
// index of the resident wave).  smem: 2 * (PC + longest sequence) bytes, and vc_kept_lds_bytes(NC) for the row records.
#ifdef VC_ADD_PROF          // development (tools/build_variant.sh addprof -DVC_ADD_PROF): shader-clock ticks of k_addaln's phases, summed over waves
__device__ unsigned long long vc_add_prof[8];
#define VC_ADD_STAMP(i) do { const long long n_ = clock64(); if (vc_lane() == 0) atomicAdd(&vc_add_prof[i], (unsigned long long)(n_ - t_prof)); t_prof = n_; } while (0)
#else
#define VC_ADD_STAMP(i) do { } while (0)
#endif
// -> the alignment was added (false: nothing to do for this window, or it failed and carries its status)
__device__ __forceinline__ bool vc_addaln_body(const VcAddArgs& a, uint8_t* smem, const uint32_t slot, const uint32_t layer, const uint32_t scr) {
#ifdef VC_ADD_PROF
    long long t_prof = clock64();
#endif
    uint16_t* s_curr = (uint16_t*)smem;                 // [PC] node chosen for each pair (forward order)
    uint16_t* s_anchor = s_curr + a.PC;                 // [max_len] new node t goes in front of old position anchor[t]
    // bulky per-pair notes live in HBM scratch, not LDS: this kernel shares CUs with k_fwd of the other
    // stream and must leave it the LDS (each note is written once and read a few times in pass D)
    uint16_t* s_row = a.scratch + (uint64_t)scr * (4 * a.PC + a.NC);   // [PC] DP row of the pair (0 = none)
    uint16_t* s_bs = s_row + a.PC;                      // [PC] first / last position in VcGraph::ord of the
    uint16_t* s_be = s_bs + a.PC;                       //      aligned group of the pair's node
    uint16_t* s_pn = s_be + a.PC;                       // [PC] position of the pair's node itself
    uint16_t* s_ord = s_pn + a.PC;                      // [NC] old order
    if (slot >= a.nslots) return false;
    const uint32_t w = a.w0 + slot;
    if (a.b.status[w] != VC_WIN_OK) return false;
    const int lane = vc_lane();
    const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
    if (layer >= ns) return false;
    const uint64_t so = a.b.seq_off[s0 + layer];
    const uint32_t len = (uint32_t)(a.b.seq_off[s0 + layer + 1] - so);
    const bool hq = a.b.seq_has_qual[s0 + layer] != 0;
    const uint32_t P = a.npairs[slot];
    const uint32_t* pr = a.pairs + (uint64_t)slot * a.PC;
    const uint64_t nb = (uint64_t)slot * a.NC, eb = (uint64_t)slot * a.EC;
    const uint32_t N0 = a.g.n_nodes[slot], E0 = a.g.n_edges[slot];
    int err = 0;

    // pass A: choose the node of every aligned base; count new nodes.  The lookups of a pair are a chain (pair -> node -> its
    // position, code and aligned mates -> their positions and codes); VC_ADD_U blocks of 64 pairs go through it level by level,
    // so that a lane has the loads of several pairs in flight instead of one chain after the other
    uint32_t nnew = 0, nvalid = 0;
    constexpr int AU = VC_ADD_U, AM = 4;                      // AM: aligned mates fetched with the node (longer lists: the loop below)
    const uint32_t ma = a.g.ma;
    for (uint32_t f0 = 0; f0 < P; f0 += 64 * AU) {
        uint32_t pv[AU], nd[AU], cb[AU], pn[AU], cnt[AU], cd[AU], alv[AU][AM], pa[AU][AM], ca[AU][AM];
        bool act[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const uint32_t f = f0 + 64 * u + lane;
            act[u] = f < P;
            pv[u] = act[u] ? pr[P - 1 - f] : 0u;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const uint32_t row = pv[u] >> 16, col = pv[u] & 0xFFFF;
            nd[u] = (act[u] && row != 0) ? (uint32_t)a.dp.rank2node[nb + row - 1] : 0u;
            cb[u] = (act[u] && col != 0) ? (uint32_t)a.b.bases[so + col - 1] : 0u;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            pn[u] = 0; cnt[u] = 0; cd[u] = 0;
#pragma unroll
            for (int t = 0; t < AM; ++t) alv[u][t] = 0;
            if (act[u] && (pv[u] >> 16) != 0) {
                pn[u] = a.g.pos[nb + nd[u]];
                cnt[u] = a.g.al_cnt[nb + nd[u]];
                cd[u] = a.g.code[nb + nd[u]];
#pragma unroll
                for (int t = 0; t < AM; ++t) if ((uint32_t)t < ma) alv[u][t] = a.g.al[(nb + nd[u]) * ma + t];
            }
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
#pragma unroll
            for (int t = 0; t < AM; ++t) {
                pa[u][t] = 0; ca[u][t] = 0;
                if ((uint32_t)t < cnt[u]) { pa[u][t] = a.g.pos[nb + alv[u][t]]; ca[u][t] = a.g.code[nb + alv[u][t]]; }
            }
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const uint32_t f = f0 + 64 * u + lane;
            const uint32_t row = pv[u] >> 16, col = pv[u] & 0xFFFF;
            const bool valid = act[u] && col != 0;
            uint32_t curr = VC_NONE16;
            bool isnew = false;
            if (act[u]) {
                s_row[f] = (uint16_t)row;
                if (row != 0) {
                    uint32_t bs = pn[u], be = pn[u];
#pragma unroll
                    for (int t = 0; t < AM; ++t) if ((uint32_t)t < cnt[u]) { bs = min(bs, pa[u][t]); be = max(be, pa[u][t]); }
                    for (uint32_t t = AM; t < cnt[u]; ++t) {              // alphabets beyond five bytes
                        const uint32_t pa2 = a.g.pos[nb + a.g.al[(nb + nd[u]) * ma + t]];
                        bs = min(bs, pa2); be = max(be, pa2);
                    }
                    s_pn[f] = (uint16_t)pn[u]; s_bs[f] = (uint16_t)bs; s_be[f] = (uint16_t)be;
                }
            }
            if (valid) {
                if (row == 0) isnew = true;                                    // graph.cpp:249-251
                else if (cd[u] == cb[u]) curr = nd[u];                         // :254-257
                else {
#pragma unroll
                    for (int t = AM - 1; t >= 0; --t) if ((uint32_t)t < cnt[u] && ca[u][t] == cb[u]) curr = alv[u][t];   // :258-266, first match wins
                    if (curr == VC_NONE16) {
                        for (uint32_t t = AM; t < cnt[u]; ++t) {
                            const uint32_t al = a.g.al[(nb + nd[u]) * ma + t];
                            if (a.g.code[nb + al] == cb[u]) { curr = al; break; }
                        }
                    }
                    if (curr == VC_NONE16) isnew = true;                       // :267-277
                }
            }
            uint32_t tot;
            const uint32_t my = wave_excl_sum(isnew ? 1u : 0u, tot);
            if (isnew) curr = N0 + nnew + my;
            if (act[u]) s_curr[f] = valid ? (uint16_t)curr : VC_NONE16;
            nnew += tot;
            nvalid += __popcll(__ballot(valid));
        }
    }
    // every base must be aligned (NW alignments always are); otherwise the reference would add
    // unaligned prefix/suffix chains (graph.cpp:233-236), which this flow never produces
    if (nvalid != len || P == 0) err = VC_WIN_INVALID;
    if (N0 + nnew > a.NC || N0 + nnew >= 0xFFFF) err = VC_WIN_OVERFLOW;
#ifdef VC_DBG_ADD
    if (err) { if (lane == 0) vc_fail(a.b, w, err, 6, (nvalid & 0x3FF) | ((P & 0x3F) << 10)); return false; }
#endif
    if (err) { if (lane == 0) vc_fail(a.b, w, err, 6, nvalid != len || P == 0 ? 1 : 2); return false; }
    __syncthreads();
    VC_ADD_STAMP(0);

    // pass B: create nodes, extend aligned groups
    for (uint32_t f0 = 0; f0 < P; f0 += 64) {
        const uint32_t f = f0 + lane;
        if (f >= P) continue;
        const uint32_t pv = pr[P - 1 - f];
        const uint32_t row = pv >> 16, col = pv & 0xFFFF;
        if (col == 0) continue;
        const uint32_t curr = s_curr[f];
        if (curr < N0) continue;
        a.g.code[nb + curr] = a.b.bases[so + col - 1];
        a.g.in_first[nb + curr] = VC_NONE16; a.g.in_last[nb + curr] = VC_NONE16;
        a.g.out_first[nb + curr] = VC_NONE16; a.g.out_last[nb + curr] = VC_NONE16;
        uint32_t mycnt = 0;
        if (row != 0) {
            const uint32_t nd = a.dp.rank2node[nb + row - 1];
            const uint32_t cnt = a.g.al_cnt[nb + nd];
            if (cnt + 1 > a.g.ma) { err = VC_WIN_INVALID; }              // cannot happen: a group holds distinct bytes, ma >= alphabet - 1
            else {
                for (uint32_t t = 0; t < cnt; ++t) {
                    const uint32_t al = a.g.al[(nb + nd) * a.g.ma + t];
                    const uint32_t ac = a.g.al_cnt[nb + al];
                    if (ac + 1 > a.g.ma) { err = VC_WIN_INVALID; break; }
                    a.g.al[(nb + al) * a.g.ma + ac] = (uint16_t)curr;
                    a.g.al_cnt[nb + al] = (uint8_t)(ac + 1);
                    a.g.al[(nb + curr) * a.g.ma + t] = (uint16_t)al;
                }
                a.g.al[(nb + nd) * a.g.ma + cnt] = (uint16_t)curr;
                a.g.al_cnt[nb + nd] = (uint8_t)(cnt + 1);
                a.g.al[(nb + curr) * a.g.ma + cnt] = (uint16_t)nd;
                mycnt = cnt + 1;
            }
        }
        a.g.al_cnt[nb + curr] = (uint8_t)mycnt;
        a.g.visits[nb + curr] = 0;
        a.g.nrec[nb + curr] = make_uint4((uint32_t)a.b.bases[so + col - 1], 0u, 0u, 0u);
    }
    if (__any(err)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 7, 0); return false; }
    __syncthreads();      // pass B's stores are complete before pass C touches the same nodes
    VC_ADD_STAMP(1);

    // every node on the path gains this sequence's label on an adjacent edge (Node::Coverage, graph.cpp:38-56)
    if (len >= 2) {
        for (uint32_t f = lane; f < P; f += 64) {
            const uint32_t curr = s_curr[f];
            if (curr != VC_NONE16) a.g.visits[nb + curr] = (uint16_t)((curr < N0 ? a.g.visits[nb + curr] : 0) + 1);
        }
    }

    // pass C: edges between consecutive aligned bases (graph.cpp:282-290 -> AddEdge :94-107), VC_ADD_U blocks of pairs level by
    // level like pass A.  A path meets a node once: it is `prev` of one pair and `curr` of one pair, so the list heads, tails and
    // node records the pairs of different blocks touch are different words and the blocks can go side by side.
    uint32_t enew = 0;
    uint32_t carry_prev = VC_NONE16;                   // node of the last aligned base of earlier chunks
    for (uint32_t f0 = 0; f0 < P; f0 += 64 * AU) {
        uint32_t col[AU], curr[AU], prev[AU], wgt[AU], e0[AU], qa[AU], qb[AU], found[AU];
        bool valid[AU], link[AU], make[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const uint32_t f = f0 + 64 * u + lane;
            const bool act = f < P;
            col[u] = act ? (pr[P - 1 - f] & 0xFFFF) : 0u;
            curr[u] = act ? (uint32_t)s_curr[f] : (uint32_t)VC_NONE16;
            valid[u] = act && col[u] != 0;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            // previous aligned base: nearest lower lane with valid, else carry
            const unsigned long long vm = __ballot(valid[u]);
            const unsigned long long below = vm & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll((long long)below) : lane;
            const uint32_t pv_prev = (uint32_t)__shfl((int)curr[u], src, 64);     // outside divergent code
            prev[u] = below ? pv_prev : carry_prev;
            if (vm) {
                const int last = 63 - __clzll((long long)vm);
                carry_prev = (uint32_t)__shfl((int)curr[u], last, 64);
            }
            link[u] = valid[u] && prev[u] != VC_NONE16;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            qa[u] = 0; qb[u] = 0; e0[u] = VC_NONE16;
            if (link[u]) {
                if (hq) { qa[u] = a.b.quals[so + col[u] - 2]; qb[u] = a.b.quals[so + col[u] - 1]; }
                if (prev[u] < N0 && curr[u] < N0) e0[u] = a.g.out_first[nb + prev[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            wgt[u] = 0; found[u] = VC_NONE16;
            if (link[u]) wgt[u] = hq ? a.b.lut_w[qa[u]] + a.b.lut_w[qb[u]] : 2u;
        }
        // the edge prev -> curr, if it exists: walk prev's out-list (all blocks step together)
        for (;;) {
            bool more = false;
            uint32_t hn[AU];
#pragma unroll
            for (int u = 0; u < AU; ++u) hn[u] = e0[u] != VC_NONE16 ? a.g.e_hn[eb + e0[u]] : 0u;
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                if (e0[u] != VC_NONE16) {
                    if ((hn[u] & 0xFFFF) == curr[u]) { found[u] = e0[u]; e0[u] = VC_NONE16; }
                    else e0[u] = hn[u] >> 16;
                }
                more |= e0[u] != VC_NONE16;
            }
            if (!__any(more)) break;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            make[u] = false;
            if (link[u]) {
                if (found[u] != VC_NONE16) a.g.e_w[eb + found[u]] += wgt[u];
                else make[u] = true;
            }
        }
        // new edges: ids in path order; list tails and node records of all blocks first, then the writes
        uint32_t eid[AU], ol[AU], il[AU];
        uint4 nr[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            uint32_t tot;
            const uint32_t my = wave_excl_sum(make[u] ? 1u : 0u, tot);
            eid[u] = E0 + enew + my;
            enew += tot;
            if (make[u] && (eid[u] >= a.EC || eid[u] >= 0xFFFF)) { err = VC_WIN_OVERFLOW; make[u] = false; }
            ol[u] = VC_NONE16; il[u] = VC_NONE16; nr[u] = make_uint4(0, 0, 0, 0);
            if (make[u]) {
                if (prev[u] < N0) ol[u] = a.g.out_last[nb + prev[u]];
                if (curr[u] < N0) il[u] = a.g.in_last[nb + curr[u]];
                nr[u] = a.g.nrec[nb + curr[u]];
            }
        }
        uint32_t hol[AU], til[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            hol[u] = (make[u] && ol[u] != VC_NONE16) ? a.g.e_hn[eb + ol[u]] : 0u;
            til[u] = (make[u] && il[u] != VC_NONE16) ? a.g.e_tn[eb + il[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            if (!make[u]) continue;
            const uint32_t e = eid[u];
            a.g.e_tn[eb + e] = prev[u] | ((uint32_t)VC_NONE16 << 16);
            a.g.e_hn[eb + e] = curr[u] | ((uint32_t)VC_NONE16 << 16);
            a.g.e_w[eb + e] = wgt[u];
            // append to prev's out-list
            if (ol[u] == VC_NONE16) a.g.out_first[nb + prev[u]] = (uint16_t)e;
            else a.g.e_hn[eb + ol[u]] = (hol[u] & 0xFFFF) | (e << 16);
            a.g.out_last[nb + prev[u]] = (uint16_t)e;
            // append to curr's in-list
            if (il[u] == VC_NONE16) a.g.in_first[nb + curr[u]] = (uint16_t)e;
            else a.g.e_tn[eb + il[u]] = (til[u] & 0xFFFF) | (e << 16);
            a.g.in_last[nb + curr[u]] = (uint16_t)e;
            // node record: the tail joins the head's inline predecessor list (one new in-edge per node per call)
            uint4 r4 = nr[u];
            const uint32_t k = r4.x >> 16;
            if (k == 0) r4.y = prev[u];
            else if (k == 1) r4.y |= prev[u] << 16;
            else if (k == 2) r4.z = prev[u];
            else if (k == 3) r4.z |= prev[u] << 16;
            else if (k == 4) r4.w = prev[u];
            else if (k == 5) r4.w |= prev[u] << 16;
            r4.x += 1u << 16;
            a.g.nrec[nb + curr[u]] = r4;
        }
    }
    if (__any(err)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_OVERFLOW, 8, E0 + enew); return false; }
    VC_ADD_STAMP(2);

    // pass D: keep VcGraph::ord a valid DP order with aligned groups contiguous.
    //   a node created for a mismatch joins its group right behind the node it was aligned to;
    //   a run of inserted bases goes in front of the group of the next graph node on the path
    //   (behind the previous one's group when the read ends with it).
    // New ids grow along the path and the path follows ord, so anchors are non-decreasing and the new
    // order is the merge: old position p moves right by #(anchors <= p), new node t lands at anchor[t] + t.
    for (uint32_t f0 = 0; f0 < P; f0 += 64) {
        const uint32_t f = f0 + lane;
        if (f >= P) continue;
        const uint32_t curr = s_curr[f];
        if (curr == VC_NONE16 || curr < N0) continue;
        uint32_t anchor;
        if (s_row[f] != 0) anchor = (uint32_t)s_pn[f] + 1;
        else {
            uint32_t gidx = f + 1;
            while (gidx < P && s_row[gidx] == 0) gidx++;
            if (gidx < P) anchor = s_bs[gidx];
            else {
                int h = (int)f - 1;
                while (h >= 0 && s_row[h] == 0) h--;
                anchor = h >= 0 ? (uint32_t)s_be[h] + 1 : N0;
            }
        }
        s_anchor[curr - N0] = (uint16_t)anchor;
    }
    for (uint32_t p = lane; p < N0; p += 64) s_ord[p] = a.g.ord[nb + p];
    __syncthreads();
    for (uint32_t t = lane; t + 1 < nnew; t += 64) if (s_anchor[t + 1] < s_anchor[t]) err = 1;
    if (__any(err)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 16, 0); return false; }
    for (uint32_t p = lane; p < N0; p += 64) {
        uint32_t lo = 0, hi = nnew;                     // first t with anchor[t] > p
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_anchor[mid] <= p) lo = mid + 1; else hi = mid; }
        const uint32_t v = s_ord[p], np_ = p + lo;
        a.g.ord[nb + np_] = (uint16_t)v;
        a.g.pos[nb + v] = (uint16_t)np_;
    }
    for (uint32_t t = lane; t < nnew; t += 64) {
        const uint32_t np_ = (uint32_t)s_anchor[t] + t;
        a.g.ord[nb + np_] = (uint16_t)(N0 + t);
        a.g.pos[nb + N0 + t] = (uint16_t)np_;
    }
    if (lane == 0) { a.g.n_nodes[slot] = N0 + nnew; a.g.n_edges[slot] = E0 + enew; }
    __syncthreads();
    VC_ADD_STAMP(3);
    if (a.make_rows) vc_rows_full(a.b, a.g, a.dp, slot, w, a.NC, a.EC, (int)layer + 1, a.ring, N0 + nnew, a.kept, smem);   // the next layer's rows (full-span layers)
    VC_ADD_STAMP(4);
#ifdef VC_ADD_PROF
    if (vc_lane() == 0) atomicAdd(&vc_add_prof[7], 1ull);
#endif
    return true;
}

VC_KL __global__ __launch_bounds__(64) void k_addaln(VcAddArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    // every reader of the layer's counters (k_resolve, the redo pass) is an earlier kernel of this stream; a memset per counter
    // per layer was 2 000 tiny launches per step, each waiting ~100 us for a slot beside k_fwd
    if (blockIdx.x == 0 && threadIdx.x == 0) { *a.tie_n = 0; *a.redo_n = 0; }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t layer = a.layer;
    if (a.cursor) {
        if (blockIdx.x >= a.nslots) return;
        const uint32_t cv = a.cursor[blockIdx.x];
        if (cv >> 31) return;                                 // its backtrack left the band: the layer is aligned again first
        layer = cv & 0xFFFFu;
    }
    const bool added = vc_addaln_body(a, smem, blockIdx.x, layer, blockIdx.x);
    if (a.cursor && added && threadIdx.x == 0) a.cursor[blockIdx.x] = layer + 1u;
}

// ------------------------------------------------------------------------------------------------
// k_prune_lcc: PruneGraph (graph.cpp:811-982) + LargestSubgraph (graph.cpp:984-1102): src -> dst.
// LDS carve: adj_off 2*(NC+1) | adj 4*EC (u16 x 2E) | vis/newid 2*NC | nin 2*NC | keep EC |
//            { osum 4*NC, isum 4*NC }  aliased later by { frames 4*NC, comp 2*NC, best 2*NC }
// ------------------------------------------------------------------------------------------------
struct VcPruneArgs {
    VcBatchDev b;
    VcGraph src, dst;
    uint32_t w0, nslots, NC, EC;
    uint32_t NCl, ECl;            // capacity of the LDS image (<= NC, EC; the host knows the maximum after a round)
    uint8_t* ws; uint32_t ws_stride;   // != nullptr: image too large for the LDS, use this HBM workspace per workgroup
    double min_conf, min_supp;
};

__host__ __device__ inline uint32_t vc_prune_lds_bytes(uint32_t NC, uint32_t EC) {
    return ((2 * (NC + 1) + 15) & ~15u) + 4 * EC + 2 * NC + 2 * NC + ((EC + 15) & ~15u) + 8 * NC + 64;
}

VC_KL __global__ __launch_bounds__(64) void k_prune_lcc(VcPruneArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t slot = blockIdx.x;
    if (slot >= a.nslots) return;
    const uint32_t w = a.w0 + slot;
    if (a.b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint32_t NC = a.NC, EC = a.EC;
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    const uint32_t N = a.src.n_nodes[slot], E = a.src.n_edges[slot];
    const double avg = a.b.win_avg[w];

    const uint32_t NCl = a.NCl, ECl = a.ECl;
    if (N > NCl || E > ECl) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 25, N); return; }
    uint8_t* const img = a.ws ? a.ws + (size_t)blockIdx.x * a.ws_stride : smem;
    uint16_t* s_adj_off = (uint16_t*)img;                                      // [NCl+1]
    uint16_t* s_adj = (uint16_t*)(img + ((2 * (NCl + 1) + 15) & ~15u));        // [2*ECl]
    uint16_t* s_vis = s_adj + 2 * ECl;                                         // [NCl] visited, later newid
    uint16_t* s_nin = s_vis + NCl;                                             // [NCl] live in-degree
    uint8_t*  s_keep = (uint8_t*)(s_nin + NCl);                                // [ECl]
    uint32_t* s_osum = (uint32_t*)(s_keep + ((ECl + 15) & ~15u));              // [NCl]
    uint32_t* s_isum = s_osum + NCl;                                           // [NCl]
    uint32_t* s_frame = s_osum;                                                // [NCl] (v | cursor << 16)
    uint16_t* s_comp = (uint16_t*)(s_frame + NCl);                             // [NCl]
    uint16_t* s_best = s_comp + NCl;                                           // [NCl]

    // 1. weight totals around every node (the reference re-sums them per edge, graph.cpp:845-872)
    for (uint32_t v = lane; v < N; v += 64) {
        uint32_t so_ = 0, si_ = 0;
        for (uint32_t e = a.src.out_first[nb + v]; e != VC_NONE16; e = a.src.e_hn[eb + e] >> 16) so_ += a.src.e_w[eb + e];
        for (uint32_t e = a.src.in_first[nb + v]; e != VC_NONE16; e = a.src.e_tn[eb + e] >> 16) si_ += a.src.e_w[eb + e];
        s_osum[v] = so_; s_isum[v] = si_;
    }
    __syncthreads();
    // 2. keep / prune decision per edge, fp64 exactly as graph.cpp:861-904 (0/0 = NaN compares false)
    for (uint32_t e = lane; e < E; e += 64) {
        const uint32_t t = a.src.e_tn[eb + e] & 0xFFFF, h = a.src.e_hn[eb + e] & 0xFFFF;
        const double wv = (double)(long long)a.src.e_w[eb + e];
        const double cuv = wv / (double)(long long)s_osum[t];
        const double sup = wv / avg;
        const double cvu = wv / (double)(long long)s_isum[h];
        s_keep[e] = (cuv >= a.min_conf && cvu >= a.min_conf && sup >= a.min_supp) ? 1 : 0;
    }
    __syncthreads();
    // 3. adjacency of the pruned graph: live in-edge tails (list order) then live out-edge heads (list order)
    uint32_t base = 0;
    for (uint32_t v0 = 0; v0 < N; v0 += 64) {
        const uint32_t v = v0 + lane;
        uint32_t din = 0, dout = 0;
        if (v < N) {
            for (uint32_t e = a.src.in_first[nb + v]; e != VC_NONE16; e = a.src.e_tn[eb + e] >> 16) din += s_keep[e];
            for (uint32_t e = a.src.out_first[nb + v]; e != VC_NONE16; e = a.src.e_hn[eb + e] >> 16) dout += s_keep[e];
        }
        uint32_t tot;
        const uint32_t off = wave_excl_sum(din + dout, tot) + base;
        if (v < N) {
            s_adj_off[v] = (uint16_t)off;
            s_nin[v] = (uint16_t)din;
            uint32_t k = off;
            for (uint32_t e = a.src.in_first[nb + v]; e != VC_NONE16; ) {
                const uint32_t tn = a.src.e_tn[eb + e];
                if (s_keep[e]) s_adj[k++] = (uint16_t)(tn & 0xFFFF);
                e = tn >> 16;
            }
            for (uint32_t e = a.src.out_first[nb + v]; e != VC_NONE16; ) {
                const uint32_t hn = a.src.e_hn[eb + e];
                if (s_keep[e]) s_adj[k++] = (uint16_t)(hn & 0xFFFF);
                e = hn >> 16;
            }
            s_vis[v] = 0;
        }
        base += tot;
    }
    if (lane == 0) s_adj_off[N] = (uint16_t)base;
    __syncthreads();

    // 4. components by recursive preorder DFS (DfsUtil, graph.cpp:984-1019); `>=` keeps the later one (:1049)
    __shared__ uint32_t s_nbest;
    if (lane == 0) {
        uint32_t nbest = 0;
        for (uint32_t r = 0; r < N; ++r) {
            if (s_vis[r]) continue;
            uint32_t nc = 0, sp = 0;
            s_vis[r] = 1; s_comp[nc++] = (uint16_t)r;
            s_frame[sp++] = r | ((uint32_t)s_adj_off[r] << 16);
            while (sp) {
                const uint32_t fr = s_frame[sp - 1];
                const uint32_t v = fr & 0xFFFF;
                uint32_t k = fr >> 16;
                const uint32_t kend = s_adj_off[v + 1];
                uint32_t u = VC_NONE16;
                while (k < kend) {
                    const uint32_t cand = s_adj[k++];
                    if (!s_vis[cand]) { u = cand; break; }
                }
                if (u == VC_NONE16) { sp--; continue; }
                s_frame[sp - 1] = v | (k << 16);
                s_vis[u] = 1; s_comp[nc++] = (uint16_t)u;
                s_frame[sp++] = u | ((uint32_t)s_adj_off[u] << 16);
            }
            if (nc >= nbest) {
                nbest = nc;
                for (uint32_t k = 0; k < nc; ++k) s_best[k] = s_comp[k];
            }
        }
        s_nbest = nbest;
    }
    __syncthreads();
    const uint32_t nbest = s_nbest;

    // 5. the new graph: nodes in DFS preorder, edges per node in out-list order with weight 0 (:1069-1085)
    uint16_t* s_newid = s_vis;
    for (uint32_t v = lane; v < N; v += 64) s_newid[v] = VC_NONE16;
    __syncthreads();
    for (uint32_t k = lane; k < nbest; k += 64) s_newid[s_best[k]] = (uint16_t)k;
    __syncthreads();
    uint16_t* s_obase = s_comp;                       // first new edge id of every new node
    uint32_t ebase = 0;
    for (uint32_t k0 = 0; k0 < nbest; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint32_t od = 0, v = 0, ob = 0;
        if (k < nbest) {
            v = s_best[k];
            ob = s_adj_off[v] + s_nin[v];
            od = s_adj_off[v + 1] - ob;
        }
        uint32_t tot;
        const uint32_t my = wave_excl_sum(od, tot) + ebase;
        if (k < nbest) {
            s_obase[k] = (uint16_t)my;
            a.dst.code[nb + k] = a.src.code[nb + v];
            a.dst.al_cnt[nb + k] = 0;
            a.dst.out_first[nb + k] = od ? (uint16_t)my : VC_NONE16;
            a.dst.out_last[nb + k] = od ? (uint16_t)(my + od - 1) : VC_NONE16;
            for (uint32_t t = 0; t < od; ++t) {
                const uint32_t hnew = s_newid[s_adj[ob + t]];
                const uint32_t nxt = t + 1 < od ? my + t + 1 : VC_NONE16;
                a.dst.e_hn[eb + my + t] = hnew | (nxt << 16);
                a.dst.e_w[eb + my + t] = 0;
            }
        }
        ebase += tot;
    }
    __syncthreads();
    // in-lists: edges into h' ordered by new edge id == by the preorder position of their tails
    for (uint32_t k = lane; k < nbest; k += 64) {
        const uint32_t v = s_best[k];
        const uint32_t nin = s_nin[v];
        const uint32_t ib = s_adj_off[v];
        uint32_t prev_e = VC_NONE16, prev_t = 0, first_e = VC_NONE16;
        uint32_t last_t = 0;
        for (uint32_t t = 0; t < nin; ++t) {
            uint32_t bt = 0xFFFFFFFFu;                 // next smallest tail (new numbering) not yet placed
            for (uint32_t u = 0; u < nin; ++u) {
                const uint32_t tl = s_newid[s_adj[ib + u]];
                if ((t == 0 || tl > last_t) && tl < bt) bt = tl;
            }
            last_t = bt;
            const uint32_t tv = s_best[bt];            // its edge id: position of v among tv's live out-heads
            const uint32_t tob = s_adj_off[tv] + s_nin[tv], toe = s_adj_off[tv + 1];
            uint32_t eid = VC_NONE16;
            for (uint32_t x = tob; x < toe; ++x) if (s_adj[x] == v) { eid = s_obase[bt] + (x - tob); break; }
            if (prev_e == VC_NONE16) first_e = eid;
            else a.dst.e_tn[eb + prev_e] = prev_t | (eid << 16);
            prev_e = eid; prev_t = bt;
        }
        if (prev_e != VC_NONE16) a.dst.e_tn[eb + prev_e] = prev_t | ((uint32_t)VC_NONE16 << 16);
        a.dst.in_first[nb + k] = (uint16_t)first_e;
        a.dst.in_last[nb + k] = (uint16_t)prev_e;
    }
    if (lane == 0) { a.dst.n_nodes[slot] = nbest; a.dst.n_edges[slot] = ebase; }
}

// ------------------------------------------------------------------------------------------------
// k_addw: Graph::AddWeights (graph.cpp:1104-1165) for every re-aligned sequence of the window.
// Consecutive fully matched pairs are always joined by an existing edge (a diagonal move is a step
// along an in-edge), so this is a scatter-add; the sum is order-independent.
// ------------------------------------------------------------------------------------------------
struct VcAddwArgs {
    VcBatchDev b;
    VcGraph g;
    VcDp dp;
    uint32_t w0, nslots, NC, EC;
    const uint32_t* pairs; const uint32_t* npairs; uint32_t PC; uint32_t pair_group;
};

VC_KL __global__ __launch_bounds__(64) void k_addw(VcAddwArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    const uint32_t slot = blockIdx.x;
    if (slot >= a.nslots) return;
    const uint32_t w = a.w0 + slot;
    if (a.b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
    const uint64_t nb = (uint64_t)slot * a.NC, eb = (uint64_t)slot * a.EC;
    int err = 0;
    for (uint32_t k = 0; k < ns && k < a.pair_group; ++k) {
        const uint64_t pj = (uint64_t)slot * a.pair_group + k;
        const uint32_t P = a.npairs[pj];
        const uint32_t* pr = a.pairs + pj * a.PC;
        const uint64_t so = a.b.seq_off[s0 + k];
        const bool hq = a.b.seq_has_qual[s0 + k] != 0;
        // stored tail-first: forward neighbours (f-1, f) are stored at (x+1, x)
        for (uint32_t x = lane; x + 1 < P; x += 64) {
            const uint32_t cur = pr[x], prv = pr[x + 1];
            if ((cur >> 16) == 0 || (cur & 0xFFFF) == 0 || (prv >> 16) == 0 || (prv & 0xFFFF) == 0) continue;
            const uint32_t nprev = a.dp.rank2node[nb + (prv >> 16) - 1];
            const uint32_t ncur = a.dp.rank2node[nb + (cur >> 16) - 1];
            const uint32_t q = (cur & 0xFFFF) - 1;
            const uint32_t wgt = vc_weight(a.b, so, q - 1, hq) + vc_weight(a.b, so, q, hq);
            uint32_t found = VC_NONE16;
            for (uint32_t e = a.g.out_first[nb + nprev]; e != VC_NONE16; ) {
                const uint32_t hn = a.g.e_hn[eb + e];
                if ((hn & 0xFFFF) == ncur) { found = e; break; }
                e = hn >> 16;
            }
            if (found == VC_NONE16) err = 1;
            else atomicAdd(&a.g.e_w[eb + found], wgt);
        }
    }
    if (__any(err)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 9, 0); }
}

// ------------------------------------------------------------------------------------------------
// k_finish: GenerateCorrectedSequence (graph.cpp:1167-1179) from the final local alignment
// ------------------------------------------------------------------------------------------------
struct VcFinishArgs {
    VcBatchDev b;
    VcGraph g;
    VcDp dp;
    uint32_t w0, nslots, NC;
    const uint32_t* pairs; const uint32_t* npairs; uint32_t PC;
};

VC_KL __global__ __launch_bounds__(64) void k_finish(VcFinishArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    const uint32_t slot = blockIdx.x;
    if (slot >= a.nslots) return;
    const uint32_t w = a.w0 + slot;
    if (a.b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint64_t nb = (uint64_t)slot * a.NC;
    const uint32_t P = a.npairs[slot];
    const uint32_t* pr = a.pairs + (uint64_t)slot * a.PC;
    uint32_t outn = 0;
    int err = 0;
    for (uint32_t f0 = 0; f0 < P; f0 += 64) {
        const uint32_t f = f0 + lane;
        const uint32_t pv = f < P ? pr[P - 1 - f] : 0;
        const bool has = (pv >> 16) != 0;
        uint32_t tot;
        const uint32_t my = wave_excl_sum(has ? 1u : 0u, tot);
        if (has) {
            const uint32_t pos = outn + my;
            if (pos < a.b.cons_cap) a.b.cons[(uint64_t)w * a.b.cons_cap + pos] = a.g.code[nb + a.dp.rank2node[nb + (pv >> 16) - 1]];
            else err = 1;
        }
        outn += tot;
    }
    if (__any(err)) { if (lane == 0) { vc_fail(a.b, w, VC_WIN_OVERFLOW, 10, outn); a.b.cons_len[w] = 0; } return; }
    if (lane == 0) a.b.cons_len[w] = outn;
}

// ------------------------------------------------------------------------------------------------
// k_consensus: the racon-linear overload's tail (window.cpp:138-171): Graph::GenerateConsensus =
// TraverseHeaviestBundle + BranchCompletion (graph.cpp:534-638), coverage summary (graph.cpp:476-484;
// Node::Coverage == number of sequences through the node, kept in VcGraph::visits) and the TGS trim.
// Serial by nature (scores propagate along the reference's rank order, ties decided by that order);
// runs on lane 0 out of LDS, the graph is staged cooperatively.
// LDS carve: rank 2N | noderank 2N | in_first 2N | out_first 2N | pred 2N | cons 2N | etn 4E | ehn 4E | w 4E | score 8N
// ------------------------------------------------------------------------------------------------
struct VcConsArgs {
    VcBatchDev b;
    VcGraph g;
    VcDp dp;
    uint32_t w0, nslots, NC, EC;
    int trim, window_type;
    uint8_t* ws; uint32_t ws_stride;   // != nullptr: image too large for the LDS, use this HBM workspace per workgroup
};

__host__ __device__ inline uint32_t vc_cons_lds_bytes(uint32_t NC, uint32_t EC) { return 12 * NC + 12 * EC + 8 * NC + 128; }

VC_KL __global__ __launch_bounds__(64) void k_consensus(VcConsArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t slot = blockIdx.x;
    if (slot >= a.nslots) return;
    const uint32_t w = a.w0 + slot;
    if (a.b.status[w] != VC_WIN_OK) return;
    const int lane = vc_lane();
    const uint32_t NC = a.NC, EC = a.EC;
    const uint64_t nb = (uint64_t)slot * NC, eb = (uint64_t)slot * EC;
    const uint32_t N = a.g.n_nodes[slot], E = a.g.n_edges[slot];
    const uint32_t nseq = a.b.win_seq_off[w + 1] - a.b.win_seq_off[w];
    long long* s_score = (long long*)(a.ws ? a.ws + (size_t)blockIdx.x * a.ws_stride : smem);   // [NC]
    uint32_t* s_etn = (uint32_t*)(s_score + NC);                // [EC]
    uint32_t* s_ehn = s_etn + EC;
    uint32_t* s_w = s_ehn + EC;
    uint16_t* s_rank = (uint16_t*)(s_w + EC);                   // [NC] each
    uint16_t* s_nrank = s_rank + NC;
    uint16_t* s_inf = s_nrank + NC;
    uint16_t* s_outf = s_inf + NC;
    uint16_t* s_pred = s_outf + NC;
    uint16_t* s_cons = s_pred + NC;
    for (uint32_t i = lane; i < N; i += 64) {
        const uint32_t v = a.dp.rank2node[nb + i];
        s_rank[i] = (uint16_t)v; s_nrank[v] = (uint16_t)i;
        s_inf[i] = a.g.in_first[nb + i]; s_outf[i] = a.g.out_first[nb + i];
        s_pred[i] = VC_NONE16; s_score[i] = -1;
    }
    for (uint32_t e = lane; e < E; e += 64) { s_etn[e] = a.g.e_tn[eb + e]; s_ehn[e] = a.g.e_hn[eb + e]; s_w[e] = a.g.e_w[eb + e]; }
    __syncthreads();
    __shared__ uint32_t s_ncons;
    if (lane == 0) {
        // relax one node over its in-edges in list order (graph.cpp:549-559 / :615-632)
        auto relax = [&](uint32_t it, bool skip_dead) {
            for (uint32_t e = s_inf[it]; e != VC_NONE16; ) {
                const uint32_t tn = s_etn[e];
                const uint32_t tl = tn & 0xFFFF;
                const long long wt = (long long)s_w[e];
                e = tn >> 16;
                if (skip_dead && s_score[tl] == -1) continue;
                if (s_score[it] < wt || (s_score[it] == wt && s_score[s_pred[it]] <= s_score[tl])) { s_score[it] = wt; s_pred[it] = (uint16_t)tl; }
            }
            if (s_pred[it] != VC_NONE16) s_score[it] += s_score[s_pred[it]];
        };
        uint32_t mx = VC_NONE16;
        for (uint32_t r = 0; r < N; ++r) {
            const uint32_t it = s_rank[r];
            relax(it, false);
            if (mx == VC_NONE16 || s_score[mx] < s_score[it]) mx = it;
        }
        while (s_outf[mx] != VC_NONE16) {                         // BranchCompletion, graph.cpp:590-638
            const uint32_t start = mx, rk = s_nrank[mx];
            for (uint32_t e = s_outf[start]; e != VC_NONE16; ) {
                const uint32_t hn = s_ehn[e];
                for (uint32_t e2 = s_inf[hn & 0xFFFF]; e2 != VC_NONE16; ) {
                    const uint32_t tn = s_etn[e2];
                    if ((tn & 0xFFFF) != start) s_score[tn & 0xFFFF] = -1;
                    e2 = tn >> 16;
                }
                e = hn >> 16;
            }
            uint32_t m2 = VC_NONE16;
            for (uint32_t r = rk + 1; r < N; ++r) {
                const uint32_t it = s_rank[r];
                s_score[it] = -1; s_pred[it] = VC_NONE16;
                relax(it, true);
                if (m2 == VC_NONE16 || s_score[m2] < s_score[it]) m2 = it;
            }
            if (m2 == VC_NONE16) break;                           // cannot happen: a non-sink has a successor
            mx = m2;
        }
        uint32_t n = 0;
        while (s_pred[mx] != VC_NONE16) { s_cons[n++] = (uint16_t)mx; mx = s_pred[mx]; }
        s_cons[n++] = (uint16_t)mx;
        for (uint32_t x = 0, y = n - 1; x < y; ++x, --y) { const uint16_t t = s_cons[x]; s_cons[x] = s_cons[y]; s_cons[y] = t; }
        s_ncons = n;
    }
    __syncthreads();
    const uint32_t n = s_ncons;
    // coverage of every consensus position (own + aligned nodes), then the TGS trim (window.cpp:141-171)
    uint16_t* s_cov = s_pred;                                    // reuse
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t v = s_cons[i];
        uint32_t cv = a.g.visits[nb + v];
        const uint32_t cnt = a.g.al_cnt[nb + v];
        for (uint32_t t = 0; t < cnt; ++t) cv += a.g.visits[nb + a.g.al[(nb + v) * a.g.ma + t]];
        s_cov[i] = (uint16_t)(cv > 0xFFFF ? 0xFFFF : cv);
    }
    __syncthreads();
    __shared__ int s_be[2];
    if (lane == 0) {
        int begin = 0, end = (int)n - 1;
        if (a.window_type == 1 && a.trim) {
            const uint32_t avgc = (nseq - 1) / 2;
            for (; begin < (int)n; ++begin) if (s_cov[begin] >= avgc) break;
            for (; end >= 0; --end) if (s_cov[end] >= avgc) break;
            if (begin >= end) { begin = 0; end = (int)n - 1; }      // "might be chimeric": left untrimmed
        }
        s_be[0] = begin; s_be[1] = end;
    }
    __syncthreads();
    const int begin = s_be[0], end = s_be[1];
    const uint32_t outn = end >= begin ? (uint32_t)(end - begin + 1) : 0;
    if (outn > a.b.cons_cap) { if (lane == 0) vc_fail(a.b, w, VC_WIN_OVERFLOW, 18, outn); return; }
    for (uint32_t i = lane; i < outn; i += 64) a.b.cons[(uint64_t)w * a.b.cons_cap + i] = a.g.code[nb + s_cons[begin + i]];
    if (lane == 0) a.b.cons_len[w] = outn;
}

// which byte values occur in the batch (256-bit mask): sizes the aligned lists (VcGraph::ma)
VC_KL __global__ void k_byte_presence(const uint8_t* bases, uint64_t n, uint32_t* mask) {
    __shared__ uint32_t s_m[8];
    if (threadIdx.x < 8) s_m[threadIdx.x] = 0;
    __syncthreads();
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(bases + i);           // the buffer is 16-byte aligned and padded
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) { const uint32_t c = (w[k] >> (8 * b)) & 0xFF; m[c >> 5] |= 1u << (c & 31); }
        } else {
            for (uint64_t j = i; j < n; ++j) { const uint32_t c = bases[j]; m[c >> 5] |= 1u << (c & 31); }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (m[k]) atomicOr(&s_m[k], m[k]);
    __syncthreads();
    if (threadIdx.x < 8 && s_m[threadIdx.x]) atomicOr(&mask[threadIdx.x], s_m[threadIdx.x]);
}

VC_KL __global__ void k_max_u32(const uint32_t* v, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(out, v[i]);
}

// build phase with a layer cursor per window: how many layers the slowest window of the chunk still has to go (windows that had
// to repeat a layer with whole rows are behind the launch count)
VC_KL __global__ void k_lag(VcBatchDev b, const uint32_t* cursor, uint32_t w0, uint32_t nslots, uint32_t* out) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lag = 0;
    if (slot < nslots) {
        const uint32_t w = w0 + slot, ns = b.win_seq_off[w + 1] - b.win_seq_off[w];
        const uint32_t at = cursor[slot] & 0xFFFFu;
        if (ns >= 3 && b.status[w] == VC_WIN_OK && at < ns) lag = ns - at;
    }
    if (lag) atomicMax(out, lag);
}

// compacts the per-window consensus slots into one contiguous buffer (offsets from an exclusive scan)
VC_KL __global__ void k_gather_cons(VcBatchDev b, const uint64_t* off, uint8_t* out, uint64_t cap) {
    const uint32_t w = blockIdx.x;
    if (w >= b.n_windows) return;
    const uint64_t o = off[w];
    const uint32_t n = b.cons_len[w];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        if (o + i < cap) out[o + i] = b.cons[(uint64_t)w * b.cons_cap + i];
}
