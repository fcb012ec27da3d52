// k_traceb: the backtrack of sisd_alignment_engine.cpp:362-459 walked out of LDS (round 6).
//
// k_tracew (vc_kernels.h) goes to HBM for every cell it looks at: a round is two dependent round trips (the speculated diagonal
// cells, then the general step at the first one that did not confirm), ~113 rounds per alignment, and a wave of it spends two thirds
// of its life waiting -- with 112 VGPRs that the forward waves next to it cannot have.  Here the rows come to the walk instead:
//
//   * every alignment (8 lanes of a wave, as in k_tracew<8>) keeps the last VC_TB_SLOTS BLOCKS of 8 stored rows in LDS -- of every
//     row a WINDOW of 48 bytes (the band lanes around the column the walk is expected to cross that block at: 4 lanes of 8 / 10
//     columns, 2 of 16 / 20) and the row's 16-byte record.  The rows of a block share one band position (vc_band_row_start: blocks of
//     VC_BAND_ROWS rows, a multiple of 8), so its window is one rectangle of the stored band;
//   * blocks are fetched with global_load_lds (16 bytes per lane straight into LDS, no registers, nothing to wait for until the
//     data is used): 3 instructions for the rows and 1 for the records of a block, issued when the walk enters the block
//     VC_TB_SLOTS - 1 blocks above it;
//   * a move is ONE general step -- k_tracew's step D: lanes 0..2 test the diagonal through in-edge p, lanes 4..6 the vertical move,
//     lane 7 the horizontal one, ballots pick the first match in the reference's order (sisd :392-448) -- whose cells and records come
//     from LDS.  No speculation table, no division (the lane and the cell of the column move with the walk), no HBM round trip on
//     the path of a move;
//   * a cell that is not in LDS (a predecessor further up than the cached blocks, a column that left its block's window) is read
//     from memory exactly as k_tracew reads it -- the cache never decides anything, it only answers faster -- and a cell outside
//     the stored band puts the alignment on the redo list as before.
//
// Stored row forms handled: byte-packed rows (VcTraceArgs::packed), banded or whole, singly or doubly tilted (VC_JOB_DT); raw int16
// rows and the 32-bit matrices of k_fwd_wide stay with k_tracew / k_trace.
#pragma once

#ifndef VC_TB_SLOTS
#define VC_TB_SLOTS 4          // blocks of 8 rows in LDS per alignment (a power of two)
#endif
#define VC_TB_NCH 3u           // 16-byte pieces of a row's window
static_assert(VC_BAND_ROWS % 8 == 0, "k_traceb caches blocks of 8 rows: they must share a band position (vc_band_row_start)");
static_assert((VC_TB_SLOTS & (VC_TB_SLOTS - 1)) == 0 && VC_TB_SLOTS >= 2 && VC_TB_SLOTS <= 4, "slots: 2 or 4 (their windows' lanes share one register)");
__host__ __device__ inline uint32_t vc_traceb_lds_bytes() { return VC_TB_SLOTS * (VC_TB_NCH + 1u) * 1024u; }

typedef __attribute__((address_space(3))) void* vc_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* vc_glb_ptr_t;
__device__ __forceinline__ void vc_glds16(const void* src, uint8_t* lds_base) {        // this lane's 16 bytes -> lds_base + lane * 16
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds((vc_glb_ptr_t)src, (vc_lds_ptr_t)lds_base, 16, 0, 0);
#else
    (void)src; (void)lds_base;
#endif
}

__device__ __forceinline__ bool vc_traceb_body(const VcTraceArgs& a, uint8_t* smem, uint32_t job, const uint32_t slot, const uint32_t k,
                                               const uint64_t pj, bool valid, const bool redo) {
    constexpr uint32_t TL = 8, ND_ = 3, NS = VC_TB_SLOTS, NCH = VC_TB_NCH;
    constexpr uint32_t ROWB = 0, RECB = NS * NCH * 1024u;            // LDS: rows [slot][piece t][group][lane] x 16 B, then records [slot][group][row] x 16 B
    const uint32_t lane = (uint32_t)vc_lane();
    const uint32_t grp = lane / TL, gl = lane % TL, gbase = grp * TL;
    const uint32_t w = a.w0 + slot;
    const uint8_t type_raw = valid ? a.job_type[job] : (uint8_t)255;
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
    const uint32_t cpl = max(vc_cpl_for((uint32_t)(a.b.seq_off[sq + 1] - so)), a.cpl_lo), nds = (uint32_t)vc_nds((int)cpl);
    const uint32_t band_lanes = vc_band_lanes(cpl, a.band_chain != 0);
    const bool band = a.band != 0 && !redo && valid && type == 1;
    const uint32_t band_ql = band ? a.band_par[job] : 0u;
    // stored rows of this job: row r (1-based) at rows32 + (r - 1) * rstride, its first stored lane first (band: vc_band_row_start; whole rows: lane 0)
    const uint32_t* const rows32 = band ? a.bmat + (uint64_t)(valid ? job : 0) * vc_band_job_dwords(a.hstride) : a.hmat + (uint64_t)(valid ? job : 0) * a.hstride;
    const uint32_t rlanes = band ? band_lanes : 64u, rstride = rlanes * nds;
    const int16_t* c0 = a.c0 + (uint64_t)(valid ? job : 0) * a.NC;
    const uint32_t W = max(min((4u * NCH) / nds, rlanes), 1u);      // lanes of a row that fit the 48-byte window
    const float rcpl = 1.0f / (float)cpl;
    bool oob = false;
    const uint32_t nrows = valid ? a.dp.nrows[slot] : 0u;

    // ---- the cache: blocks [clow, block of the walk's row]; block b = rows 8 b + 1 .. 8 b + 8 sits in slot b % NS with the lanes [wl, wl + W) of its rows
    uint32_t clow = 0x7FFFFFFFu;                               // lowest block asked for (nothing yet)
    uint32_t lland = 0x7FFFFFFFu;                              // blocks >= lland are known to have landed
    uint32_t wlp = 0;                                          // wl of the NS slots, a byte each
    uint8_t* const rowl = smem + ROWB + grp * 128u;
    uint8_t* const recl = smem + RECB + grp * 128u;
    auto lds_at = [&](uint32_t s, uint32_t rowin, uint32_t o) __attribute__((always_inline)) -> const uint8_t* {      // byte o of the window of row `rowin` of slot s
        return rowl + s * (NCH * 1024u) + (o >> 4) * 1024u + rowin * 16u + (o & 15u);                                   // (piece t of the eight rows = one load instruction)
    };
    auto cell_mem = [&](uint32_t r, uint32_t lcx, uint32_t ccx) __attribute__((always_inline)) -> int {
        uint32_t bl = lcx;
        if (band) {
            bl = lcx - vc_band_row_start(r - 1, band_ql, band_lanes);
            if (bl >= band_lanes) { oob = true; return 0; }
        }
        const int v = vc_packed_cell(rows32 + (uint64_t)(r - 1) * rstride + bl * nds, ccx, cpl);
        return dtj ? vc_dt_cell(v, r, g) : v;
    };
    // T at (row r, column col); (lcx, ccx) = the lane and the cell inside it of column col (col >= 1)
    auto Tat = [&](uint32_t r, uint32_t col, uint32_t lcx, uint32_t ccx) __attribute__((always_inline)) -> int {
        if (r == 0) return nw ? 0 : -(int)col * g;
        if (col == 0) return nw ? (int)c0[r - 1] : 0;
        const uint32_t b = (r - 1) >> 3, s = b & (NS - 1u);
        const uint32_t wls = (wlp >> (8u * s)) & 0xFFu;
        if (b >= clow && lcx - wls < W) {
            const uint32_t o = (lcx - wls) * nds * 4u, rowin = (r - 1) & 7u;
            const uint32_t bb = *lds_at(s, rowin, o + ccx);
            const uint32_t an = *reinterpret_cast<const uint16_t*>(lds_at(s, rowin, o + cpl));
            const int v = (int)(short)an + (int)((bb - an) & 0xFFu);
            return dtj ? vc_dt_cell(v, r, g) : v;
        }
        return cell_mem(r, lcx, ccx);
    };
    auto rec_at = [&](uint32_t r) __attribute__((always_inline)) -> uint4 {                 // r >= 1
        const uint32_t b = (r - 1) >> 3;
        if (b >= clow) return *reinterpret_cast<const uint4*>(recl + (b & (NS - 1u)) * 1024u + ((r - 1) & 7u) * 16u);
        return a.dp.rec[nb + r - 1];
    };
    auto gmask = [&](unsigned long long mm) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(mm >> gbase) & ((1u << TL) - 1u); };

    bool walking = valid && end != 0;
    bool gredo = false;
    uint32_t gi = end >> 16, gj = end & 0xFFFF, gnout = 0;
    bool govf = false, gbroken = false;
    int gT = 0;
    uint4 grec = make_uint4(0, 0, 0, 0);
    // lane and cell of column gj (gj >= 1), kept beside gj: the walk only ever steps one column to the left
    uint32_t glc = 0, gcc = 0;
    if (walking && gj) { const uint32_t ci = gj - 1; glc = (uint32_t)(((float)ci + 0.5f) * rcpl); gcc = ci - glc * cpl; }

    // bring the blocks [max(0, bg - NS + 1), bg] of the walk's row in (those not there yet); the window of a block sits where the
    // straight line from the walk's cell to the origin (global) or the diagonal (local) crosses the block's top row
    auto maintain = [&]() __attribute__((always_inline)) {
        const bool act = walking && gi != 0;
        const uint32_t bg = act ? (gi - 1) >> 3 : 0u;
        const uint32_t tgt = bg >= NS - 1u ? bg - (NS - 1u) : 0u;
        if (act && (clow == 0x7FFFFFFFu || bg + 1u < clow)) { clow = bg + 1u; lland = 0x7FFFFFFFu; }       // nothing cached at or below the row's block: start over there
        for (;;) {
            const bool want = act && clow > tgt;
            if (!__any(want)) break;
            const uint32_t b = want ? clow - 1u : 0u;
            // window of block b
            const uint32_t rt = min(8u * b + 8u, gi);
            uint32_t cp;
            if (nw) cp = (uint32_t)((float)gj * (float)rt / (float)max(gi, 1u));
            else cp = gj > gi - rt ? gj - (gi - rt) : 0u;
            const uint32_t ci = (cp ? cp - 1u : 0u) + cpl / 2u;
            const uint32_t L = min((uint32_t)(((float)ci + 0.5f) * rcpl), 63u);
            const uint32_t lo = band ? vc_band_row_start(8u * b, band_ql, band_lanes) : 0u, hi = lo + rlanes - W;
            const uint32_t wl = min(max(L >= W - 1u ? L - (W - 1u) : 0u, lo), hi);
#pragma unroll
            for (uint32_t s = 0; s < NS; ++s) {
                const bool me = want && (b & (NS - 1u)) == s;
                if (__any(me)) {
                    // lane gl of the group brings row gl of the block: piece t of its window with instruction t, then its record
                    const uint32_t r = 8u * b + 1u + gl;
                    const uint32_t* const src = rows32 + (uint64_t)(r - 1) * rstride + (wl - lo) * nds;
#pragma unroll
                    for (uint32_t t = 0; t < NCH; ++t)
                        if (me && r <= nrows) vc_glds16(src + t * 4u, smem + ROWB + s * (NCH * 1024u) + t * 1024u);
                    if (me && r <= nrows) vc_glds16(a.dp.rec + nb + r - 1, smem + RECB + s * 1024u);
                    if (me) wlp = (wlp & ~(0xFFu << (8u * s))) | (wl << (8u * s));
                }
            }
            if (want) clow = b;
        }
    };
    // reads of a block that may still be on its way wait for everything this wave has asked for
    auto landed = [&](uint32_t lowest_row) __attribute__((always_inline)) {
        const uint32_t b = lowest_row ? (lowest_row - 1) >> 3 : 0u;
        const bool wait = walking && clow != 0x7FFFFFFFu && max(b, clow) < lland;
        if (__any(wait)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lland = clow;
        }
    };

    maintain();
    landed(0);
    if (walking) {
        gT = Tat(gi, gj, glc, gcc);
        if (gi) grec = rec_at(gi);
        if (oob) { gredo = true; walking = false; }           // (cannot happen: the band ends on the end cell)
    }
    for (;;) {
        if (walking && (nw ? (gi == 0 && gj == 0) : (gT == -(int)gj * g))) walking = false;
        if (!__any(walking)) break;
        maintain();
        const bool need = walking;
        // ---- one general step at (gi, gj): k_tracew's step D
        uint32_t pi_ = 0, pj_ = 0;
        int hv = 0;
        bool found = false, have_v = false;
        uint32_t v_pi = 0; int v_hv = 0;
        oob = false;
        // column gj - 1: lane and cell
        uint32_t plc = glc, pcc = gcc;
        if (gj >= 2) { if (gcc) pcc = gcc - 1; else { plc = glc - 1; pcc = cpl - 1; } }
        const bool isovf = ((grec.x >> 8) & VC_RF_OVF) != 0;
        const uint32_t np = (need && gi != 0) ? (isovf ? grec.z : ((grec.x >> 16) & 0xFF)) : 0u;
        int sc = 0;
        if (np && gj != 0) sc = ((a.b.bases[so + gj - 1] == (grec.x & 0xFF)) ? m : n) - g;
        const bool isd = gl < ND_, isv = gl >= ND_ + 1 && gl < 2 * ND_ + 1;
        // the first pass's candidates: every row they touch is at most 0xFFFF rows up; the lowest block any lane will read decides the wait
        {
            uint32_t lowest = gi;
            if (need && gi != 0 && !isovf) {
                const uint32_t d0 = grec.y & 0xFFFF, d1 = grec.y >> 16, d2 = grec.z & 0xFFFF;
                const uint32_t dm = max(d0, max(np > 1 ? d1 : 0u, np > 2 ? d2 : 0u));
                lowest = gi > dm ? gi - dm : 0u;
            }
            landed(lowest);
        }
        bool hmatch = false;                                         // the horizontal move matches (lane 7 of the group looks at T[gi][gj-1] in the first pass)
        const bool ish = gl == TL - 1;
        for (uint32_t base = 0; __any((!found && base < np) || (base == 0 && need)); base += ND_) {
            const uint32_t p = base + (gl & 3u);
            const bool act = !found && (isd || isv) && p < np && (!isd || gj != 0) && (isd || !have_v);
            const bool act_h = ish && need && base == 0 && gj != 0;
            uint32_t delta = 0;
            if (act) {
                if (isovf) { delta = a.dp.ovf[eb + grec.y + p]; if (a.kept && (delta & 0x8000u)) delta &= 0x7Fu; }
                else {
                    const uint32_t wsel = p < 2 ? grec.y : (p < 4 ? grec.z : grec.w);
                    delta = (p & 1) ? (wsel >> 16) : (wsel & 0xFFFF);
                }
            }
            const uint32_t pr = gi - delta;                              // (the horizontal candidate: delta == 0, the walk's own row)
            if (base != 0 || isovf) landed(act ? pr : gi);           // (later passes and long lists: rows the first look did not cover)
            int cv = 0;
            if (act || act_h) { const bool left = isd || ish; cv = Tat(pr, left ? gj - 1 : gj, left ? plc : glc, left ? pcc : gcc); }
            const bool match = (act && gT == cv + (isd ? sc : g)) || (act_h && gT == cv);
            const uint32_t gm = gmask(__ballot(match));
            const uint32_t dm = gm & ((1u << ND_) - 1u), vm = (gm >> (ND_ + 1)) & ((1u << ND_) - 1u);
            if (base == 0) hmatch = (gm >> (TL - 1)) != 0;
            const bool take_d = !found && dm != 0, take_v = !found && dm == 0 && vm != 0 && !have_v;
            const uint32_t src = gbase + (dm ? (uint32_t)__ffs((int)dm) - 1 : (vm ? ND_ + 1 + (uint32_t)__ffs((int)vm) - 1 : 0u));
            const uint32_t s_pr = (uint32_t)__shfl((int)pr, (int)src, 64);
            // (a matching candidate's value follows from the match: T = gT - (score - g) through a diagonal, gT - g through a vertical move)
            if (take_d) { pi_ = s_pr; pj_ = gj - 1; hv = gT - sc; found = true; }
            else if (take_v) { v_pi = s_pr; v_hv = gT - g; have_v = true; }
        }
        if (need && !found && have_v) { pi_ = v_pi; pj_ = gj; hv = v_hv; found = true; }
        {
            // a candidate of this step lay outside the stored band: the decision cannot be taken from what is stored -- the alignment goes on the redo list
            const bool goob = gmask(__ballot(oob && need)) != 0;
            if (need && goob) { gredo = true; walking = false; }
        }
        if (need && !found && gj != 0 && hmatch) { pi_ = gi; pj_ = gj - 1; hv = gT; found = true; }
        if (need && !gredo) {
            if (!found) { gbroken = true; walking = false; }
            else if (gnout >= a.PC) { govf = true; walking = false; }
            else {
                if (gl == 0) out[gnout] = ((gi == pi_ ? 0u : gi) << 16) | (gj == pj_ ? 0u : gj);
                gnout++;
                if (pj_ != gj) { glc = plc; gcc = pcc; }
                const bool moved_row = pi_ != gi;
                gi = pi_; gj = pj_; gT = hv;
                if (moved_row) grec = gi ? rec_at(gi) : make_uint4(0, 0, 0, 0);
            }
        }
    }
    if (valid && gl == 0) {
        if (gredo) { gnout = 0; a.redo_out[atomicAdd(a.redo_out_n, 1u)] = job; }
        else if (gbroken) { vc_fail(a.b, w, VC_WIN_INVALID, 17, gi); gnout = 0; }
        else if (govf) { vc_fail(a.b, w, VC_WIN_OVERFLOW, 5, gnout); gnout = 0; }
        a.npairs[pj] = gnout;
        if (a.cursor) a.cursor[slot] = k | (gredo ? 0x80000000u : 0u);
    }
    {   // statistics: moves of the wave (there are no speculated moves or rounds here)
        uint32_t s0 = (valid && gl == 0) ? gnout : 0u;
#pragma unroll
        for (int o = TL; o < 64; o <<= 1) s0 += (uint32_t)__shfl_xor((int)s0, o, 64);
        if (lane == 0) atomicAdd(vc_stat_slot(a.stat) + 4, (unsigned long long)s0);
    }
    return valid && gredo;
}

VC_KL __global__ __launch_bounds__(64) void k_traceb(VcTraceArgs a) {
    VC_LATENCY_KERNEL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t grp = (uint32_t)vc_lane() / 8u;
    const uint32_t njobs = a.nslots * a.group;
    const bool redo = a.redo_list != nullptr;
    uint32_t job = blockIdx.x * 8u + grp;
    bool valid = job < njobs;
    if (redo) {                                               // second pass: the jobs the first one gave up on
        const uint32_t nr = *a.redo_n;
        if (blockIdx.x * 8u >= nr) return;
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
    (void)vc_traceb_body(a, smem, job, slot, k, pj, valid, whole);
}
