// k_fwd_dt: the forward pass of global (NW) alignments on byte-packed rows, DOUBLY tilted -- round 6's form of vc_fwd_body
// (sisd_alignment_engine.cpp:118-254 Initialize, :292-360 Linear).  Included by vc_kernels.h behind the helpers it shares with k_fwd.
//
// k_fwd works on T[i][j] = H[i][j] - j*g: the horizontal pass of sisd:347-349 is then a plain prefix maximum, but every row still pays
// the vertical move with an add per register (T[p][j] + g).  Here the matrix is tilted along the rows as well:
//
//      T''[i][j] = H[i][j] - (i + j) * g           (i: row number in DP order, 1-based; the virtual row is row 0)
//
// With X''[j] = max over the predecessors p of ( T''[p][j] + (i - 1 - p) * (-g) ) the recurrence of sisd:315-360 becomes
//
//      T''[i][j] = prefix-max over j of  max( X''[j-1] + (score - 2g),  X''[j] )
//
// * the row directly above (p = i - 1) enters X'' as it stands in the registers -- no add: 5 of a row's 51 vector instructions at 10
//   columns per lane; a predecessor further up pays one packed add of the uniform (i - 1 - p) * (-g) when it is merged (it paid nothing
//   in k_fwd: 0.4 merges per row against 1.0 rows);
// * column 0: c0''[i] = H[i][0] - i*g = max_p ( c0''[p] + (i - 1 - p) * (-g) ) -- unchanged along a chain of rows, so the value the lane
//   scan starts from and the cell left of column 1 are loop invariants of a run of "plain" rows;
// * every value is >= 0 (H[i][j] >= (depth_i + j) * g and depth_i <= i), so the scores are UNSIGNED 16-bit numbers: 0 is the identity of
//   the maximum, which is what DPP hands a lane that has no source lane (bound_ctrl) -- the lane scan needs no seed register and no copy,
//   and the range is 65 535 instead of 32 767: m*cols + (rows + cols)*(-g) must fit (vc_dt_ok; config E's 6 000 x 1 100 does);
// * inside a row neighbours still differ by 0 .. max(m, n) - 2g, so the stored form is k_fwd's: low byte per cell + the lane's first cell
//   as a 16-bit anchor (vc_pack_row), banded (vc_band_row_start).  A reader adds row * g to a rebuilt cell and walks on T as before
//   (vc_dt_cell; job_type carries VC_JOB_DT), column 0 is stored as the true H[i][0] as before: the backtrack is the same code.
//
// The scalar half of a row.  k_fwd spends 37 scalar instructions and 13 branches per row on record decode, flag tests, ring-slot and
// band arithmetic (PMC) -- as many issue slots as the vector half.  Here
// * the banded store is a RAW BUFFER store: descriptor = the job's band rows, scalar offset = the row, vector offset = the lane's place in
//   the band or 0x80000000 for the 48 lanes outside it (the range check drops those; tools/buffer_probe.hip) -- no exec-mask writes, no
//   asm block around a store (the gfx950 store-data hazard of round 5 is the compiler's to see again), one s_add per row; the band's
//   vector offset is worked out once per VC_BAND_ROWS rows in the loop that walks the band blocks;
// * a row whose only predecessor is the row above, that is no sink and is not read back from the stored matrix ("fast": one test of the
//   record word) touches nothing but its base, its keep bit and the loop counter.
#pragma once

typedef unsigned short vc_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pku_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(vc_us2, a), __builtin_bit_cast(vc_us2, b)));
}
// d.lo = a.lo ; d.hi = max(a.hi, a.lo)
__device__ __forceinline__ uint32_t pku_max_hi_with_lo(uint32_t a) {
    uint32_t d;
    asm("v_pk_max_u16 %0, %1, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
// d.lo = max(a.lo, b.hi) ; d.hi = max(a.hi, b.hi)
__device__ __forceinline__ uint32_t pku_max_bcast_hi(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// lane l takes v of lane l - k inside the DPP pattern `ctrl`; a lane without a source takes 0 (the identity of an unsigned maximum)
#define VC_DPP_Z(v, ctrl, rmask, bc) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xF, (bc)))

// Does a global alignment of 64 * cpl columns against nrows rows stay inside unsigned 16 bits in the doubly tilted form, with byte-packed
// rows?  (0 <= T'' <= max(m, 0) * columns + (rows + columns) * (-g); the diagonal term score - 2g must not be negative, or a cell could
// dip below 0 on the way to its maximum.)
__host__ __device__ inline bool vc_dt_ok(long long m, long long n, long long g, long long nrows, long long cpl) {
    if (g >= 0 || n - 2 * g < 0 || m - 2 * g < 0) return false;
    const long long cols = 64 * cpl + 1, mm = m > 0 ? m : 0;
    return mm * cols + (nrows + cols + 8) * (-g) <= 65000 && vc_row_packed((int)m, (int)n, (int)g, (int)cpl);
}

template <int CPL, int RING, bool KEPT>
__device__ __forceinline__ uint32_t vc_fwd_dt(const VcFwdArgs& a, uint32_t* ring_raw, const VcJob& jb) {
    static_assert(KEPT || (RING & (RING - 1)) == 0, "ring slots are taken with a mask");
    static_assert(CPL < 32, "the lean classes keep k_fwd");
    constexpr int ND = CPL / 2, NDS = vc_nds(CPL);
    const int lane = vc_lane();
    // (everything the row loop branches on is wave-uniform; say so, or a value the compiler cannot prove uniform -- a load behind a load --
    // turns the scalar half of the loop into vector code and the buffer store into a waterfall loop)
    auto uni = [](uint32_t v) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const bool redo = uni(jb.redo ? 1u : 0u) != 0;
    const uint32_t job = uni(jb.job), slot = uni(jb.slot), k = uni(jb.k);
    const uint32_t w = a.w0 + slot;
    if (a.do_init && lane == 0 && !redo) { a.job_type[job] = 255; a.job_end[job] = 0; a.tie_cnt[job] = 0; }
    if (a.b.status[w] != VC_WIN_OK) return VC_FWD_NONE;
    const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
    if (k >= ns) return VC_FWD_NONE;
    const uint64_t so = a.b.seq_off[s0 + k];
    const uint32_t len = uni((uint32_t)(a.b.seq_off[s0 + k + 1] - so));
    if (a.fold ? len > 64u * CPL : vc_cpl_for(len) != (uint32_t)CPL) return VC_FWD_NONE;      // another width class handles this sequence
    if (a.mode != 0) {                                          // a launch of this kernel holds global alignments only (the host knows)
        const uint32_t L = (uint32_t)(a.b.seq_off[s0 + 1] - a.b.seq_off[s0]);
        const bool nw = a.mode == 1 && (k == 0 || vc_full_span(a.b.seq_begin[s0 + k], a.b.seq_end[s0 + k], L));
        if (!nw) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 28, a.k0); return VC_FWD_NONE; }
    }
    const int m = a.m, n = a.n, g = a.g;
    const uint32_t nrows = uni(a.dp.nrows[slot]);
    const uint64_t nb = (uint64_t)slot * a.NC;
    {
        const bool ok = vc_dt_ok(m, n, g, nrows, CPL) && len <= 64u * CPL && len > 0 && nrows > 0 && !(a.dp.flags[slot] & 1u);
        if (!ok) {
            // outside this form's envelope: not an error -- the job keeps type 255 and k_fwd_wide (int32 lanes) takes it
            if (len == 0 || nrows == 0 || (a.dp.flags[slot] & 1u)) { if (lane == 0) vc_fail(a.b, w, VC_WIN_INVALID, 3, (a.dp.flags[slot] & 1u) ? 1 : 2); }
            else if (a.wcols == 0) { if (lane == 0) vc_fail(a.b, w, VC_WIN_OVERFLOW, 27, nrows); }
            return VC_FWD_NONE;
        }
    }
    if (lane == 0 && !redo) {
        a.job_type[job] = (uint8_t)(1u | VC_JOB_DT);
        unsigned long long* st = vc_stat_slot(a.stat);
        atomicAdd(st + 0, (unsigned long long)nrows * len);
        atomicAdd(st + 1, (unsigned long long)nrows);
    }
    if (lane == 0 && redo) atomicAdd(vc_stat_slot(a.stat) + 7, 1ull);

    // profile of my columns for the four usual bases: score - 2g per cell (packed pairs); columns past the sequence end never match
    uint32_t pfA[ND], pfC[ND], pfG[ND], pfT[ND];
    const uint32_t ng = (uint32_t)(-g);
    const int mt = m - 2 * g, nt = n - 2 * g;
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        const uint32_t i0 = lane * CPL + 2 * q, i1 = i0 + 1;
        const uint32_t b0 = i0 < len ? a.b.bases[so + i0] : 0xFFu;
        const uint32_t b1 = i1 < len ? a.b.bases[so + i1] : 0xFFu;
        auto sc = [&](uint32_t x) { return ((uint32_t)((b0 == x) ? mt : nt) & 0xFFFFu) | ((uint32_t)((b1 == x) ? mt : nt) << 16); };
        pfA[q] = sc('A'); pfC[q] = sc('C'); pfG[q] = sc('G'); pfT[q] = sc('T');
    }

    uint32_t* const hrow0 = a.hmat + (uint64_t)job * a.hstride;
    const bool band = uni((a.band && !redo) ? 1u : 0u) != 0;
    // the job's band rows behind a buffer descriptor: [row][vc_band_lanes(CPL)][NDS dwords]
    const uint64_t bjd = vc_band_job_dwords(a.hstride);
    const uintptr_t bbase = reinterpret_cast<uintptr_t>(a.bmat + (uint64_t)job * bjd);
    const uint32_t bb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bbase), bb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bbase >> 32));
    const uint32_t bbytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bjd * 4ull < 0xFFFFF000ull ? bjd * 4ull : 0xFFFFF000ull));
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)bb_hi << 32) | bb_lo), 0, (int)bbytes, 0x00020000);
    const uint32_t band_ql = (uint32_t)__builtin_amdgcn_readfirstlane((int)vc_band_slope(len, nrows, CPL));
    if (band && lane == 0) a.band_par[job] = band_ql;
    constexpr uint32_t BL = vc_band_lanes(CPL, !KEPT);                          // lanes of a band row in this width class and phase (build: 80 columns, at least 8 lanes)
    constexpr uint32_t TLB = NDS * 4u, TBB = BL * TLB;                          // a lane's bytes in a band row, a band row
    const uint32_t lane_tlb = (uint32_t)lane * TLB;
    int16_t* const c0p_out = a.c0 + (uint64_t)job * a.NC;
    const uint16_t* const ovfp = a.dp.ovf + (uint64_t)slot * a.EC;

    // end cell (sisd:353-355): the value compared is the singly tilted one, T = T'' + i*g, as a 32-bit number
    int best = VC_INT_MIN;
    uint32_t best_row = 0, ntie = 0;
    const uint32_t lane_e = (len - 1) / CPL, c_e = (len - 1) % CPL;
    uint32_t far_reads = 0;

    uint32_t acc[ND];                         // between iterations: T'' of the row just finished
#pragma unroll
    for (int q = 0; q < ND; ++q) acc[q] = 0;
    uint32_t c0prev = 0;                      // c0'' of the row just finished
    uint32_t c0vec = 0;                       // lane t: c0'' << 16 of the latest row r with (r - 1) % 64 == t
    uint32_t vcol0 = 0, vc0l = 0;             // c0'' << 16 of the row in work: in every lane (the lane scan's carry starts from it) / in lane 0 only (the cell left of column 1)

    uint4 myrec = make_uint4(0, 0, 0, 0), nextrec = make_uint4(0, 0, 0, 0);
    if ((uint32_t)lane < nrows) nextrec = a.dp.frec[nb + lane];
    constexpr uint32_t rowdw = NDS * 64;                         // dwords per whole stored row
    const uint32_t loff = (uint32_t)lane * NDS;                  // my dword offset inside a whole row

    auto ring_slot_merge = [&](uint32_t slot_, uint32_t c0lane, uint32_t kadd, uint32_t& c0m) __attribute__((always_inline)) {
        const uint32_t* rp = ring_raw + slot_ * (ND * 64) + lane;
        uint32_t hp[ND];
#pragma unroll
        for (int q = 0; q < ND; ++q) hp[q] = rp[q * 64];
        const uint32_t c0p = ((uint32_t)__builtin_amdgcn_readlane((int)c0vec, (int)c0lane) >> 16) + kadd;
        const uint32_t k2 = kadd * 0x10001u;
#pragma unroll
        for (int q = 0; q < ND; ++q) { const uint32_t t = pk_add(hp[q], k2); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(acc[q]) : "v"(t)); }
        c0m = max(c0m, c0p);
    };
    // a listed predecessor known to sit in the ring: `e` is its forward entry (plain ring: the distance; kept-row ring: 0x8000 | slot << 8 | distance)
    auto ring_merge = [&](uint32_t i, uint32_t e, uint32_t& c0m) __attribute__((always_inline)) {
        if (KEPT) { const uint32_t d = e & 0x7Fu; ring_slot_merge((e >> 8) & 7u, (i - d - 1u) & 63u, (d - 1u) * ng, c0m); }
        else ring_slot_merge((i - e) & (RING - 1), (i - e - 1u) & 63u, (e - 1u) * ng, c0m);
    };
    auto set_col0 = [&](uint32_t c0m) __attribute__((always_inline)) {
        const uint32_t hi = c0m << 16;
        asm volatile("v_mov_b32 %0, %1" : "=v"(vcol0) : "s"(hi));
        vc0l = lane == 0 ? vcol0 : 0u;
    };

    // the DP of one row once X'' stands in acc and c0'' in c0m.  FAST: the row is no sink and is not stored whole for a later reader
    // (both known from the one test that found it fast)
    uint32_t vband = 0x80000000u;                              // my byte offset inside a band row, or out of every range
    auto row_tail = [&](const uint32_t r0, const uint32_t c0m, const uint32_t i, const uint32_t ri, const bool FAST) __attribute__((always_inline)) {
        // diagonal: cell j-1 of X'' (shift right by one cell; the hole is filled by the left lane's last cell, lane 0 takes column 0)
        const uint32_t left = vc0l | VC_DPP_Z(acc[ND - 1], 0x138, 0xF, true);
        uint32_t P[ND];
#pragma unroll
        for (int q = 0; q < ND; ++q) P[q] = __builtin_amdgcn_alignbit(acc[q], q == 0 ? left : acc[q - 1], 16);
        // (the adds are asm volatile so that the four arms stay branches: left to itself the compiler selects the profile registers with
        // v_cndmask -- five more vector instructions per row)
        const uint32_t bi = (r0 >> 24) & 7u;
#define VC_DT_PROF(pf) _Pragma("unroll") for (int q = 0; q < ND; ++q) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(P[q]) : "v"(pf[q]))
        if (bi < 2) {
            if (bi == 0) { VC_DT_PROF(pfA); } else { VC_DT_PROF(pfC); }
        } else if (bi == 2) { VC_DT_PROF(pfG); }
        else if (FAST || bi == 3) { VC_DT_PROF(pfT); }
#undef VC_DT_PROF
        else {
            // a row byte outside A / C / G / T (an N, an IUPAC letter: rare): my columns' bases come back from memory -- five registers that
            // only this arm would need are worth more to the other kernels on the SIMD
            const uint32_t x = r0 & 0xFF;
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const uint32_t i0 = lane * CPL + 2 * q, i1 = i0 + 1;
                const uint32_t b0 = i0 < len ? a.b.bases[so + i0] : 0xFFu;
                const uint32_t b1 = i1 < len ? a.b.bases[so + i1] : 0xFFu;
                P[q] = pk_add(P[q], ((uint32_t)((b0 == x) ? mt : nt) & 0xFFFFu) | ((uint32_t)((b1 == x) ? mt : nt) << 16));
            }
        }
        // vertical: X'' itself
#pragma unroll
        for (int q = 0; q < ND; ++q) P[q] = pku_max(P[q], acc[q]);
        // horizontal pass (sisd:347-349): prefix maximum, in-lane then across lanes.  The scan runs on the raw dword of the lane's last pair:
        // as an unsigned 32-bit number it orders by its high half (the lane's running maximum); only the high half of the result is used
        P[0] = pku_max_hi_with_lo(P[0]);
#pragma unroll
        for (int q = 1; q < ND; ++q) P[q] = pku_max_bcast_hi(pku_max_hi_with_lo(P[q]), P[q - 1]);
        uint32_t sc = P[ND - 1];
        sc = max(sc, VC_DPP_Z(sc, 0x111, 0xF, true));
        sc = max(sc, VC_DPP_Z(sc, 0x112, 0xF, true));
        sc = max(sc, VC_DPP_Z(sc, 0x114, 0xF, true));
        sc = max(sc, VC_DPP_Z(sc, 0x118, 0xF, true));
        sc = max(sc, VC_DPP_Z(sc, 0x142, 0xA, false));
        sc = max(sc, VC_DPP_Z(sc, 0x143, 0xC, false));
        const uint32_t carry = max(vcol0, VC_DPP_Z(sc, 0x138, 0xF, true));       // column 0 enters as T''[i][0]
#pragma unroll
        for (int q = 0; q < ND; ++q) acc[q] = pku_max_bcast_hi(P[q], carry);

        if (!FAST && (r0 & (VC_RF_SINK << 8))) {                 // sisd:353-355
            // (the empty asm keeps this a chain of selects: left to itself the compiler folds it into ONE indexed load of acc[c_e / 2], and an
            // array that is indexed by a register lives in scratch memory -- every write of acc then became a scratch store, at 16 / 20 columns
            // per lane, and config E ran 14 % slower than on k_fwd)
            uint32_t hv = acc[0];
#pragma unroll
            for (int q = 1; q < ND; ++q) { uint32_t t = acc[q]; asm("" : "+v"(t)); hv = (c_e / 2 == (uint32_t)q) ? t : hv; }
            uint32_t vv = (c_e & 1) ? (hv >> 16) : (hv & 0xFFFFu);
            vv = (uint32_t)__builtin_amdgcn_readlane((int)vv, (int)lane_e);
            const int v = (int)vv + (int)i * g;
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

        // keep the row: registers (acc), column 0, LDS ring, HBM
        c0prev = c0m;
        {   // lane ri keeps this row's column 0: v_cndmask under a lane mask made on the scalar side (v_writelane takes ONE scalar operand,
            // and the value and the lane are two; written as `lane == ri ? ..` the compare is a vector instruction of its own)
            const unsigned long long lm = 1ull << ri;
            asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(c0vec) : "v"(vcol0), "s"(lm));
        }
        __builtin_amdgcn_wave_barrier();          // one wave per workgroup: its LDS operations retire in order -- only keep the compiler from reordering
        if (!KEPT || (r0 & (VC_RF_KEEP << 8))) {
            uint32_t* wp = ring_raw + (KEPT ? ((r0 >> 27) & 7u) : (i & (RING - 1))) * (ND * 64) + lane;
#pragma unroll
            for (int q = 0; q < ND; ++q) wp[q * 64] = acc[q];
        }
        uint32_t wv[NDS];
        vc_pack_row<ND, NDS>(acc, wv);
        if (!band || (!FAST && (r0 & (VC_RF_FULL << 8)))) {       // whole row: always without the band; with it only where a later row reads the row back
            uint32_t* hr = hrow0 + (uint64_t)(i - 1u) * rowdw + loff;
            if constexpr (NDS == 2) *reinterpret_cast<uint2*>(hr) = make_uint2(wv[0], wv[1]);
            else if constexpr (NDS == 4) *reinterpret_cast<uint4*>(hr) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            else if constexpr (NDS == 3) { struct __attribute__((packed, aligned(4))) u3 { uint32_t a, b, c; }; *reinterpret_cast<u3*>(hr) = u3{wv[0], wv[1], wv[2]}; }
            else {
#pragma unroll
                for (int t = 0; t < NDS; ++t) hr[t] = wv[t];
            }
        }
        if (band) {
            const uint32_t sband = (i - 1u) * TBB;             // byte offset of this row among the job's band rows (scalar)
            typedef uint32_t vc_u2v __attribute__((ext_vector_type(2)));
            typedef uint32_t vc_u3v __attribute__((ext_vector_type(3)));
            typedef uint32_t vc_u4v __attribute__((ext_vector_type(4)));
            if constexpr (NDS == 2) { const vc_u2v d = {wv[0], wv[1]}; __builtin_amdgcn_raw_buffer_store_b64(d, brs, vband, sband, 0); }
            else if constexpr (NDS == 3) { const vc_u3v d = {wv[0], wv[1], wv[2]}; __builtin_amdgcn_raw_buffer_store_b96(d, brs, vband, sband, 0); }
            else {
                const vc_u4v d = {wv[0], wv[1], wv[2], wv[3]};
                __builtin_amdgcn_raw_buffer_store_b128(d, brs, vband, sband, 0);
                if constexpr (NDS == 5) __builtin_amdgcn_raw_buffer_store_b32(wv[4], brs, vband + 16u, sband, 0);
                else if constexpr (NDS == 6) { const vc_u2v e = {wv[4], wv[5]}; __builtin_amdgcn_raw_buffer_store_b64(e, brs, vband + 16u, sband, 0); }
                else if constexpr (NDS == 7) { const vc_u3v e = {wv[4], wv[5], wv[6]}; __builtin_amdgcn_raw_buffer_store_b96(e, brs, vband + 16u, sband, 0); }
                static_assert(NDS <= 7, "stored row form of the classes below 32 columns per lane");
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // a row that is not "fast": X'' and c0'' from its predecessors (the row above, when it is one of them, stands in acc / c0prev already)
    auto row_general = [&](const uint32_t r0, const uint32_t i, const uint32_t ri) __attribute__((always_inline)) -> uint32_t {
        uint32_t c0m = c0prev;
        // "hop": ONE predecessor, and it is not the row above but sits in the ring (42 % of the rows of a grown graph: the order keeps aligned
        // groups together, so a chain hops over the other members).  X'' is that row plus its distance term: no clear, no maximum
        constexpr uint32_t HOP_MASK = ((VC_RF_PREV | VC_RF_SLOW) << 8) | (0xFFu << 16), HOP_PAT = 1u << 16;
        if ((r0 & HOP_MASK) == HOP_PAT) {
            const uint32_t e = __builtin_amdgcn_readlane(myrec.y, ri) & 0xFFFFu;
            const uint32_t d = KEPT ? (e & 0x7Fu) : e;
            const uint32_t slot_ = KEPT ? ((e >> 8) & 7u) : ((i - e) & (RING - 1));
            const uint32_t kadd = (d - 1u) * ng, k2 = kadd * 0x10001u;
            const uint32_t* rp = ring_raw + slot_ * (ND * 64) + lane;
            uint32_t hp[ND];
#pragma unroll
            for (int q = 0; q < ND; ++q) hp[q] = rp[q * 64];
#pragma unroll
            for (int q = 0; q < ND; ++q) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(acc[q]) : "v"(hp[q]), "s"(k2));
            return ((uint32_t)__builtin_amdgcn_readlane((int)c0vec, (int)((i - d - 1u) & 63u)) >> 16) + kadd;
        }
        if (!(r0 & (VC_RF_PREV << 8))) {
            c0m = 0;
#pragma unroll
            for (int q = 0; q < ND; ++q) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[q]));
        }
        const uint32_t nq = (r0 >> 16) & 0xFF;
        if (r0 & (VC_RF_SLOW << 8)) {
            // general path: the virtual row 0 analytically, a recent row from the LDS ring, an older one back from the stored matrix in HBM;
            // long lists come from VcDp::ovf
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
                    if (KEPT && (delta & 0x8000u)) { ring_merge(i, delta, c0m); continue; }
                }
                const uint32_t pr = i - delta;
                const uint32_t kadd = (delta - 1u) * ng;
                if (pr == 0) {                                   // H[0][j] = j*g: T''[0][j] = 0, i - 1 rows further down
                    const uint32_t k2 = kadd * 0x10001u;
#pragma unroll
                    for (int q = 0; q < ND; ++q) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(acc[q]) : "v"(k2));
                    c0m = max(c0m, kadd);
                } else if (!KEPT && delta <= (uint32_t)RING) {
                    ring_merge(i, delta, c0m);
                } else {
                    uint32_t hA[ND];
                    __threadfence_block();                       // my own earlier stores must have landed
                    const uint32_t* hr = hrow0 + (uint64_t)(pr - 1) * rowdw + loff;
                    uint32_t wv[NDS];
#pragma unroll
                    for (int t = 0; t < NDS; ++t) wv[t] = hr[t];
                    vc_unpack_row<ND, NDS>(wv, hA);
                    far_reads++;
                    uint32_t cA;
                    if (delta <= 64) cA = (uint32_t)__builtin_amdgcn_readlane((int)c0vec, (int)((pr - 1) & 63)) >> 16;
                    else cA = (uint32_t)((int)__builtin_amdgcn_readfirstlane((int)c0p_out[pr - 1]) - (int)pr * g);
                    const uint32_t k2 = kadd * 0x10001u;
#pragma unroll
                    for (int q = 0; q < ND; ++q) { const uint32_t t = pk_add(hA[q], k2); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(acc[q]) : "v"(t)); }
                    c0m = max(c0m, cA + kadd);
                }
            }
        } else if (nq) {
            // every listed predecessor sits in the LDS ring: straight-line, one test per further in-edge
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
        return c0m;
    };

    // fast: the row above is the only predecessor (PLAIN), no sink, not stored whole, one of A / C / G / T -- one masked compare of the record word
    constexpr uint32_t FAST_MASK = ((VC_RF_SLOW | VC_RF_PLAIN | VC_RF_SINK | VC_RF_FULL) << 8) | (4u << 24);
    constexpr uint32_t FAST_PAT = VC_RF_PLAIN << 8;

    const unsigned long long clk_w0 = wall_clock64(), clk_c0 = clock64();
    for (uint32_t i0 = 1; i0 <= nrows; i0 += 64) {              // blocks of 64 rows: one record fetch, one column-0 flush
        myrec = nextrec;
        {
            const uint32_t r = i0 - 1 + 64 + lane;
            if (r < nrows) nextrec = a.dp.frec[nb + r];
        }
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, nrows - i0 + 1));
        for (uint32_t rb = 0; rb < cnt; rb += VC_BAND_ROWS) {   // the band moves every VC_BAND_ROWS rows (vc_band_row_start)
            if (band) {
                const uint32_t bs = vc_band_row_start(i0 - 1u + rb, band_ql, BL);      // (scalar: row and slope are uniform)
                const uint32_t bl = (uint32_t)lane - bs;
                vband = bl < BL ? lane_tlb - bs * TLB : 0x80000000u;
            }
            const uint32_t re = min(rb + (uint32_t)VC_BAND_ROWS, cnt);
            for (uint32_t ri = rb; ri < re; ++ri) {
                const uint32_t r0 = __builtin_amdgcn_readlane(myrec.x, ri);
                if ((r0 & FAST_MASK) == FAST_PAT) {
                    row_tail(r0, c0prev, i0 + ri, ri, true);
                } else {
                    const uint32_t c0m = row_general(r0, i0 + ri, ri);
                    set_col0(c0m);
                    row_tail(r0, c0m, i0 + ri, ri, false);
                }
            }
        }
        // column 0 of the block just completed, as the true H[i][0] = c0'' + i*g
        if ((uint32_t)lane < cnt) c0p_out[i0 - 1 + lane] = (int16_t)((int)(c0vec >> 16) + (int)(i0 + lane) * g);
        __threadfence_block();
    }
    if (lane == 0 && far_reads && !redo) atomicAdd(vc_stat_slot(a.stat) + 3, (unsigned long long)far_reads);
    {
        const unsigned long long dc = clock64() - clk_c0, dw = wall_clock64() - clk_w0;
        if (lane == 0) { unsigned long long* ck = vc_clk_slot(a.stat); atomicAdd(ck, dc); atomicAdd(ck + 1, dw); }
    }
    if (redo) return VC_FWD_DONE;                               // the end cell, ties and counters stand from the first pass
    const uint32_t end = (best_row << 16) | len;
    uint32_t outcome = VC_FWD_DONE;
    if (ntie > 1 && (a.dp.flags[slot] & 2u)) {                  // tie on a non-reference order: k_resolve decides
        if (lane == 0) { a.tie_cnt[job] = ntie; a.tie_list[atomicAdd(a.tie_n, 1u)] = slot; }
        outcome = VC_FWD_TIE;
    }
    if (lane == 0) a.job_end[job] = end;
    return outcome;
}

// CA <= CB: the two adjacent width classes of a batch share one launch; each alignment takes the narrowest body that holds its sequence
template <int CA, int CB, int RING, bool KEPT>
VC_KL __global__ __launch_bounds__(64) void k_fwd_dt(VcFwdArgs a) {
    __shared__ uint32_t ring_raw[RING * (CB / 2) * 64];
    VcJob jb;
    if (!vc_fwd_pick(a, jb)) return;
    if (CA != CB) {
        const uint32_t w = a.w0 + jb.slot;
        const uint32_t s0 = a.b.win_seq_off[w], ns = a.b.win_seq_off[w + 1] - s0;
        uint32_t cls = CB;
        if (jb.k < ns) cls = vc_cpl_for((uint32_t)(a.b.seq_off[s0 + jb.k + 1] - a.b.seq_off[s0 + jb.k]));
        if (cls == (uint32_t)CA || (a.fold && cls < (uint32_t)CA)) { (void)vc_fwd_dt<CA, RING, KEPT>(a, ring_raw, jb); return; }
    }
    (void)vc_fwd_dt<CB, RING, KEPT>(a, ring_raw, jb);
}
