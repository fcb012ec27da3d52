// Overlap alignment on the device: the global alignment with path that the reference
// obtains from edlib for overlaps that come without a CIGAR (src/overlap.cpp:205-220:
// edlibAlign(q, t, EDLIB_MODE_NW, EDLIB_TASK_PATH) -> edlibAlignmentToCigar(EDLIB_CIGAR_STANDARD)).
//
// PARITY UNPINNED, and it cannot be pinned here: edlib 1.2.7 is not vendored (CMake fetches it), and an
// optimal alignment is not unique -- which of the co-optimal paths comes out depends on edlib's own
// traceback.  What this file guarantees, and what tests/test_align.py checks against a CPU DP, is that the
// path is a valid global alignment whose cost equals the unit-cost edit distance.  Ties are broken
// diagonal first, then insertion (query base only), then deletion (target base only), walking from the end.
//
// Kernels (gfx950, wave64):
//   k_aln_fwd    one wave per overlap.  Rows = query bases, columns = target bases in tiles of 2048 (64 lanes x
//                32 cells as packed int16 pairs).  Same arithmetic as the POA forward kernel: tilted scores
//                (H + jj, unit gap), so the horizontal pass is a prefix maximum; the row above stays in
//                registers (a sequence is a chain: one predecessor).  Scores are relative to the row's value at
//                the tile boundary, so int16 suffices for any length; boundary columns are int32 in HBM.  No band: the full n x m matrix costs
//                ~1e8 cells for a 10 kb overlap, 40 us of this chip.  Stored per row, tile and lane: the first
//                cell as int16 and the 31 steps to its right neighbours (each 0, 1 or 2) as 2-bit fields --
//                12 bytes per 32 cells, 0.375 B/cell, otherwise the stores would exceed HBM bandwidth.
//   k_aln_trace  one thread per overlap, re-derives each move from the stored scores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vechat_hip.h"

namespace {

constexpr int kCPL = 32, kND = 16, kTile = 64 * kCPL;     // cells per lane, packed dwords per lane, columns per tile
constexpr int kRowDw = 64 * 3;                           // stored dwords per row and tile

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pmax(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b))); }
__device__ __forceinline__ uint32_t padd(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s2)(__builtin_bit_cast(s2, a) + __builtin_bit_cast(s2, b))); }
__device__ __forceinline__ uint32_t psub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s2)(__builtin_bit_cast(s2, a) - __builtin_bit_cast(s2, b))); }
__device__ __forceinline__ uint32_t pminu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u2, a), __builtin_bit_cast(u2, b))); }
__device__ __forceinline__ uint32_t pdup(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }
__device__ __forceinline__ uint32_t hi_with_lo(uint32_t a) { uint32_t d; asm("v_pk_max_i16 %0, %1, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t bcast_hi(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
#define ALN_DPP(v, old, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xF, false)
#define ALN_INT_MIN (-2147483647 - 1)

struct AlnArgs {
    uint32_t n_jobs;
    const uint64_t* q_off; const uint8_t* q;      // query pieces, oriented as they align
    const uint64_t* t_off; const uint8_t* t;      // target pieces
    const uint8_t* skip;                          // [n_jobs] 1: outside the envelope, treated as an empty pair
    const uint64_t* mat_off;                      // [n_jobs] dword offset of a job's stored matrix
    uint32_t* mat;                                // rows x tiles x 64 lanes x 3 dwords
    const uint64_t* bnd_off; int32_t* bnd;        // [n_jobs] offsets; per job [tiles][n+1]: H at each tile's left boundary column
    int32_t* dist;                                // [n_jobs] edit distance
    const uint64_t* ops_off; uint8_t* ops;        // [n_jobs] offsets; n+m op bytes per job (0 M, 1 I, 2 D), end first
    uint32_t* n_ops;                              // [n_jobs]
};

__global__ __launch_bounds__(64) void k_aln_fwd(AlnArgs a) {
    const uint32_t job = blockIdx.x;
    if (job >= a.n_jobs) return;
    const int lane = threadIdx.x & 63;
    const uint8_t* q = a.q + a.q_off[job];
    const uint8_t* t = a.t + a.t_off[job];
    const bool skip = a.skip[job] != 0;
    const uint32_t n = skip ? 0u : (uint32_t)(a.q_off[job + 1] - a.q_off[job]), m = skip ? 0u : (uint32_t)(a.t_off[job + 1] - a.t_off[job]);
    const uint32_t ntiles = (m + kTile - 1) / kTile;
    uint32_t* mat = a.mat + a.mat_off[job];
    int32_t* bnd = a.bnd + a.bnd_off[job];
    const uint32_t ones = 0x00010001u;
    int last = 0;
    // Scores are kept RELATIVE to the row's value at the tile's left boundary column ts:
    //     R[i][jj] = H[i][ts + jj] - H[i][ts] + jj   in [0, 2 * 2048],
    // so int16 holds them whatever the overlap's length; the boundary columns themselves (B[ct][i] = H[i][ts]) are
    // int32 in HBM.  With d = B[i-1] - B[i] (-1, 0 or +1): diagonal R' + 1 + s + d, vertical R' - 1 + d, horizontal = prefix
    // maximum starting from R[i][0] = 0.
    for (uint32_t ct = 0; ct < ntiles; ++ct) {
        const int ts = (int)(ct * kTile);
        const int32_t* bin = bnd + (uint64_t)ct * (n + 1);                  // this tile's boundary column (ct > 0)
        int32_t* bout = bnd + (uint64_t)(ct + 1) * (n + 1);                 // the next tile's
        uint32_t sbp[kND];
#pragma unroll
        for (int k = 0; k < kND; ++k) {
            const uint32_t j0 = ct * kTile + lane * kCPL + 2 * k, j1 = j0 + 1;
            const uint32_t b0 = j0 < m ? t[j0] : 0xFFu, b1 = j1 < m ? t[j1] : 0xFFu;
            sbp[k] = b0 | (b1 << 16);
        }
        uint32_t acc[kND];
#pragma unroll
        for (int k = 0; k < kND; ++k) acc[k] = 0;                          // row 0: R = 0 everywhere
        int bprev = -ts;                                                    // B[ct][0] = H[0][ts]
        for (uint32_t i0 = 1; i0 <= n; i0 += 64) {
            const uint32_t cnt = min(64u, n - i0 + 1);
            // this block's query bases and boundary scores: lane r holds row i0 + r
            const uint32_t myq = (uint32_t)lane < cnt ? q[i0 - 1 + lane] : 0u;
            int mybnd = -(int)(i0 + lane);                                  // first tile: H[i][0] = -i
            if (ct != 0 && (uint32_t)lane < cnt) mybnd = bin[i0 + lane];
            int outv = 0;                                                   // lane r: H[i0 + r][ts + 2048]
            for (uint32_t ri = 0; ri < cnt; ++ri) {
                const uint32_t i = i0 + ri;
                const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)myq, ri);
                const int bcur = __builtin_amdgcn_readlane(mybnd, ri);
                const int d = bprev - bcur;
                const uint32_t cd = pdup(1 + d), cv = pdup(d - 1);
                const uint32_t xx = x * 0x10001u;
                const uint32_t left = (uint32_t)ALN_DPP((int)acc[kND - 1], 0, 0x138, 0xF);      // lane 0: R[i-1][0] = 0
                uint32_t P[kND];
#pragma unroll
                for (int k = 0; k < kND; ++k) {
                    const uint32_t sh = __builtin_amdgcn_alignbit(acc[k], k == 0 ? left : acc[k - 1], 16);
                    const uint32_t dg = psub(padd(sh, cd), pminu(sbp[k] ^ xx, ones));           // match: + 1 + d, mismatch: + d
                    P[k] = pmax(dg, padd(acc[k], cv));                                           // vertical: - 1 + d
                }
                P[0] = hi_with_lo(P[0]);
#pragma unroll
                for (int k = 1; k < kND; ++k) P[k] = bcast_hi(hi_with_lo(P[k]), P[k - 1]);
                int sc = (int)P[kND - 1] >> 16;
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x111, 0xF));
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x112, 0xF));
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x114, 0xF));
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x118, 0xF));
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x142, 0xA));
                sc = max(sc, ALN_DPP(sc, ALN_INT_MIN, 0x143, 0xC));
                int carry = ALN_DPP(sc, ALN_INT_MIN, 0x138, 0xF);
                carry = max(carry, 0);                                       // R[i][0] = 0
                const uint32_t cc = pdup(carry);
#pragma unroll
                for (int k = 0; k < kND; ++k) acc[k] = pmax(P[k], cc);
                // H at the tile's last column: the next tile's boundary value
                const int endv = __builtin_amdgcn_readlane((int)acc[kND - 1] >> 16, 63) - kTile + bcur;
                outv = ((uint32_t)lane == ri) ? endv : outv;
                if (i == n && ct == ntiles - 1) {
                    // the cell (n, m): lane and slot of column m in this tile
                    const uint32_t jj = m - ct * kTile, l = (jj - 1) / kCPL, c = (jj - 1) % kCPL;
                    uint32_t hv = acc[0];
#pragma unroll
                    for (int k = 1; k < kND; ++k) hv = (c / 2 == (uint32_t)k) ? acc[k] : hv;
                    const int v = (c & 1) ? ((int)hv >> 16) : (int)(short)(hv & 0xFFFF);
                    last = __builtin_amdgcn_readlane(v, l) - (int)jj + bcur;
                }
                bprev = bcur;
                // stored form: anchor | pairs 1..4, pairs 5..12, pairs 13..15 | step 1
                uint32_t w0, w1 = 0, w2 = 0;
                {
                    const uint32_t e0 = psub(acc[0], acc[0] << 16);                            // lo: first cell, hi: step 1
                    w0 = e0 & 0xFFFFu;
                    w2 = (e0 >> 16) << 12;
#pragma unroll
                    for (int k = 1; k < kND; ++k) {
                        const uint32_t e = psub(acc[k], __builtin_amdgcn_alignbit(acc[k], acc[k - 1], 16));
                        const uint32_t c4 = (e | (e >> 14)) & 0xFu;                             // step 2k | step 2k+1 << 2
                        if (k <= 4) w0 |= c4 << (12 + 4 * k);
                        else if (k <= 12) w1 |= c4 << (4 * (k - 5));
                        else w2 |= c4 << (4 * (k - 13));
                    }
                }
                uint32_t* dst = mat + ((uint64_t)(i - 1) * ntiles + ct) * kRowDw + lane * 3;
                dst[0] = w0; dst[1] = w1; dst[2] = w2;
            }
            if (ct + 1 < ntiles && (uint32_t)lane < cnt) bout[i0 + lane] = outv;
        }
    }
    if (lane == 0) a.dist[job] = n == 0 ? (int32_t)m : (m == 0 ? (int32_t)n : -last);
}

// H[i][j] from the stored tiles (i >= 1, j >= 1): the relative score plus the row's boundary value of that tile
__device__ __forceinline__ int aln_cell(const uint32_t* mat, const int32_t* bnd, uint32_t n, uint32_t ntiles, uint32_t i, uint32_t j) {
    const uint32_t ct = (j - 1) / kTile, jj = j - ct * kTile, l = (jj - 1) / kCPL, c = (jj - 1) % kCPL;
    const uint32_t* w = mat + ((uint64_t)(i - 1) * ntiles + ct) * kRowDw + l * 3;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    int v = (int)(short)(w0 & 0xFFFF);
    if (c >= 1) {
        v += (int)((w2 >> 12) & 3u);
        // steps 2..c are the first c-1 two-bit fields of the stream w0[16..31] w1 w2[0..11]
        const unsigned long long stream = (unsigned long long)(w0 >> 16) | ((unsigned long long)w1 << 16) | ((unsigned long long)(w2 & 0xFFFu) << 48);
        const uint32_t nf = c - 1;
        const unsigned long long x = nf ? (stream & (~0ull >> (64 - 2 * nf))) : 0ull;
        unsigned long long s = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
        s = (s & 0x0F0F0F0F0F0F0F0Full) + ((s >> 4) & 0x0F0F0F0F0F0F0F0Full);
        v += (int)((s * 0x0101010101010101ull) >> 56);
    }
    return v - (int)jj + (ct == 0 ? -(int)i : bnd[(uint64_t)ct * (n + 1) + i]);
}

__global__ void k_aln_trace(AlnArgs a) {
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= a.n_jobs) return;
    const uint8_t* q = a.q + a.q_off[job];
    const uint8_t* t = a.t + a.t_off[job];
    const bool skip = a.skip[job] != 0;
    const uint32_t n = skip ? 0u : (uint32_t)(a.q_off[job + 1] - a.q_off[job]), m = skip ? 0u : (uint32_t)(a.t_off[job + 1] - a.t_off[job]);
    const uint32_t ntiles = (m + kTile - 1) / kTile;
    const uint32_t* mat = a.mat + a.mat_off[job];
    const int32_t* bnd = a.bnd + a.bnd_off[job];
    uint8_t* ops = a.ops + a.ops_off[job];
    auto H = [&](uint32_t i, uint32_t j) -> int {
        if (i == 0) return -(int)j;
        if (j == 0) return -(int)i;
        return aln_cell(mat, bnd, n, ntiles, i, j);
    };
    uint32_t i = n, j = m, k = 0;
    int h = H(i, j);
    while (i || j) {
        if (i && j) {
            const int d = H(i - 1, j - 1);
            if (h == d - (q[i - 1] != t[j - 1])) { ops[k++] = 0; --i; --j; h = d; continue; }
        }
        if (i) {
            const int v = H(i - 1, j);
            if (h == v - 1) { ops[k++] = 1; --i; h = v; continue; }
        }
        const int l = H(i, j - 1);                          // h == l - 1 by construction
        ops[k++] = 2; --j; h = l;
    }
    a.n_ops[job] = k;
}

template <typename T>
bool dalloc(std::vector<void*>& list, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return false;
    list.push_back(q);
    *p = (T*)q;
    return true;
}

thread_local std::string g_err;
int fail(const char* m) { g_err = m; return VC_ERR_HIP; }

// The stored matrices are tens of GB per call; mapping and unmapping that much takes seconds, so the buffer is
// kept between calls (grow-only, per device) and released by vc_align_release().
struct MatCache { int device = -1; uint32_t* p = nullptr; uint64_t dwords = 0; };
MatCache g_mat;
uint32_t* mat_buffer(int device, uint64_t dwords) {
    if (g_mat.p && g_mat.device == device && g_mat.dwords >= dwords) return g_mat.p;
    if (g_mat.p) { (void)hipFree(g_mat.p); g_mat = MatCache{}; }
    void* q = nullptr;
    if (hipMalloc(&q, std::max<uint64_t>(dwords, 1) * 4) != hipSuccess) return nullptr;
    g_mat.device = device; g_mat.p = (uint32_t*)q; g_mat.dwords = dwords;
    return g_mat.p;
}

}  // namespace

extern "C" {

const char* vc_align_last_error(void) { return g_err.c_str(); }

void vc_align_release(void) {
    if (g_mat.p) { (void)hipSetDevice(g_mat.device); (void)hipFree(g_mat.p); }
    g_mat = MatCache{};
}

// Aligns every (query piece, target piece) pair globally; writes edlib-standard CIGAR strings (M / I / D) back to
// back into `cigar` (NUL after each) with offsets in cigar_off[n+1], and the edit distances.
int vc_align(int device, const vc_align_batch* b, char* cigar, uint64_t cigar_cap, uint64_t* cigar_off, int32_t* edit_distance) {
    if (!b || !cigar_off || !edit_distance || (b->n && (!b->q_off || !b->t_off || !b->q || !b->t))) return VC_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed (the overlap aligner has no CPU path)");
    const uint32_t n = b->n;
    cigar_off[0] = 0;
    if (n == 0) return VC_OK;
    size_t free_b0 = 0, total_b0 = 0;
    (void)hipMemGetInfo(&free_b0, &total_b0);
    // an overlap whose stored matrix alone would not fit the device is reported (distance -1, empty CIGAR)
    std::vector<uint8_t> skip(n, 0);
    for (uint32_t k = 0; k < n; ++k) {
        const uint64_t ql = b->q_off[k + 1] - b->q_off[k], tl = b->t_off[k + 1] - b->t_off[k];
        if (ql >= (1ull << 31) || tl >= (1ull << 31) || ql * ((tl + kTile - 1) / kTile) * kRowDw * 4 > total_b0 / 3) skip[k] = 1;
    }
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const uint64_t budget = (uint64_t)((free_b + (g_mat.device == device ? g_mat.dwords * 4 : 0)) * 0.5);
    std::vector<void*> fixed;
    uint8_t *d_q = nullptr, *d_t = nullptr;
    uint64_t *d_qo = nullptr, *d_to = nullptr;
    uint8_t* d_skip = nullptr;
    const uint64_t qbytes = b->q_off[n], tbytes = b->t_off[n];
    auto cleanup = [&](std::vector<void*>& l) { for (void* p : l) (void)hipFree(p); l.clear(); };
    if (!dalloc(fixed, &d_q, qbytes) || !dalloc(fixed, &d_t, tbytes) || !dalloc(fixed, &d_qo, (size_t)n + 1) || !dalloc(fixed, &d_to, (size_t)n + 1) || !dalloc(fixed, &d_skip, (size_t)n)) {
        cleanup(fixed); return fail("hipMalloc failed");
    }
    (void)hipMemcpy(d_q, b->q, qbytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_t, b->t, tbytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_qo, b->q_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_to, b->t_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_skip, skip.data(), n, hipMemcpyHostToDevice);
    uint64_t out = 0;
    std::string cg;
    for (uint32_t k0 = 0; k0 < n;) {
        // a chunk of overlaps whose matrices fit the budget
        std::vector<uint64_t> mat_off, bnd_off, ops_off;
        uint64_t mat_dw = 0, bnd_n = 0, ops_n = 0;
        uint32_t k1 = k0;
        while (k1 < n) {
            const uint64_t ql = skip[k1] ? 0 : b->q_off[k1 + 1] - b->q_off[k1], tl = skip[k1] ? 0 : b->t_off[k1 + 1] - b->t_off[k1];
            const uint64_t need = ql * ((tl + kTile - 1) / kTile) * kRowDw;
            if (k1 > k0 && (mat_dw + need) * 4 > budget) break;
            mat_off.push_back(mat_dw); bnd_off.push_back(bnd_n); ops_off.push_back(ops_n);
            mat_dw += need; bnd_n += (ql + 1) * ((tl + kTile - 1) / kTile + 1); ops_n += ql + tl;
            ++k1;
        }
        const uint32_t nj = k1 - k0;
        std::vector<void*> tmp;
        AlnArgs a{};
        uint64_t *d_mo = nullptr, *d_bo = nullptr, *d_oo = nullptr;
        a.mat = mat_buffer(device, mat_dw);
        if (!a.mat || !dalloc(tmp, &a.bnd, bnd_n) || !dalloc(tmp, &a.ops, ops_n) || !dalloc(tmp, &a.n_ops, nj) ||
            !dalloc(tmp, &a.dist, nj) || !dalloc(tmp, &d_mo, nj) || !dalloc(tmp, &d_bo, nj) || !dalloc(tmp, &d_oo, nj)) {
            cleanup(tmp); cleanup(fixed); return fail("hipMalloc failed (overlap too large for the device memory budget?)");
        }
        (void)hipMemcpy(d_mo, mat_off.data(), (size_t)nj * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_bo, bnd_off.data(), (size_t)nj * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_oo, ops_off.data(), (size_t)nj * 8, hipMemcpyHostToDevice);
        a.n_jobs = nj; a.q_off = d_qo + k0; a.q = d_q; a.t_off = d_to + k0; a.t = d_t; a.skip = d_skip + k0;
        a.mat_off = d_mo; a.bnd_off = d_bo; a.ops_off = d_oo;
        hipLaunchKernelGGL(k_aln_fwd, dim3(nj), dim3(64), 0, 0, a);
        hipLaunchKernelGGL(k_aln_trace, dim3((nj + 63) / 64), dim3(64), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { cleanup(tmp); cleanup(fixed); return fail("alignment kernels failed"); }
        std::vector<uint8_t> ops(ops_n);
        std::vector<uint32_t> nops(nj);
        (void)hipMemcpy(ops.data(), a.ops, ops_n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(nops.data(), a.n_ops, (size_t)nj * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(edit_distance + k0, a.dist, (size_t)nj * 4, hipMemcpyDeviceToHost);
        cleanup(tmp);
        for (uint32_t k = 0; k < nj; ++k) {
            cg.clear();
            if (skip[k0 + k]) edit_distance[k0 + k] = -1;
            const uint8_t* o = ops.data() + ops_off[k];
            for (uint32_t p = nops[k]; p > 0;) {                    // stored end first
                const uint8_t op = o[p - 1];
                uint32_t run = 0;
                while (p > 0 && o[p - 1] == op) { ++run; --p; }
                cg += std::to_string(run);
                cg += "MID"[op];
            }
            if (out + cg.size() + 1 > cigar_cap) { cleanup(fixed); return fail("cigar buffer too small"); }
            std::memcpy(cigar + out, cg.c_str(), cg.size() + 1);
            out += cg.size() + 1;
            cigar_off[k0 + k + 1] = out;
        }
        k0 = k1;
    }
    cleanup(fixed);
    return VC_OK;
}

}  // extern "C"
