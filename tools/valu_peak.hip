// Calibration microbenchmark for the VALU roofline of k_fwd (bench.py reads profiles/r2_valu_peak.json made from its output):
// issue rate of the packed-int16 instructions the forward DP is made of (v_pk_max_i16 / v_pk_add_i16, plus the DPP and
// v_perm forms it uses), the dependent-issue latency of the same, and the engine clock the chip sustains under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak.bin && tools/valu_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: 8 independent chains of v_pk_max_i16 / v_pk_add_i16 (issue-bound)
// MODE 1: one dependent chain of the same (latency-bound)
// MODE 2: dependent chain of DPP row_shr max (the cross-lane scan step)
// MODE 3: 8 independent v_perm_b32
// MODE 4: 8 independent v_max_i32 (32-bit reference)
// MODE 5: 8 independent chains of v_pk_max_f16 / v_pk_add_f16   (round 4: does a packed-f16 max-plus issue faster than packed i16?)
// MODE 6: 8 independent chains of v_max_f32 / v_add_f32
// MODE 7: 4 independent chains of v_pk_add_f32 (64-bit register pairs)
// MODE 8: 8 independent v_fma_f32
// MODE 9: 8 independent v_pk_fma_f16
// round 5 -- the opcode classes of k_fwd's row loop, each on its own (roofline.peak_mix weights them by the loop's histogram):
// MODE 10: v_pk_max_i16 only     MODE 11: v_pk_add_u16 only     MODE 12: v_alignbit_b32     MODE 13: v_max_i32_dpp row_shr (8 independent registers)
// MODE 14: v_readlane_b32 / v_writelane_b32 pairs     MODE 15: v_cndmask_b32 / v_mov_b32 / v_and / v_lshl_or (the 32-bit odds and ends)
// MODE 16: scalar ALU only (s_add / s_and / s_max / s_lshl)     MODE 17: 8 packed-int16 and 8 scalar instructions interleaved
template <int MODE>
__global__ __launch_bounds__(256) void k_valu(uint32_t* out, int iters, unsigned long long* cyc, unsigned long long* wall) {
    uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 ^ 0x55, r5 = r0 + 77, r6 = r0 * 11, r7 = r0 + 9;
    const uint32_t c = 0x00010001u * (blockIdx.x & 3) + 0x00020001u;
    unsigned long long q0 = r0 | ((unsigned long long)r1 << 32), q1 = r2 | ((unsigned long long)r3 << 32), q2 = r4 | ((unsigned long long)r5 << 32), q3 = r6 | ((unsigned long long)r7 << 32);
    const unsigned long long qc = c | ((unsigned long long)c << 32);
    const unsigned long long w0 = wall_clock64();          // s_memrealtime: constant 100 MHz
    const unsigned long long t0 = clock64();               // s_memtime: one tick per shader cycle (MI355X_MICROARCH.md)
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(
                "v_pk_max_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                "v_pk_max_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n"
                "v_pk_max_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                "v_pk_max_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 1) {
            asm volatile(
                "v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n"
                "v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n"
                "v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n"
                "v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_add_i16 %0, %0, %1\n"
                : "+v"(r0) : "v"(c));
        } else if (MODE == 2) {
            asm volatile(
                "s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                : "+v"(r0));
        } else if (MODE == 3) {
            asm volatile(
                "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "v"(0x06040200u));
        } else if (MODE == 5) {
            asm volatile(
                "v_pk_max_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n"
                "v_pk_max_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8\n"
                "v_pk_max_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n"
                "v_pk_max_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 6) {
            asm volatile(
                "v_max_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                "v_max_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                "v_max_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                "v_max_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 7) {
            asm volatile(
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc));
        } else if (MODE == 8) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 9) {
            asm volatile(
                "v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n"
                "v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n"
                "v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n"
                "v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 10) {
            asm volatile(
                "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 11) {
            asm volatile(
                "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 12) {
            asm volatile(
                "v_alignbit_b32 %0, %0, %8, 16\n v_alignbit_b32 %1, %1, %8, 16\n v_alignbit_b32 %2, %2, %8, 16\n v_alignbit_b32 %3, %3, %8, 16\n"
                "v_alignbit_b32 %4, %4, %8, 16\n v_alignbit_b32 %5, %5, %8, 16\n v_alignbit_b32 %6, %6, %8, 16\n v_alignbit_b32 %7, %7, %8, 16\n"
                "v_alignbit_b32 %0, %0, %8, 16\n v_alignbit_b32 %1, %1, %8, 16\n v_alignbit_b32 %2, %2, %8, 16\n v_alignbit_b32 %3, %3, %8, 16\n"
                "v_alignbit_b32 %4, %4, %8, 16\n v_alignbit_b32 %5, %5, %8, 16\n v_alignbit_b32 %6, %6, %8, 16\n v_alignbit_b32 %7, %7, %8, 16\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        } else if (MODE == 13) {
            asm volatile(
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_max_i32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                "v_max_i32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_max_i32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        } else if (MODE == 14) {
            asm volatile(
                "v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s21, 5\n v_readlane_b32 s22, %2, 7\n v_writelane_b32 %3, s23, 9\n"
                "v_readlane_b32 s24, %4, 11\n v_writelane_b32 %5, s25, 13\n v_readlane_b32 s26, %6, 15\n v_writelane_b32 %7, s27, 17\n"
                "v_readlane_b32 s21, %0, 19\n v_writelane_b32 %1, s20, 21\n v_readlane_b32 s23, %2, 23\n v_writelane_b32 %3, s22, 25\n"
                "v_readlane_b32 s25, %4, 27\n v_writelane_b32 %5, s24, 29\n v_readlane_b32 s27, %6, 31\n v_writelane_b32 %7, s26, 33\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 16) {                             // scalar ALU alone: does the scalar side keep up with the vector side?
            asm volatile(
                "s_add_u32 s20, s20, 3\n s_and_b32 s21, s21, 0xffff\n s_max_u32 s22, s22, 7\n s_lshl_b32 s23, s23, 1\n"
                "s_add_u32 s24, s24, 3\n s_and_b32 s25, s25, 0xffff\n s_max_u32 s26, s26, 7\n s_lshl_b32 s27, s27, 1\n"
                "s_add_u32 s20, s20, 3\n s_and_b32 s21, s21, 0xffff\n s_max_u32 s22, s22, 7\n s_lshl_b32 s23, s23, 1\n"
                "s_add_u32 s24, s24, 3\n s_and_b32 s25, s25, 0xffff\n s_max_u32 s26, s26, 7\n s_lshl_b32 s27, s27, 1\n"
                ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (MODE == 17) {                             // 8 packed-int16 + 8 scalar instructions interleaved: issued side by side (rate of 16 ~ the slower stream) or in turn?
            asm volatile(
                "v_pk_max_i16 %0, %0, %8\n s_add_u32 s20, s20, 3\n v_pk_add_u16 %1, %1, %8\n s_and_b32 s21, s21, 0xffff\n"
                "v_pk_max_i16 %2, %2, %8\n s_max_u32 s22, s22, 7\n v_pk_add_u16 %3, %3, %8\n s_lshl_b32 s23, s23, 1\n"
                "v_pk_max_i16 %4, %4, %8\n s_add_u32 s24, s24, 3\n v_pk_add_u16 %5, %5, %8\n s_and_b32 s25, s25, 0xffff\n"
                "v_pk_max_i16 %6, %6, %8\n s_max_u32 s26, s26, 7\n v_pk_add_u16 %7, %7, %8\n s_lshl_b32 s27, s27, 1\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c)
                : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (MODE == 15) {
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, vcc\n v_mov_b32 %1, %8\n v_and_b32 %2, %2, %8\n v_lshl_or_b32 %3, %3, 16, %8\n"
                "v_cndmask_b32 %4, %4, %8, vcc\n v_or_b32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
                "v_cndmask_b32 %0, %0, %8, vcc\n v_mov_b32 %1, %8\n v_and_b32 %2, %2, %8\n v_lshl_or_b32 %3, %3, 16, %8\n"
                "v_cndmask_b32 %4, %4, %8, vcc\n v_or_b32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c) : "vcc");
        } else {
            asm volatile(
                "v_max_i32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                "v_max_i32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                "v_max_i32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                "v_max_i32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        }
    }
    const unsigned long long t1 = clock64();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)((q0 + q1 + q2 + q3) >> 32);
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * 4 + threadIdx.x / 64, nw = (size_t)gridDim.x * 4;
        cyc[wv] = t1 - t0; wall[wv] = w1 - w0; wall[nw + wv] = w0; wall[2 * nw + wv] = w1;     // (start and end of the loop on the 100 MHz clock: when was this wave resident?)
    }
}

template <int MODE>
int run(const char* name, int waves_per_simd, int n_cu, int per_iter) {
    const int iters = 200000;
    const int blocks = n_cu * waves_per_simd;          // 256-thread blocks: one wave per SIMD each
    uint32_t* d_out; unsigned long long *d_cyc, *d_wall;
    CHK(hipMalloc(&d_out, (size_t)blocks * 256 * 4)); CHK(hipMalloc(&d_cyc, (size_t)blocks * 4 * 8)); CHK(hipMalloc(&d_wall, (size_t)blocks * 4 * 8 * 3));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1000, d_cyc, d_wall);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_cyc, d_wall);
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc((size_t)blocks * 4), wal((size_t)blocks * 4 * 3);
    CHK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(wal.data(), d_wall, wal.size() * 8, hipMemcpyDeviceToHost));
    // residency: when did the waves start and end their loops (100 MHz ticks -> ms after the first start)?
    const size_t nwv = cyc.size();
    unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
    for (size_t i = 0; i < nwv; ++i) { s_min = std::min(s_min, wal[nwv + i]); s_max = std::max(s_max, wal[nwv + i]); e_min = std::min(e_min, wal[2 * nwv + i]); e_max = std::max(e_max, wal[2 * nwv + i]); }
    size_t late = 0;
    for (size_t i = 0; i < nwv; ++i) late += wal[nwv + i] > s_min + 100000ull;                 // started more than 1 ms after the first wave
    wal.resize(nwv);
    // shader clock over the loop, wave by wave: shader cycles per 100 MHz tick; the median wave
    std::vector<double> mhz(cyc.size());
    for (size_t i = 0; i < cyc.size(); ++i) mhz[i] = wal[i] ? 100.0 * (double)cyc[i] / (double)wal[i] : 0.0;
    std::sort(mhz.begin(), mhz.end());
    const double sclk_mhz = mhz[mhz.size() / 2];
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[cyc.size() / 2];
    const double insts = (double)iters * per_iter;
    // clock64() ticks at a constant 100 MHz on gfx9; the engine clock follows from the instruction count when the issue
    // rate per cycle is known -- report both views: instructions per microsecond per SIMD and ticks
    const double inst_per_us_per_simd = insts * waves_per_simd / (ms * 1e3);
    // Cycles of a SIMD per wave-instruction issued, from the AGGREGATE rate (all waves of a SIMD over the kernel's duration) at the
    // measured shader clock.  Not from a wave's own loop time: the SIMD serves its oldest ready wave first, so with independent
    // instruction streams the waves of a SIMD finish one after the other (first_end_ms is the single-wave time at every occupancy,
    // last_end_ms the kernel's) -- the median wave says nothing about the SIMD's throughput.
    const double cyc_per_inst = sclk_mhz / inst_per_us_per_simd;
    printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_insts\": %.0f, \"inst_per_us_per_simd\": %.1f, \"ns_per_wave_inst\": %.3f, \"clock64_ticks_median\": %.0f, "
           "\"sclk_mhz\": %.0f, \"simd_cycles_per_inst\": %.2f, "
           "\"last_start_ms\": %.3f, \"first_end_ms\": %.3f, \"last_end_ms\": %.3f, \"waves_started_late\": %.3f}\n",
           name, waves_per_simd, ms, insts, inst_per_us_per_simd, ms * 1e6 / insts, med, sclk_mhz, cyc_per_inst,
           (s_max - s_min) / 1e5, (e_min - s_min) / 1e5, (e_max - s_min) / 1e5, (double)late / (double)nwv);
    (void)hipFree(d_out); (void)hipFree(d_cyc); (void)hipFree(d_wall);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d}\n", p.gcnArchName, n_cu, p.clockRate);
    if (argc > 1 && std::string(argv[1]) == "quick") {      // bench.py: only the rate the forward DP is priced against (~0.2 s)
        for (int rep = 0; rep < 3; ++rep)                    // (the clock needs a moment after the bench's last kernel: best of three)
            for (int w : {4, 8}) if (run<0>("pk_i16_independent", w, n_cu, 16)) return 1;
        // vector and scalar instructions interleaved 1 : 1 (k_fwd's row is ~55 vector and ~50 scalar / branch instructions): what a SIMD issues in all
        for (int w : {5, 8}) if (run<17>("pk16_and_salu_interleaved", w, n_cu, 16) || run<16>("salu_independent", w, n_cu, 16)) return 1;
        // the opcode classes of the forward DP's row loop, each at the occupancy it runs at (5 waves per SIMD) and at 8
        for (int w : {5, 8}) {
            if (run<10>("class_pk16_max", w, n_cu, 16) || run<11>("class_pk16_add", w, n_cu, 16) || run<4>("class_i32_max_add", w, n_cu, 16) ||
                run<15>("class_i32_misc", w, n_cu, 16) || run<3>("class_perm", w, n_cu, 16) || run<12>("class_alignbit", w, n_cu, 16) ||
                run<13>("class_dpp", w, n_cu, 16) || run<14>("class_lane", w, n_cu, 16)) return 1;
        }
        return 0;
    }
    for (int w : {1, 2, 4, 8}) if (run<0>("pk_i16_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 4}) if (run<4>("i32_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 4}) if (run<3>("v_perm_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 2, 4, 8}) if (run<5>("pk_f16_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 2, 4, 8}) if (run<6>("f32_max_add_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 2, 4, 8}) if (run<7>("pk_add_f32_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 4, 8}) if (run<8>("fma_f32_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 4, 8}) if (run<9>("pk_fma_f16_independent", w, n_cu, 16)) return 1;
    for (int w : {1, 2, 4}) if (run<1>("pk_i16_dependent", w, n_cu, 16)) return 1;
    for (int w : {1, 4}) if (run<2>("dpp_max_dependent", w, n_cu, 8)) return 1;
    for (int w : {1, 4, 5, 8}) if (run<16>("salu_independent", w, n_cu, 16) || run<17>("pk16_and_salu_interleaved", w, n_cu, 16)) return 1;
    for (int w : {1, 2, 4, 5, 8}) {
        if (run<10>("class_pk16_max", w, n_cu, 16) || run<11>("class_pk16_add", w, n_cu, 16) || run<4>("class_i32_max_add", w, n_cu, 16) ||
            run<15>("class_i32_misc", w, n_cu, 16) || run<3>("class_perm", w, n_cu, 16) || run<12>("class_alignbit", w, n_cu, 16) ||
            run<13>("class_dpp", w, n_cu, 16) || run<14>("class_lane", w, n_cu, 16)) return 1;
    }
    return 0;
}
