// Probe of global_load_lds on gfx950 for k_traceb's row cache (tools/glds_probe.hip; hipcc --offload-arch=gfx950 -O2):
//   1. lanes masked off by exec leave their LDS slot alone, active lanes land at base + lane * size (not packed) -- measured: true for
//      4 and 16 bytes; the 12-byte form lands at base + lane * 16 (the probe expects that),
//   2. a 16-byte (and 12-byte) piece may start at any dword of global memory,
//   3. s_waitcnt vmcnt(0) is enough for the wave that issued the load to read the data back with ds_read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int SIZE>
__global__ __launch_bounds__(64) void k(const uint32_t* src, uint32_t* dst, uint64_t mask, uint32_t shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    uint32_t* s32 = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t i = lane; i < 2048 / 4; i += 64) s32[i] = 0xDEAD0000u | i;
    __syncthreads();
    for (uint32_t s = 0; s < 2; ++s) {
        const bool me = ((mask >> lane) & 1) && ((lane >> 3) & 1) == s;
        if (__any(me)) {
            // every lane reads SIZE bytes starting at dword (lane * 7 + shift) of src
#if defined(__HIP_DEVICE_COMPILE__)
            if (me) {
                if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + lane * 7 + shift), (lds_ptr_t)(smem + s * 1024), 16, 0, 0);
                else if constexpr (SIZE == 12) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + lane * 7 + shift), (lds_ptr_t)(smem + s * 1024), 12, 0, 0);
                else __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + lane * 7 + shift), (lds_ptr_t)(smem + s * 1024), 4, 0, 0);
            }
#endif
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t i = lane; i < 2048 / 4; i += 64) dst[i] = s32[i];
}

template <int SIZE>
static int run(uint64_t mask, uint32_t shift) {
    std::vector<uint32_t> h(64 * 7 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (uint32_t)i;
    uint32_t *d_src, *d_dst;
    if (hipMalloc(&d_src, h.size() * 4) != hipSuccess || hipMalloc(&d_dst, 2048) != hipSuccess) return 1;
    if (hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(k<SIZE>, dim3(1), dim3(64), 2048, 0, d_src, d_dst, mask, shift);
    std::vector<uint32_t> o(512);
    if (hipMemcpy(o.data(), d_dst, 2048, hipMemcpyDeviceToHost) != hipSuccess) { printf("size %d: kernel failed\n", SIZE); return 1; }
    int bad = 0;
    const int DW = SIZE / 4, ST = SIZE == 12 ? 4 : DW;          // dwords a lane writes, dwords between the lanes' pieces
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int d = 0; d < DW; ++d) {
                const int at = s * 256 + lane * ST + d;
                const bool active = ((mask >> lane) & 1) && ((lane >> 3) & 1) == s;
                const uint32_t want = active ? 0x1000u + (uint32_t)(lane * 7 + shift + d) : (0xDEAD0000u | (uint32_t)at);
                if (o[at] != want) { if (bad < 6) printf("  size %d mask %016llx shift %u: slot %d lane %d dword %d = %08x, expected %08x\n", SIZE, (unsigned long long)mask, shift, s, lane, d, o[at], want); ++bad; }
            }
    printf("size %2d mask %016llx shift %u: %s (%d wrong)\n", SIZE, (unsigned long long)mask, shift, bad ? "DIFFERENT" : "as expected", bad);
    (void)hipFree(d_src); (void)hipFree(d_dst);
    return bad != 0;
}

int main() {
    int bad = 0;
    for (uint32_t shift = 0; shift < 4; ++shift)
        for (uint64_t mask : {~0ull, 0x00FF00FF00FF00FFull, 0xF0F0F0F00F0F0F0Full, 0x8000000000000001ull}) {
            bad += run<16>(mask, shift);
            bad += run<12>(mask, shift);
            bad += run<4>(mask, shift);
        }
    printf(bad ? "glds_probe: FAILED\n" : "glds_probe: ok\n");
    return bad != 0;
}
