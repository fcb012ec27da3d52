// development probe (gfx950): what a raw buffer store does with lanes whose offset lies outside the descriptor's range, and whether the
// scalar offset takes part in the range check -- the banded store of k_fwd_dt (vc_fwd_dt.h) leans on both.
//   hipcc --offload-arch=gfx950 -O3 tools/buffer_probe.hip -o /tmp/buffer_probe && /tmp/buffer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u3 __attribute__((ext_vector_type(3)));

__global__ void probe(uint32_t* out, uint32_t bytes, uint32_t bs, uint32_t soff) {
    const uint32_t lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    // band lanes bs .. bs + 15 write 12 bytes each at (lane - bs) * 12; the others carry an offset no descriptor holds
    const uint32_t voff = (lane - bs) < 16u ? (lane - bs) * 12u : 0x80000000u;
    u3 d = {lane | 0xA0000000u, lane | 0xB0000000u, lane | 0xC0000000u};
    __builtin_amdgcn_raw_buffer_store_b96(d, rs, voff, soff, 0);
}

int main() {
    const size_t N = 4096;
    uint32_t* d = nullptr;
    hipMalloc(&d, N * 4);
    std::vector<uint32_t> h(N);
    struct { uint32_t bytes, bs, soff; const char* what; } cases[] = {
        {192, 5, 0, "num_records 192, soffset 0"},
        {192, 5, 192, "num_records 192, soffset 192 (is soffset range-checked?)"},
        {4096, 40, 384, "num_records 4096, soffset 384"},
        {4096, 60, 0, "num_records 4096, band start 60 (lanes 60..63 only)"},
    };
    for (auto& c : cases) {
        hipMemset(d, 0, N * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c.bytes, c.bs, c.soff);
        hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost);
        size_t first = N, last = 0, cnt = 0;
        for (size_t i = 0; i < N; ++i) if (h[i]) { if (first == N) first = i; last = i; cnt++; }
        printf("%-60s: %zu dwords written, first at dword %zu (lane %u), last at %zu (lane %u)\n", c.what, cnt, first == N ? 0 : first,
               first == N ? 0 : h[first] & 0xFFFF, last, h[last] & 0xFFFF);
    }
    hipFree(d);
    return 0;
}
