// Practical HBM bandwidth of the box: write-only, read-only, copy (hipcc --offload-arch=gfx950 -O3 tools/hbm_bw.hip -o /tmp/hbm_bw)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_write(uint4* p, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_uint4(v, v, v, v);
}
__global__ void k_write4(uint32_t* p, size_t n, uint32_t v) {      // dword stores, 256 B per wave instruction (k_fwd's pattern)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = v;
}
__global__ void k_read(const uint4* p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t a = 0;
    for (; i < n; i += st) { uint4 x = p[i]; a += x.x ^ x.y ^ x.z ^ x.w; }
    if (a == 0x12345678u) *out = a;
}
__global__ void k_copy(const uint4* s, uint4* d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) d[i] = s[i];
}
int main() {
    const size_t bytes = 16ull << 30;
    void *a, *b; uint32_t* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto fn, double moved) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 3; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %.2f TB/s\n", name, moved * 3 / (ms * 1e-3) / 1e12);
    };
    for (int g : {256 * 8, 256 * 32}) {
        printf("grid %d x 256\n", g);
        run("write16", [&] { k_write<<<g, 256>>>((uint4*)a, bytes / 16, 7); }, (double)bytes);
        run("write4", [&] { k_write4<<<g, 256>>>((uint32_t*)a, bytes / 4, 7); }, (double)bytes);
        run("read16", [&] { k_read<<<g, 256>>>((const uint4*)a, bytes / 16, o); }, (double)bytes);
        run("copy", [&] { k_copy<<<g, 256>>>((const uint4*)a, (uint4*)b, bytes / 16); }, 2.0 * bytes);
    }
    return 0;
}
