// development probe: what a large hipMalloc, its first touch and a second touch cost on this box (cold start of vc_reserve / vc_submit)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(nullptr);
    for (double gib : {1.0, 8.0, 32.0, 96.0}) {
        const size_t bytes = (size_t)(gib * (1ull << 30));
        void* p = nullptr;
        double t0 = now();
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("%5.0f GiB: hipMalloc failed\n", gib); continue; }
        double t1 = now();
        (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipMemset(p, 1, bytes); (void)hipDeviceSynchronize();
        double t3 = now();
        (void)hipFree(p);
        double t4 = now();
        printf("%5.0f GiB: hipMalloc %.3f s, first memset %.3f s, second memset %.3f s, hipFree %.3f s\n", gib, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    // many pieces instead of one
    { const int N = 400; void* ps[N]; double t0 = now(); for (int i = 0; i < N; ++i) (void)hipMalloc(&ps[i], (size_t)240 << 20); double t1 = now();
      for (int i = 0; i < N; ++i) (void)hipFree(ps[i]); double t2 = now();
      printf("400 x 240 MiB: hipMalloc %.3f s, hipFree %.3f s\n", t1 - t0, t2 - t1); }
    return 0;
}
