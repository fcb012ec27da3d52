// Large host buffers of the file readers and the window builder (vc_io.cpp, vc_windows.cpp): hundreds of megabytes that are
// written once, by several threads, and read once.  No zero fill (a std::vector would write every byte before the real data does)
// and transparent huge pages where the kernel grants them on request: first touch of 256 MB is 128 faults instead of 65 536.
#pragma once
#include <sys/mman.h>

#include <cstddef>
#include <cstdlib>

struct VcHostBuf {
    char* p = nullptr; size_t n = 0;
    VcHostBuf() = default;
    VcHostBuf(const VcHostBuf&) = delete;
    VcHostBuf& operator=(const VcHostBuf&) = delete;
    ~VcHostBuf() { release(); }
    void alloc(size_t k) {
        release();
        n = k;
        static const int mode = getenv("VC_HOSTBUF") ? atoi(getenv("VC_HOSTBUF")) : 1;      // development: 0 = malloc, 1 = mmap + huge pages, 2 = mmap
        if (k >= kMapFrom && mode) {
            mapped = (k + kHuge - 1) & ~(kHuge - 1);
            void* m = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
#ifdef MADV_HUGEPAGE
                if (mode == 1) (void)madvise(m, mapped, MADV_HUGEPAGE);
#endif
                p = (char*)m;
                return;
            }
            mapped = 0;
        }
        p = (char*)malloc(k ? k : 1);
    }
    void release() {
        if (mapped) munmap(p, mapped); else free(p);
        p = nullptr; n = 0; mapped = 0;
    }
    const char* data() const { return p; }
    char* data() { return p; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
private:
    static constexpr size_t kHuge = 2u << 20, kMapFrom = 8u << 20;
    size_t mapped = 0;
};
