// TEST INFRASTRUCTURE -- the drop-in boundary, compiled.
//
// The reference's Window befriends its accelerated batch class (`friend class CUDABatchProcessor`,
// src/window.hpp:61-63, under CUDA_ENABLED).  This file is built against the UNMODIFIED reference headers where they
// lie under /root/reference (oracle/Makefile, target `adapter`, output oracle/_ref/libvcadapter.so; nothing is copied)
// and defines a class of that name with the reference's four-method batch interface (src/cuda/cudabatch.hpp:39-59)
// whose generateConsensus() goes through the C ABI of libvechat_hip.so -- i.e. exactly the adapter INTEGRATION.md
// describes, reading sequences_/qualities_/positions_ of real racon::Window objects and writing consensus_.
// vcadapter_run() builds the same windows twice with the reference's createWindow / add_layer, runs one set through
// Window::generate_consensus on the CPU and the other through the batch class on the GPU, and compares.
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CUDA_ENABLED 1            // only effect in window.hpp: the friend declaration
#include "window.hpp"
#include "spoa/spoa.hpp"
#include "vechat_hip.h"

namespace racon {

class CUDABatchProcessor {
public:
    CUDABatchProcessor(uint32_t device, int8_t m, int8_t x, int8_t g, bool haplotype, bool trim, double d, double s, uint32_t k) {
        vc_params p{};
        p.device = (int32_t)device; p.match = m; p.mismatch = x; p.gap = g;
        p.sw_match = 3; p.sw_mismatch = -5; p.sw_gap = -4;                          // window.cpp:326
        p.min_confidence = d; p.min_support = s; p.num_prune = k;
        p.mode = haplotype ? 0 : 1; p.trim = trim ? 1 : 0; p.window_type = 1;
        if (vc_create(&ctx_, &p) != VC_OK) throw std::runtime_error(vc_last_error(nullptr));
    }
    ~CUDABatchProcessor() { vc_destroy(ctx_); }
    bool addWindow(std::shared_ptr<Window> w) { windows_.push_back(w); return true; }
    bool hasWindows() const { return !windows_.empty(); }
    void reset() { windows_.clear(); }

    const std::vector<bool>& generateConsensus() {
        std::vector<uint32_t> wso{0}, sb, se, rank, begins;
        std::vector<uint64_t> so{0};
        std::vector<uint8_t> hq, bases, quals, fasta;
        for (auto& w : windows_) {
            const uint32_t n = (uint32_t)w->sequences_.size();
            begins.resize(n);
            for (uint32_t i = 0; i < n; ++i) begins[i] = w->positions_[i].first;
            rank.resize(n);
            vc_rank_layers(begins.data(), n, rank.data());
            fasta.push_back((uint8_t)vc_backbone_is_fasta(w->qualities_[0].first, w->sequences_[0].second));
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t i = rank[k];
                const auto& s = w->sequences_[i];
                const auto& q = w->qualities_[i];
                bases.insert(bases.end(), s.first, s.first + s.second);
                if (q.first) quals.insert(quals.end(), q.first, q.first + s.second);
                else quals.insert(quals.end(), s.second, (uint8_t)'!');
                so.push_back(bases.size());
                sb.push_back(w->positions_[i].first); se.push_back(w->positions_[i].second);
                hq.push_back(q.first != nullptr);
            }
            wso.push_back((uint32_t)sb.size());
        }
        vc_batch b{(uint32_t)windows_.size(), wso.data(), so.data(), sb.data(), se.data(), hq.data(), bases.data(), quals.data(), fasta.data()};
        check(vc_submit(ctx_, &b)); check(vc_run(ctx_)); check(vc_sync(ctx_));
        uint64_t bytes = 0;
        check(vc_result_size(ctx_, &bytes));
        std::vector<uint64_t> off(windows_.size() + 1);
        std::vector<uint8_t> cons(bytes + 1), st(windows_.size());
        vc_result r{off.data(), cons.data(), cons.size(), st.data()};
        check(vc_collect(ctx_, &r));
        status_.assign(windows_.size(), false);
        for (size_t i = 0; i < windows_.size(); ++i) {
            if (st[i] > VC_WIN_UNPOLISHED) throw std::runtime_error("window outside the device envelope");
            windows_[i]->consensus_.assign((const char*)cons.data() + off[i], off[i + 1] - off[i]);
            status_[i] = st[i] == VC_WIN_OK;
        }
        return status_;
    }

private:
    void check(int rc) { if (rc != VC_OK) throw std::runtime_error(vc_last_error(ctx_)); }
    vc_ctx* ctx_ = nullptr;
    std::vector<std::shared_ptr<Window>> windows_;
    std::vector<bool> status_;
};

}  // namespace racon

extern "C" {

// Windows in add_layer() order, flattened: win_layer_off[nw+1] indexes the per-layer arrays; backbones separately.
// mode 0 = haplotype overload, 1 = racon-linear.  Returns the number of windows whose consensus or flag differ between
// the reference CPU path and the GPU batch class, or -1 on an exception (message in err).
int vcadapter_run(uint32_t nw, const char* const* bb, const uint32_t* bb_len, const char* const* bq,
                  const uint32_t* win_layer_off, const char* const* seqs, const uint32_t* lens, const char* const* quals,
                  const uint32_t* begins, const uint32_t* ends, int mode, int trim, int m, int n, int g,
                  double min_conf, double min_supp, uint32_t num_prune, char* err, uint32_t err_cap) {
    try {
        auto make = [&](uint32_t w) {
            auto win = racon::createWindow(w, 0, racon::WindowType::kTGS, bb[w], bb_len[w], bq[w], bb_len[w]);
            for (uint32_t i = win_layer_off[w]; i < win_layer_off[w + 1]; ++i)
                win->add_layer(seqs[i], lens[i], quals[i], quals[i] ? lens[i] : 0, begins[i], ends[i]);
            return win;
        };
        std::vector<std::shared_ptr<racon::Window>> cpu, gpu;
        for (uint32_t w = 0; w < nw; ++w) { cpu.push_back(make(w)); gpu.push_back(make(w)); }
        std::vector<bool> cpu_flag(nw);
        for (uint32_t w = 0; w < nw; ++w) {
            std::shared_ptr<spoa::AlignmentEngine> engine(spoa::AlignmentEngine::Create(spoa::AlignmentType::kNW, m, n, g));
            engine->Prealloc(bb_len[w], 5);
            cpu_flag[w] = mode == 0 ? cpu[w]->generate_consensus(engine, trim != 0, true, min_conf, min_supp, num_prune)
                                    : cpu[w]->generate_consensus(engine, trim != 0);
        }
        racon::CUDABatchProcessor proc(0, (int8_t)m, (int8_t)n, (int8_t)g, mode == 0, trim != 0, min_conf, min_supp, num_prune);
        for (auto& w : gpu) proc.addWindow(w);
        const std::vector<bool>& gpu_flag = proc.generateConsensus();
        int bad = 0;
        for (uint32_t w = 0; w < nw; ++w)
            if (cpu[w]->consensus() != gpu[w]->consensus() || cpu_flag[w] != gpu_flag[w]) ++bad;
        proc.reset();
        return bad;
    } catch (const std::exception& e) {
        if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
        return -1;
    }
}

}  // extern "C"
