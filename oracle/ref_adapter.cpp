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
        prm_ = p;
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
        failed_.assign(windows_.size(), false);
        for (size_t i = 0; i < windows_.size(); ++i) {
            failed_[i] = st[i] > VC_WIN_UNPOLISHED;              // overflow / invalid: no bytes, the caller's CPU path takes the window
            if (failed_[i]) continue;
            windows_[i]->consensus_.assign((const char*)cons.data() + off[i], off[i + 1] - off[i]);
            status_[i] = st[i] == VC_WIN_OK;
        }
        return status_;
    }
    const std::vector<bool>& failed() const { return failed_; }
    void limit_graph(uint32_t max_nodes, uint32_t max_edges) {            // test knob: capacities small enough that windows overflow
        if (!max_nodes && !max_edges) return;
        vc_destroy(ctx_); ctx_ = nullptr;
        prm_.max_nodes = max_nodes; prm_.max_edges = max_edges;
        if (vc_create(&ctx_, &prm_) != VC_OK) throw std::runtime_error(vc_last_error(nullptr));
    }

private:
    void check(int rc) { if (rc != VC_OK) throw std::runtime_error(vc_last_error(ctx_)); }
    vc_ctx* ctx_ = nullptr;
    std::vector<std::shared_ptr<Window>> windows_;
    std::vector<bool> status_, failed_;
    vc_params prm_{};
};

// The loop of CUDAPolisher::polish (src/cuda/cudapolisher.cpp:217-414) over that batch class, as the `HipPolisher::polish()`
// of INTEGRATION.md would run it: batches are filled from windows_ in order (:260-283), generateConsensus() gives the
// per-window flags (:300-321), windows the device did not take are polished by the reference's own CPU path (:355-379), and
// the sequences are stitched with their LN/RC/XC tags (:386-409 == src/polisher.cpp:520-547).  racon::Polisher itself cannot
// be compiled here (polisher.hpp needs the un-vendored thread_pool), so this class holds what the loop reads from it --
// windows_, the target names and coverages, the polisher type -- and runs single-threaded; `dst` receives (name + tags, data)
// where the reference calls createSequence(name + tags, data).
class HipPolisherLoop {
public:
    std::vector<std::shared_ptr<Window>> windows_;
    std::vector<std::string> target_names_;
    std::vector<uint32_t> targets_coverages_;
    bool fragment_correction_ = true, trim_ = true, haplotype_ = true;
    int8_t match_ = 3, mismatch_ = -5, gap_ = -4;
    double min_confidence_ = 0.2, min_support_ = 0.2;
    uint32_t num_prune_ = 3;
    uint32_t window_length_ = 500;
    uint32_t cpu_windows_ = 0;                                   // windows that went through the CPU path

    bool cpu_window(uint64_t j) {
        std::shared_ptr<spoa::AlignmentEngine> engine(spoa::AlignmentEngine::Create(spoa::AlignmentType::kNW, match_, mismatch_, gap_));
        engine->Prealloc(window_length_, 5);                                   // src/polisher.cpp:186-190
        ++cpu_windows_;
        return haplotype_ ? windows_[j]->generate_consensus(engine, trim_, true, min_confidence_, min_support_, num_prune_)
                          : windows_[j]->generate_consensus(engine, trim_);
    }

    void polish(std::vector<std::pair<std::string, std::string>>& dst, bool drop_unpolished_sequences, uint32_t batch_windows,
                bool cpu_only, uint32_t max_nodes, uint32_t max_edges) {
        std::vector<bool> window_consensus_status_(windows_.size(), false), on_cpu(windows_.size(), cpu_only);
        if (!cpu_only) {
            CUDABatchProcessor batch(0, match_, mismatch_, gap_, haplotype_, trim_, min_confidence_, min_support_, num_prune_);
            batch.limit_graph(max_nodes, max_edges);
            uint64_t next_window_index = 0;
            while (next_window_index < windows_.size()) {
                batch.reset();
                const uint64_t first = next_window_index;
                while (next_window_index < windows_.size() && next_window_index - first < batch_windows)
                    batch.addWindow(windows_[next_window_index++]);
                const std::vector<bool>& flags = batch.generateConsensus();
                for (uint64_t i = first; i < next_window_index; ++i) {
                    window_consensus_status_[i] = flags[i - first];
                    on_cpu[i] = batch.failed()[i - first];
                }
            }
        }
        for (uint64_t i = 0; i < windows_.size(); ++i)
            if (on_cpu[i]) window_consensus_status_[i] = cpu_window(i);
        std::string polished_data;
        uint32_t num_polished_windows = 0;
        for (uint64_t i = 0; i < windows_.size(); ++i) {
            num_polished_windows += window_consensus_status_[i] ? 1 : 0;
            polished_data += windows_[i]->consensus();
            if (i == windows_.size() - 1 || windows_[i + 1]->rank() == 0) {
                const double polished_ratio = num_polished_windows / static_cast<double>(windows_[i]->rank() + 1);
                if (!drop_unpolished_sequences || polished_ratio > 0) {
                    std::string tags = fragment_correction_ ? "r" : "";
                    tags += " LN:i:" + std::to_string(polished_data.size());
                    tags += " RC:i:" + std::to_string(targets_coverages_[windows_[i]->id()]);
                    tags += " XC:f:" + std::to_string(polished_ratio);
                    dst.emplace_back(target_names_[windows_[i]->id()] + tags, polished_data);
                }
                num_polished_windows = 0;
                polished_data.clear();
            }
            windows_[i].reset();
        }
    }
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


// The polish() loop above over windows of several targets: win_id[w] / win_rank[w] as Polisher::initialize numbers them
// (src/polisher.cpp:389-462: id = target index, rank = window index inside the target).  Writes ">name tags\ndata\n" per kept
// target into out; returns the text length, -1 on an exception, -2 when out is too small; *cpu_windows = windows that took
// the CPU path (all of them with cpu_only, else the ones the device reported as overflowed / invalid).
long vcadapter_polish(uint32_t nw, const uint32_t* win_id, const uint32_t* win_rank, const char* const* bb, const uint32_t* bb_len,
                      const char* const* bq, const uint32_t* win_layer_off, const char* const* seqs, const uint32_t* lens,
                      const char* const* quals, const uint32_t* begins, const uint32_t* ends,
                      uint32_t n_targets, const char* const* target_names, const uint32_t* target_cov,
                      int haplotype, int trim, int fragment, int drop_unpolished, uint32_t batch_windows, int cpu_only,
                      uint32_t max_nodes, uint32_t max_edges, char* out, uint64_t out_cap, uint32_t* cpu_windows, char* err, uint32_t err_cap) {
    try {
        racon::HipPolisherLoop pl;
        pl.haplotype_ = haplotype != 0; pl.trim_ = trim != 0; pl.fragment_correction_ = fragment != 0;
        for (uint32_t t = 0; t < n_targets; ++t) { pl.target_names_.emplace_back(target_names[t]); pl.targets_coverages_.push_back(target_cov[t]); }
        for (uint32_t w = 0; w < nw; ++w) {
            auto win = racon::createWindow(win_id[w], win_rank[w], racon::WindowType::kTGS, bb[w], bb_len[w], bq[w], bb_len[w]);
            for (uint32_t i = win_layer_off[w]; i < win_layer_off[w + 1]; ++i)
                win->add_layer(seqs[i], lens[i], quals[i], quals[i] ? lens[i] : 0, begins[i], ends[i]);
            pl.windows_.push_back(std::move(win));
        }
        std::vector<std::pair<std::string, std::string>> dst;
        pl.polish(dst, drop_unpolished != 0, batch_windows ? batch_windows : nw, cpu_only != 0, max_nodes, max_edges);
        std::string text;
        for (auto& d : dst) { text += ">"; text += d.first; text += "\n"; text += d.second; text += "\n"; }
        if (cpu_windows) *cpu_windows = pl.cpu_windows_;
        if (text.size() > out_cap) return -2;
        std::memcpy(out, text.data(), text.size());
        return (long)text.size();
    } catch (const std::exception& e) {
        if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
        return -1;
    }
}

}  // extern "C"
