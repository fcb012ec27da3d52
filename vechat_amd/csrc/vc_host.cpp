// Host-side helpers of libvechat_hip.so that keep reference semantics on the host side of the
// C-ABI boundary (no device code in this file; it is also built into libvechat_host.so so the
// CPU-only tests can exercise it).
//
//   vc_rank_layers        <- src/window.cpp:203-210   (unstable std::sort on positions_.first)
//   vc_backbone_is_fasta  <- src/window.cpp:223       (C-string vs std::string comparison quirk)
//   vc_weight_lut         <- vendor/spoa/src/graph.cpp:165-170, src/window.cpp:366
//   vc_synth_*            <- SURVEY 8(d) synthetic-window generator (our own; BASELINE configs)
#include "vechat_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

extern "C" {

void vc_rank_layers(const uint32_t* begins, uint32_t n_seqs, uint32_t* rank_out) {
    std::vector<uint32_t> rank;
    rank.reserve(n_seqs);
    for (uint32_t i = 0; i < n_seqs; ++i) rank.emplace_back(i);
    if (n_seqs > 1) {
        // identical call shape to the reference so libstdc++'s introsort yields the same
        // permutation among equal keys
        std::sort(rank.begin() + 1, rank.end(),
                  [&](uint32_t lhs, uint32_t rhs) { return begins[lhs] < begins[rhs]; });
    }
    for (uint32_t i = 0; i < n_seqs; ++i) rank_out[i] = rank[i];
}

int vc_backbone_is_fasta(const char* quality_cstr, uint32_t backbone_len) {
    return quality_cstr == std::string(backbone_len, '!') ? 1 : 0;
}

void vc_weight_lut(uint32_t lut[256]) {
    for (int c = 0; c < 256; ++c) {
        int q = static_cast<int>(static_cast<signed char>(c));
        double w = (1 - pow(10, (33 - q) / 10.)) * 1000;
        lut[c] = (w >= 0 && w < 4294967296.0) ? static_cast<uint32_t>(w)
                                               : static_cast<uint32_t>(static_cast<int64_t>(w));
    }
}

}  // extern "C"

// ------------------------------------------------------------------------------- synthetic data
struct vc_synth {
    std::vector<uint32_t> win_seq_off;
    std::vector<uint64_t> seq_off;
    std::vector<uint32_t> seq_begin, seq_end;
    std::vector<uint8_t>  seq_has_qual, bases, quals, win_fasta;
    std::vector<uint32_t> seq_orig;   // add_layer() index of each stored (rank-ordered) sequence
};

namespace {

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    double u() { return static_cast<double>(g() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return static_cast<uint32_t>(g() % n); }
};

const char kBases[4] = {'A', 'C', 'G', 'T'};

void mutate(Rng& r, const std::string& src, const vc_synth_cfg& c, std::string* dst) {
    dst->clear();
    const double ps = c.error_rate * c.frac_sub;
    const double pd = ps + c.error_rate * c.frac_del;
    const double pi = pd + c.error_rate * c.frac_ins;
    for (char b : src) {
        double u = r.u();
        if (u < ps) {
            char x;
            do { x = kBases[r.below(4)]; } while (x == b);
            dst->push_back(x);
        } else if (u < pd) {
            // deletion
        } else if (u < pi) {
            dst->push_back(kBases[r.below(4)]);
            dst->push_back(b);
        } else {
            dst->push_back(b);
        }
    }
    if (dst->empty()) dst->push_back(src.empty() ? 'A' : src[0]);
}

struct WinBuf {
    std::vector<std::string> seq, qual;
    std::vector<uint32_t> begin, end;
    std::vector<uint8_t> has_qual;
    uint8_t fasta;
};

void gen_window(const vc_synth_cfg& c, uint64_t index, WinBuf* w) {
    Rng r(c.seed * 0x9E3779B97F4A7C15ull + index * 0xD1B54A32D192ED03ull + 0x5851F42D4C957F2Dull);
    const uint32_t L = c.backbone_len;
    std::string hap[2];
    hap[0].resize(L);
    for (uint32_t i = 0; i < L; ++i) hap[0][i] = kBases[r.below(4)];
    hap[1] = hap[0];
    if (c.n_haplotypes > 1) {
        for (uint32_t i = 0; i < L; ++i) {
            if (r.u() < c.snp_rate) {
                char x;
                do { x = kBases[r.below(4)]; } while (x == hap[0][i]);
                hap[1][i] = x;
            }
        }
    }
    auto rand_qual = [&](size_t n) {
        std::string q(n, '!');
        for (size_t i = 0; i < n; ++i) q[i] = static_cast<char>('&' + r.below(20));   // Phred 5..24
        return q;
    };
    w->seq.clear(); w->qual.clear(); w->begin.clear(); w->end.clear(); w->has_qual.clear();

    std::string bb;
    mutate(r, hap[0], c, &bb);
    if (bb.size() > L) bb.resize(L);
    while (bb.size() < L) bb.push_back(kBases[r.below(4)]);
    w->seq.push_back(bb);
    w->qual.push_back(c.backbone_fastq ? rand_qual(L) : std::string(L, '!'));
    w->begin.push_back(0); w->end.push_back(0); w->has_qual.push_back(1);
    w->fasta = c.backbone_fastq ? 0 : 1;

    std::string s;
    for (uint32_t d = 0; d < c.n_layers; ++d) {
        const std::string& src = hap[c.n_haplotypes > 1 ? (r.below(2)) : 0];
        uint32_t b = 0, e = L - 1;
        if (c.frac_partial > 0 && r.u() < c.frac_partial && L >= 16) {
            uint32_t span = L / 4 + r.below(L / 2 + 1);
            b = r.below(L - span + 1);
            e = b + span - 1;
            if (e >= L) e = L - 1;
        }
        mutate(r, src.substr(b, e - b + 1), c, &s);
        w->seq.push_back(s);
        w->qual.push_back(c.fastq ? rand_qual(s.size()) : std::string(s.size(), '!'));
        w->begin.push_back(b); w->end.push_back(e); w->has_qual.push_back(c.fastq ? 1 : 0);
    }
}

}  // namespace

extern "C" {

vc_synth* vc_synth_generate(const vc_synth_cfg* cfg, uint64_t first, uint32_t n, uint32_t n_threads) {
    if (!cfg || cfg->backbone_len == 0) return nullptr;
    if (n_threads == 0) n_threads = 1;
    if (n_threads > n) n_threads = n ? n : 1;
    struct Part { std::vector<WinBuf> wins; };
    std::vector<Part> parts(n_threads);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            uint64_t lo = static_cast<uint64_t>(n) * t / n_threads, hi = static_cast<uint64_t>(n) * (t + 1) / n_threads;
            parts[t].wins.resize(hi - lo);
            for (uint64_t i = lo; i < hi; ++i) gen_window(*cfg, first + i, &parts[t].wins[i - lo]);
        });
    }
    for (auto& x : th) x.join();

    vc_synth* s = new vc_synth();
    s->win_seq_off.push_back(0);
    s->seq_off.push_back(0);
    std::vector<uint32_t> rank;
    for (auto& p : parts) {
        for (auto& w : p.wins) {
            uint32_t ns = static_cast<uint32_t>(w.seq.size());
            rank.resize(ns);
            vc_rank_layers(w.begin.data(), ns, rank.data());
            for (uint32_t k = 0; k < ns; ++k) {
                uint32_t i = rank[k];
                s->bases.insert(s->bases.end(), w.seq[i].begin(), w.seq[i].end());
                s->quals.insert(s->quals.end(), w.qual[i].begin(), w.qual[i].end());
                s->seq_off.push_back(s->bases.size());
                s->seq_begin.push_back(w.begin[i]);
                s->seq_end.push_back(w.end[i]);
                s->seq_has_qual.push_back(w.has_qual[i]);
                s->seq_orig.push_back(i);
            }
            s->win_seq_off.push_back(static_cast<uint32_t>(s->seq_begin.size()));
            s->win_fasta.push_back(w.fasta);
        }
        std::vector<WinBuf>().swap(p.wins);
    }
    return s;
}

void vc_synth_batch(const vc_synth* s, vc_batch* out) {
    out->n_windows = static_cast<uint32_t>(s->win_fasta.size());
    out->win_seq_off = s->win_seq_off.data();
    out->seq_off = s->seq_off.data();
    out->seq_begin = s->seq_begin.data();
    out->seq_end = s->seq_end.data();
    out->seq_has_qual = s->seq_has_qual.data();
    out->bases = s->bases.data();
    out->quals = s->quals.data();
    out->win_fasta = s->win_fasta.data();
}

uint64_t vc_synth_n_seqs(const vc_synth* s) { return s->seq_begin.size(); }
const uint32_t* vc_synth_orig_index(const vc_synth* s) { return s->seq_orig.data(); }
uint64_t vc_synth_n_bytes(const vc_synth* s) { return s->bases.size(); }
void vc_synth_free(vc_synth* s) { delete s; }

}  // extern "C"
