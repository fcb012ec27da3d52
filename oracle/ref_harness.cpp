// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Thin extern "C" harness around the *unmodified* reference sources, compiled where they lie
// under /root/reference (see oracle/Makefile, target `ref`).  Output goes to oracle/_ref/ only.
// Nothing from the reference is copied into this repository: this file merely calls the
// reference's public API
//   racon::createWindow / Window::add_layer / Window::generate_consensus   (src/window.hpp:27-55)
//   spoa::AlignmentEngine::Create / Align, spoa::Graph::AddAlignment / GenerateConsensus
//                                   (vendor/spoa/include/spoa/{alignment_engine,graph}.hpp)
// so that tests can (a) pin oracle/vc_oracle.c against the real implementation and
// (b) generate the golden fixtures under tests/golden/ (script: tests/golden/make_golden.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <stdexcept>

#include "window.hpp"
#include "spoa/spoa.hpp"

extern "C" {

// One window through the reference.  Layers are given in add_layer() order (NOT rank order:
// the reference sorts internally, src/window.cpp:203-210).
//   bb/bq         : backbone bases / quality; bq must be a NUL-terminated buffer because the
//                   reference compares it as a C string (src/window.cpp:223)
//   quals[i]==NULL: FASTA layer
//   mode          : 0 = haplotype overload (window.cpp:176), 1 = racon-linear overload (window.cpp:74)
// returns 0 ok, -1 exception, -2 output too small.  *polished receives the bool result.
int vcref_window(const char* bb, uint32_t bb_len, const char* bq,
                 uint32_t n_layers, const char* const* seqs, const uint32_t* lens,
                 const char* const* quals, const uint32_t* begins, const uint32_t* ends,
                 int mode, int window_type, int trim,
                 int m, int n, int g, double min_conf, double min_supp, uint32_t num_prune,
                 char* out, uint32_t out_cap, uint32_t* out_len, int* polished) {
    try {
        auto win = racon::createWindow(0, 0,
            window_type ? racon::WindowType::kTGS : racon::WindowType::kNGS,
            bb, bb_len, bq, bb_len);
        for (uint32_t i = 0; i < n_layers; ++i) {
            win->add_layer(seqs[i], lens[i], quals[i], quals[i] ? lens[i] : 0, begins[i], ends[i]);
        }
        std::shared_ptr<spoa::AlignmentEngine> engine(
            spoa::AlignmentEngine::Create(spoa::AlignmentType::kNW, m, n, g));
        engine->Prealloc(bb_len, 5);   // as src/polisher.cpp:186-190
        bool ok;
        if (mode == 0) {
            ok = win->generate_consensus(engine, trim != 0, true, min_conf, min_supp, num_prune);
        } else {
            ok = win->generate_consensus(engine, trim != 0);
        }
        const std::string& c = win->consensus();
        *out_len = c.size();
        *polished = ok ? 1 : 0;
        if (c.size() > out_cap) return -2;
        memcpy(out, c.data(), c.size());
        return 0;
    } catch (std::exception&) {
        return -1;
    }
}

// spoa known-answer flow (vendor/spoa/test/spoa_test.cpp:38-52): align every sequence to the
// growing graph, add it (with or without qualities), return GenerateConsensus().
//   type: 0 = kSW, 1 = kNW, 2 = kOV
int vcref_spoa_consensus(uint32_t n_seqs, const char* const* seqs, const uint32_t* lens,
                         const char* const* quals, int type, int m, int n, int g,
                         char* out, uint32_t out_cap, uint32_t* out_len) {
    try {
        auto ae = spoa::AlignmentEngine::Create(static_cast<spoa::AlignmentType>(type), m, n, g);
        spoa::Graph gr{};
        for (uint32_t i = 0; i < n_seqs; ++i) {
            auto a = ae->Align(seqs[i], lens[i], gr);
            if (quals && quals[i]) {
                gr.AddAlignment(a, seqs[i], lens[i], quals[i], lens[i]);
            } else {
                gr.AddAlignment(a, seqs[i], lens[i]);
            }
        }
        std::string c = gr.GenerateConsensus();
        *out_len = c.size();
        if (c.size() > out_cap) return -2;
        memcpy(out, c.data(), c.size());
        return 0;
    } catch (std::exception&) {
        return -1;
    }
}

// Intermediate probe: one Align() call of `query` against the graph built from `seqs`
// (spoa KAT flow above, NW/SW of the given scores).  Emits the alignment pair list and the
// graph's rank_to_node order so the restatement's internals can be compared, not just its output.
int vcref_spoa_align_probe(uint32_t n_seqs, const char* const* seqs, const uint32_t* lens,
                           const char* const* quals, int build_type, int m, int n, int g,
                           const char* query, uint32_t query_len, int query_type,
                           int32_t* pairs, uint32_t pairs_cap, uint32_t* n_pairs,
                           uint32_t* rank_to_node, uint32_t rank_cap, uint32_t* n_nodes) {
    try {
        auto ae = spoa::AlignmentEngine::Create(static_cast<spoa::AlignmentType>(build_type), m, n, g);
        spoa::Graph gr{};
        for (uint32_t i = 0; i < n_seqs; ++i) {
            auto a = ae->Align(seqs[i], lens[i], gr);
            if (quals && quals[i]) gr.AddAlignment(a, seqs[i], lens[i], quals[i], lens[i]);
            else gr.AddAlignment(a, seqs[i], lens[i]);
        }
        auto qe = spoa::AlignmentEngine::Create(static_cast<spoa::AlignmentType>(query_type), m, n, g);
        auto a = qe->Align(query, query_len, gr);
        *n_pairs = a.size();
        *n_nodes = gr.nodes().size();
        if (a.size() > pairs_cap || gr.nodes().size() > rank_cap) return -2;
        for (size_t i = 0; i < a.size(); ++i) { pairs[2 * i] = a[i].first; pairs[2 * i + 1] = a[i].second; }
        for (size_t i = 0; i < gr.rank_to_node().size(); ++i) rank_to_node[i] = gr.rank_to_node()[i]->id;
        return 0;
    } catch (std::exception&) {
        return -1;
    }
}

}  // extern "C"

// Per-stage digests of one window's haplotype-overload run, for localising a divergence (tests/golden/stages.json).
// Window::generate_consensus is one function with private state, so the stages are replayed here step by step with the
// reference's own public Graph / AlignmentEngine calls in the order of src/window.cpp:176-428 (layers in rank order by
// the same std::sort); the caller checks that the replay's consensus equals vcref_window's, i.e. that it IS the same run.
// A record is 8 x u64: kind, index, nodes, edges, hash(nodes: byte, aligned ids), hash(edges: tail, head, weight),
// alignment pairs, hash(pairs).  kind 1: graph after AddAlignment of layer `index` (rank order, with that layer's
// alignment); 2: graph after prune + LargestSubgraph number `index`; 3: after the AddWeights round `index`; 4: the final
// local alignment of the backbone.
namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void u8(uint8_t b) { h ^= b; h *= 1099511628211ull; }
    void u32(uint32_t v) { for (int i = 0; i < 4; ++i) u8((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; ++i) u8((uint8_t)(v >> (8 * i))); }
};
void stage_record(std::vector<uint64_t>& out, uint64_t kind, uint64_t index, const spoa::Graph* g, const spoa::Alignment* a) {
    Fnv hn, he, hp;
    uint64_t nn = 0, ne = 0, np = 0;
    if (g) {
        nn = g->nodes().size(); ne = g->edges().size();
        for (const auto& n : g->nodes()) {
            hn.u8(g->decoder(n->code));
            hn.u32((uint32_t)n->aligned_nodes.size());
            for (const auto* m : n->aligned_nodes) hn.u32(m->id);
        }
        for (const auto& e : g->edges()) { he.u32(e->tail->id); he.u32(e->head->id); he.u64((uint64_t)e->weight); }
    }
    if (a) { np = a->size(); for (const auto& p : *a) { hp.u32((uint32_t)p.first); hp.u32((uint32_t)p.second); } }
    const uint64_t rec[8] = {kind, index, nn, ne, g ? hn.h : 0, g ? he.h : 0, np, a ? hp.h : 0};
    out.insert(out.end(), rec, rec + 8);
}
}  // namespace

extern "C" int vcref_window_stages(const char* bb, uint32_t bb_len, const char* bq,
                                   uint32_t n_layers, const char* const* seqs, const uint32_t* lens,
                                   const char* const* quals, const uint32_t* begins, const uint32_t* ends,
                                   int m, int n, int g, double min_conf, double min_supp, uint32_t num_prune,
                                   uint64_t* rec_out, uint32_t rec_cap, uint32_t* n_rec,
                                   char* out, uint32_t out_cap, uint32_t* out_len) {
    try {
        std::vector<uint64_t> recs;
        // what Window holds after createWindow + add_layer (src/window.cpp:27-72)
        std::vector<std::pair<const char*, uint32_t>> S{{bb, bb_len}}, Q{{bq, bb_len}};
        std::vector<std::pair<uint32_t, uint32_t>> P{{0, 0}};
        for (uint32_t i = 0; i < n_layers; ++i) {
            if (lens[i] == 0 || begins[i] == ends[i]) continue;
            S.emplace_back(seqs[i], lens[i]); Q.emplace_back(quals[i], quals[i] ? lens[i] : 0); P.emplace_back(begins[i], ends[i]);
        }
        *n_rec = 0; *out_len = 0;
        if (S.size() < 3) return 1;
        auto nw = spoa::AlignmentEngine::Create(spoa::AlignmentType::kNW, m, n, g);
        nw->Prealloc(bb_len, 5);
        spoa::Graph graph{};
        graph.AddAlignment(spoa::Alignment(), S[0].first, S[0].second, Q[0].first, Q[0].second);
        std::vector<uint32_t> rank(S.size());
        for (uint32_t i = 0; i < S.size(); ++i) rank[i] = i;
        std::sort(rank.begin() + 1, rank.end(), [&](uint32_t l, uint32_t r) { return P[l].first < P[r].first; });
        const uint32_t offset = 0.01 * S[0].second;
        double total = 0.0;
        const std::uint16_t window_len = S[0].second;
        bool if_fasta = false;
        if (Q[0].first == std::string(Q[0].second, '!')) { total += S[0].second; if_fasta = true; }
        else for (std::uint16_t q = 0; q < Q[0].second; ++q) total += 1 - pow(10, (33 - Q[0].first[q]) / 10.0);
        for (uint32_t j = 1; j < S.size(); ++j) {
            const uint32_t i = rank[j];
            spoa::Alignment al;
            if (P[i].first < offset && P[i].second > S[0].second - offset) al = nw->Align(S[i].first, S[i].second, graph);
            else {
                std::vector<const spoa::Graph::Node*> mapping;
                auto sub = graph.Subgraph(P[i].first, P[i].second, &mapping);
                al = nw->Align(S[i].first, S[i].second, sub);
                sub.UpdateAlignment(mapping, &al);
            }
            if (Q[i].first == nullptr) { graph.AddAlignment(al, S[i].first, S[i].second); total += S[i].second; }
            else {
                graph.AddAlignment(al, S[i].first, S[i].second, Q[i].first, Q[i].second);
                for (std::uint16_t q = 0; q < Q[i].second; ++q) total += (1 - pow(10, (33 - Q[i].first[q]) / 10.0));
            }
            stage_record(recs, 1, j, &graph, &al);
        }
        const double avg = if_fasta ? 2.0 * total / window_len : 2.0 * total / window_len * 1000;
        graph.PruneGraph(0, min_conf, min_supp, avg);
        std::unique_ptr<spoa::Graph> ptr(new spoa::Graph(graph.LargestSubgraph()));
        graph.Clear();
        stage_record(recs, 2, 0, ptr.get(), nullptr);
        auto sw = spoa::AlignmentEngine::Create(spoa::AlignmentType::kSW, 3, -5, -4);
        for (uint32_t k = 0; k + 1 < num_prune; ++k) {
            for (uint32_t j = 0; j < S.size(); ++j) {
                const uint32_t i = rank[j];
                spoa::Alignment al;
                if (j == 0 || (P[i].first < offset && P[i].second > S[0].second - offset)) al = nw->Align(S[i].first, S[i].second, *ptr);
                else al = sw->Align(S[i].first, S[i].second, *ptr);
                std::vector<std::uint32_t> w;
                if (Q[i].first == nullptr) w.assign(S[i].second, 1);
                else for (uint32_t x = 0; x < S[i].second; ++x) w.emplace_back((std::uint32_t)((1 - pow(10, (33 - Q[i].first[x]) / 10.0)) * 1000));
                ptr->AddWeights(al, S[i].first, S[i].second, w);
            }
            stage_record(recs, 3, k, ptr.get(), nullptr);
            ptr->PruneGraph(0, min_conf, min_supp, avg);
            ptr.reset(new spoa::Graph(ptr->LargestSubgraph()));
            stage_record(recs, 2, k + 1, ptr.get(), nullptr);
        }
        auto fin = sw->Align(S[0].first, S[0].second, *ptr);
        stage_record(recs, 4, 0, ptr.get(), &fin);
        const std::string c = ptr->GenerateCorrectedSequence(fin);
        *n_rec = recs.size() / 8;
        *out_len = c.size();
        if (recs.size() > rec_cap || c.size() > out_cap) return -2;
        memcpy(rec_out, recs.data(), recs.size() * 8);
        memcpy(out, c.data(), c.size());
        return 0;
    } catch (std::exception&) {
        return -1;
    }
}
