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
