/* TEST INFRASTRUCTURE -- CPU restatement ("oracle") of VeChat's per-window hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (vechat_amd/, libvechat_hip.so) never links, imports or executes it.
 *
 * Parity status: PINNED.  The restatement is checked
 *   (1) against the four linear-gap known-answer tests the reference's own suite holds
 *       (vendor/spoa/test/spoa_test.cpp:150-164,198-212,246-260,294-308) and
 *   (2) against the real reference compiled in place (oracle/_ref, see Makefile / ref_harness.cpp)
 *       on randomized windows and on the committed fixtures under tests/golden/.
 */
#ifndef VC_ORACLE_H_
#define VC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Packed window batch.  The same layout is consumed by libvechat_hip.so (include/vechat_hip.h).
 * Sequence 0 of every window is the backbone; sequences 1.. are the layers ALREADY in the
 * reference's `rank` order (src/window.cpp:203-210; the std::sort stays on the host). */
typedef struct vco_batch {
    uint32_t        n_windows;
    const uint32_t* win_seq_off;   /* [n_windows+1] first sequence of each window            */
    const uint64_t* seq_off;       /* [n_seqs+1]    byte offset of each sequence in bases[]   */
    const uint32_t* seq_begin;     /* [n_seqs]      window.cpp positions_.first  (backbone 0) */
    const uint32_t* seq_end;       /* [n_seqs]      window.cpp positions_.second (backbone 0) */
    const uint8_t*  seq_has_qual;  /* [n_seqs]      0 = FASTA layer (qualities_[i].first==nullptr) */
    const uint8_t*  bases;         /* concatenated sequences                                   */
    const uint8_t*  quals;         /* same offsets; ignored where seq_has_qual==0              */
    const uint8_t*  win_fasta;     /* [n_windows]   value of window.cpp:223's `if_fasta` test  */
} vco_batch;

typedef struct vco_params {
    int32_t  match, mismatch, gap;          /* Polisher engine scores (main.cpp:46-61: 3,-5,-4) */
    int32_t  sw_match, sw_mismatch, sw_gap; /* hard-coded 3,-5,-4 at window.cpp:326             */
    double   min_confidence, min_support;   /* -d / -s                                          */
    uint32_t num_prune;                     /* -k                                               */
    int32_t  mode;                          /* 0 = haplotype overload, 1 = racon-linear overload */
    int32_t  trim;                          /* linear overload only                             */
    int32_t  window_type;                   /* 0 = kNGS, 1 = kTGS                               */
} vco_params;

typedef struct vco_stats {
    uint64_t cells;        /* sum over every Align call of nodes_in_graph * sequence_len */
    uint64_t alignments;   /* number of Align calls                                               */
    uint64_t max_nodes;    /* largest build graph seen (nodes / edges), for capacity planning     */
    uint64_t max_edges;
} vco_stats;

/* Runs windows [w0, w1) of the batch.  cons_off[n_windows+1] / cons (capacity cons_cap bytes)
 * receive the consensus strings of the processed range starting at cons_off[w0] (caller sets
 * cons_off[w0]); polished[w] receives generate_consensus()'s bool.  Returns 0, or -2 when cons_cap
 * is too small, or -1 on an input the reference would throw on. */
int vco_run(const vco_batch* b, const vco_params* p, uint32_t w0, uint32_t w1,
            uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint8_t* polished,
            vco_stats* stats);

/* spoa known-answer flow (spoa_test.cpp:38-52): align each sequence, add it, GenerateConsensus. */
int vco_spoa_consensus(uint32_t n_seqs, const uint8_t* const* seqs, const uint32_t* lens,
                       const uint8_t* const* quals, int type /*0 SW, 1 NW*/, int m, int n, int g,
                       uint8_t* out, uint32_t out_cap, uint32_t* out_len);

/* Intermediate probe mirroring ref_harness.cpp:vcref_spoa_align_probe. */
int vco_spoa_align_probe(uint32_t n_seqs, const uint8_t* const* seqs, const uint32_t* lens,
                         const uint8_t* const* quals, int build_type, int m, int n, int g,
                         const uint8_t* query, uint32_t query_len, int query_type,
                         int32_t* pairs, uint32_t pairs_cap, uint32_t* n_pairs,
                         uint32_t* rank_to_node, uint32_t rank_cap, uint32_t* n_nodes);

/* Per-stage digests of one window's haplotype-overload run (graph after every layer, after every prune and AddWeights
 * round, the final alignment) in the record format of oracle/ref_harness.cpp:vcref_window_stages: 8 x u64 each. */
int vco_window_stages(const vco_batch* b, const vco_params* p, uint32_t w, uint64_t* rec, uint32_t rec_cap, uint32_t* n_rec);

/* Quality -> weight table, graph.cpp:165-170 / window.cpp:366: uint32((1-10^((33-q)/10))*1000). */
void vco_weight_lut(uint32_t lut[256]);

#ifdef __cplusplus
}
#endif
#endif
