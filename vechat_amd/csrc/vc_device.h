// Device-side data layout of libvechat_hip.so (gfx950).  Everything is SoA over "slots": the
// windows of the chunk that is resident in HBM.  Ids are 16-bit (graph capacity < 65535 nodes/edges).
//
// Graph (mirrors spoa::Graph, vendor/spoa/include/spoa/graph.hpp:46-100, as index arrays):
//   code[v]                     the base itself (the reference's coder/decoder is a bijection on bytes,
//                               graph.cpp:198-205, and codes are only ever compared for equality)
//   in-list / out-list of v     singly linked through the edges, append order == edge id order
//                               (Graph::AddEdge appends to edges_, tail->outedges, head->inedges at once,
//                               graph.cpp:94-107), so "list order" is reproduced by following next links
//   al[v][0..al_cnt)            Node::aligned_nodes in append order; ma entries per node, ma = max(4, distinct bytes in the
//                               batch - 1): the members of an aligned group carry distinct bytes (graph.cpp:258-277)
//   e_tn[e] = tail | next_in<<16,  e_hn[e] = head | next_out<<16,  e_w[e] = Edge::weight
#pragma once
#include <stdint.h>

#define VC_NONE16   0xFFFFu
#define VC_INLINE_PRED 6       // predecessors stored inline in a row record
#ifndef VC_BAND_LANES
#define VC_BAND_LANES 16       // banded matrix store: lanes of a DP row that are written (around the rank diagonal); see vc_band_start
#endif
#define VC_MAXTIE   16         // NW end-cell ties remembered for the exact-rank resolver

// row record flags
#define VC_RF_SINK  1u
#define VC_RF_OVF   4u
#define VC_RF_PREV  8u     // one of the predecessors is the row directly above (still in registers)
#define VC_RF_SLOW  16u    // frec only: a listed predecessor is the virtual row 0 or lies beyond the LDS ring, or the list overflowed
#define VC_RF_KEEP  64u    // frec only (kept-row ring): a later row reads this row back from the LDS ring; bits 27..29 of the word hold its slot
#define VC_RF_FULL  128u   // frec only (banded matrix store): a later row reads this row back from the stored matrix: store it whole
#define VC_RF_PLAIN 32u    // frec only: the row directly above is the ONLY predecessor (the commonest row: nothing to fetch)

struct VcGraph {
    uint32_t* n_nodes;    // [CW]
    uint32_t* n_edges;    // [CW]
    uint8_t*  code;       // [CW*NC]
    uint16_t* in_first;   // [CW*NC]
    uint16_t* in_last;
    uint16_t* out_first;
    uint16_t* out_last;
    uint8_t*  al_cnt;     // [CW*NC]
    uint16_t* al;         // [CW*NC*ma]
    uint32_t  ma;         // entries per aligned list (even)
    uint32_t* e_tn;       // [CW*EC]
    uint32_t* e_hn;       // [CW*EC]
    uint32_t* e_w;        // [CW*EC]
    uint16_t* ord;        // [CW*NC] a valid DP order of the nodes (aligned groups contiguous), kept incrementally
    uint16_t* pos;        // [CW*NC] inverse of ord
    uint16_t* visits;     // [CW*NC] sequences (len >= 2) whose path contains the node == Node::Coverage() (graph.cpp:38-56)
    uint4*    nrec;       // [CW*NC] in-side of a node in one 16-byte record: x = code | in-degree << 16, then the tails of the first
                          //   VC_INLINE_PRED in-edges as u16 NODE ids in list order.  Kept by k_init / k_addaln (build phase only);
                          //   the row records of a full-span layer are made from it with three dependent loads per row
};

// Input of the alignment kernel for one graph, produced by k_topo in rank order.
// rec[r] (16 B): byte0 code, byte1 flags, byte2 npred, byte3 unused, then 6 x u16 predecessor
// row distances (delta = row - pred_row; the virtual row 0 is at delta == row).  With VC_RF_OVF the
// first two u16 hold a u32 offset into ovf[] where all npred deltas live.
struct VcDp {
    uint32_t* nrows;      // [CW]
    uint32_t* flags;      // [CW] bit0: outside the kernel envelope; bit1: rows follow VcGraph::ord, not the reference's rank
    uint4*    rec;        // [CW*NC]
    uint4*    frec;       // [CW*NC] the forward kernel's view of the same row (see vc_make_frec)
    uint8_t*  fie;        // [CW*(NC+4)] by ROW NUMBER (1-based; entry 0 = the virtual row = 0): distance to the row of the first in-edge when it is
                          //   1..15 and the list is inline, else 0 -- what the backtrack's speculation table holds (4 bits per row in LDS).  Written
                          //   beside `rec` by the row builders, so that a backtrack wave loads 1 byte per row instead of the 16-byte record
    uint16_t* rank2node;  // [CW*NC]
    uint16_t* ovf;        // [CW*EC]
};

struct VcBatchDev {
    uint32_t n_windows;
    const uint32_t* win_seq_off;
    const uint64_t* seq_off;
    const uint32_t* seq_begin;
    const uint32_t* seq_end;
    const uint8_t*  seq_has_qual;
    const uint8_t*  bases;
    const uint8_t*  quals;
    const uint8_t*  win_fasta;
    double*   win_avg;     // [n_windows] average_weight (window.cpp:301-309)
    uint8_t*  status;      // [n_windows]
    uint32_t* errinfo;     // [n_windows] (site << 16) | detail of the first non-OK status, for diagnostics
    uint8_t*  cons;        // [n_windows*cons_cap]
    uint32_t* cons_len;    // [n_windows]
    uint32_t  cons_cap;
    const uint32_t* lut_w; // [256]
    const double*   lut_d; // [256]
};
