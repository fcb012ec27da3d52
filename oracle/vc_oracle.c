/* TEST INFRASTRUCTURE -- plain-C restatement ("oracle") of VeChat's per-window hot path.
 * See vc_oracle.h for the usage rule and the parity status (PINNED).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The data structures are our own (index-based arrays, no pointers between nodes/edges):
 *   - an edge keeps its id for life; PruneGraph's nullptr tombstones (graph.cpp:940-981) are
 *     modelled by alive[e]==0, so "edges_ order" == alive edges in id order and the per-node
 *     in/out lists still hold the dead ids at their original positions;
 *   - node ids are creation order, exactly like nodes_.size() in Graph::AddNode (graph.cpp:88-92).
 */
#include "vc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NONE 0xFFFFFFFFu

/* ------------------------------------------------------------------ small vectors */
typedef struct { uint32_t* d; uint32_t n, cap; } u32vec;

static void v_push(u32vec* v, uint32_t x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 4;
        v->d = (uint32_t*)realloc(v->d, (size_t)v->cap * sizeof(uint32_t));
    }
    v->d[v->n++] = x;
}
static void v_free(u32vec* v) { free(v->d); v->d = NULL; v->n = v->cap = 0; }

typedef struct { int32_t* d; uint32_t n, cap; } alnvec;   /* pairs (node, pos) flattened */
static void a_push(alnvec* v, int32_t node, int32_t pos) {
    if (v->n + 2 > v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->d = (int32_t*)realloc(v->d, (size_t)v->cap * sizeof(int32_t));
    }
    v->d[v->n++] = node; v->d[v->n++] = pos;
}

/* ------------------------------------------------------------------ graph */
typedef struct {
    uint32_t n_nodes, cap_nodes;
    uint8_t* code;
    u32vec*  in;        /* edge ids, append order (Node::inedges)   */
    u32vec*  out;       /* edge ids, append order (Node::outedges)  */
    u32vec*  aligned;   /* node ids, append order (Node::aligned_nodes) */
    uint32_t n_edges, cap_edges;
    uint32_t* tail; uint32_t* head; int64_t* weight; uint8_t* alive; u32vec* labels;
    u32vec   rank;      /* rank_to_node_ */
    int32_t  coder[256]; int32_t decoder[256]; uint32_t num_codes;
    uint32_t nseq;      /* sequences_.size() */
    u32vec   consensus; /* consensus_ (node ids) */
} graph;

static void g_init(graph* g) {
    memset(g, 0, sizeof(*g));
    for (int i = 0; i < 256; ++i) { g->coder[i] = -1; g->decoder[i] = -1; }
}
static void g_free(graph* g) {
    for (uint32_t i = 0; i < g->n_nodes; ++i) { v_free(&g->in[i]); v_free(&g->out[i]); v_free(&g->aligned[i]); }
    for (uint32_t i = 0; i < g->n_edges; ++i) v_free(&g->labels[i]);
    free(g->code); free(g->in); free(g->out); free(g->aligned);
    free(g->tail); free(g->head); free(g->weight); free(g->alive); free(g->labels);
    v_free(&g->rank); v_free(&g->consensus);
    memset(g, 0, sizeof(*g));
}

/* ------------------------------------------------------------------ stage digests (vco_window_stages)
 * Same records as oracle/ref_harness.cpp:vcref_window_stages writes for the real reference: 8 x u64 per stage
 * (kind, index, nodes, edges, hash(nodes), hash(edges), pairs, hash(pairs)), FNV-1a over little-endian fields. */
typedef struct { uint64_t* d; uint32_t n, cap; int overflow; } stagevec;
static __thread stagevec* g_stages = NULL;
static uint64_t fnv_u8(uint64_t h, uint8_t b) { h ^= b; return h * 1099511628211ull; }
static uint64_t fnv_u32(uint64_t h, uint32_t v) { for (int i = 0; i < 4; ++i) h = fnv_u8(h, (uint8_t)(v >> (8 * i))); return h; }
static uint64_t fnv_u64(uint64_t h, uint64_t v) { for (int i = 0; i < 8; ++i) h = fnv_u8(h, (uint8_t)(v >> (8 * i))); return h; }

/* Graph::AddNode, graph.cpp:88-92 */
static uint32_t g_add_node(graph* g, uint32_t code) {
    if (g->n_nodes == g->cap_nodes) {
        uint32_t nc = g->cap_nodes ? g->cap_nodes * 2 : 1024;
        g->code = (uint8_t*)realloc(g->code, nc);
        g->in = (u32vec*)realloc(g->in, nc * sizeof(u32vec));
        g->out = (u32vec*)realloc(g->out, nc * sizeof(u32vec));
        g->aligned = (u32vec*)realloc(g->aligned, nc * sizeof(u32vec));
        g->cap_nodes = nc;
    }
    uint32_t id = g->n_nodes++;
    g->code[id] = (uint8_t)code;
    memset(&g->in[id], 0, sizeof(u32vec));
    memset(&g->out[id], 0, sizeof(u32vec));
    memset(&g->aligned[id], 0, sizeof(u32vec));
    return id;
}

static uint32_t g_new_edge(graph* g, uint32_t tail, uint32_t head, uint32_t label, uint32_t w) {
    if (g->n_edges == g->cap_edges) {
        uint32_t nc = g->cap_edges ? g->cap_edges * 2 : 2048;
        g->tail = (uint32_t*)realloc(g->tail, nc * sizeof(uint32_t));
        g->head = (uint32_t*)realloc(g->head, nc * sizeof(uint32_t));
        g->weight = (int64_t*)realloc(g->weight, nc * sizeof(int64_t));
        g->alive = (uint8_t*)realloc(g->alive, nc);
        g->labels = (u32vec*)realloc(g->labels, nc * sizeof(u32vec));
        g->cap_edges = nc;
    }
    uint32_t e = g->n_edges++;
    g->tail[e] = tail; g->head[e] = head; g->weight[e] = (int64_t)w; g->alive[e] = 1;
    memset(&g->labels[e], 0, sizeof(u32vec));
    v_push(&g->labels[e], label);
    v_push(&g->out[tail], e);
    v_push(&g->in[head], e);
    return e;
}

/* Graph::AddEdge, graph.cpp:94-107 (find by head in tail's out-list, else append) +
 * Edge::AddSequence, graph.cpp:70-74 */
static void g_add_edge(graph* g, uint32_t tail, uint32_t head, uint32_t w) {
    const u32vec* o = &g->out[tail];
    for (uint32_t k = 0; k < o->n; ++k) {
        uint32_t e = o->d[k];
        if (g->head[e] == head) {
            v_push(&g->labels[e], g->nseq);
            g->weight[e] += (int64_t)w;
            return;
        }
    }
    g_new_edge(g, tail, head, g->nseq, w);
}

/* Graph::AddSequence, graph.cpp:109-130: fresh chain for seq[begin,end); returns first node or NONE */
static uint32_t g_add_chain(graph* g, const uint8_t* seq, const uint32_t* W, uint32_t begin, uint32_t end) {
    if (begin == end) return NONE;
    uint32_t prev = NONE, first = NONE;
    for (uint32_t i = begin; i < end; ++i) {
        uint32_t curr = g_add_node(g, (uint32_t)g->coder[seq[i]]);
        if (first == NONE) first = curr;
        if (prev != NONE) g_add_edge(g, prev, curr, W[i - 1] + W[i]);
        prev = curr;
    }
    return first;
}

/* Graph::TopologicalSort, graph.cpp:301-371 (iterative DFS, ids in order, in-edge tails then
 * aligned nodes pushed; a node is emitted followed by its aligned nodes) */
static void g_toposort(graph* g) {
    g->rank.n = 0;
    uint32_t N = g->n_nodes;
    uint8_t* marks = (uint8_t*)calloc(N ? N : 1, 1);
    uint8_t* ignored = (uint8_t*)calloc(N ? N : 1, 1);
    u32vec st = {0, 0, 0};
    for (uint32_t s = 0; s < N; ++s) {
        if (marks[s] != 0) continue;
        v_push(&st, s);
        while (st.n) {
            uint32_t c = st.d[st.n - 1];
            int valid = 1;
            if (marks[c] != 2) {
                for (uint32_t k = 0; k < g->in[c].n; ++k) {
                    uint32_t t = g->tail[g->in[c].d[k]];
                    if (marks[t] != 2) { v_push(&st, t); valid = 0; }
                }
                if (!ignored[c]) {
                    for (uint32_t k = 0; k < g->aligned[c].n; ++k) {
                        uint32_t a = g->aligned[c].d[k];
                        if (marks[a] != 2) { v_push(&st, a); ignored[a] = 1; valid = 0; }
                    }
                }
                if (valid) {
                    marks[c] = 2;
                    if (!ignored[c]) {
                        v_push(&g->rank, c);
                        for (uint32_t k = 0; k < g->aligned[c].n; ++k) v_push(&g->rank, g->aligned[c].d[k]);
                    }
                } else {
                    marks[c] = 1;
                }
            }
            if (valid) st.n--;
        }
    }
    free(marks); free(ignored); v_free(&st);
}

/* Graph::AddAlignment(alignment, sequence, len, weights), graph.cpp:182-299 */
static int g_add_alignment(graph* g, const alnvec* A, const uint8_t* seq, uint32_t len, const uint32_t* W) {
    if (len == 0) return 0;
    for (uint32_t i = 0; i < len; ++i) {                    /* graph.cpp:198-205 first-seen codes */
        if (g->coder[seq[i]] == -1) {
            g->coder[seq[i]] = (int32_t)g->num_codes;
            g->decoder[g->num_codes++] = seq[i];
        }
    }
    if (A == NULL || A->n == 0) {                           /* graph.cpp:207-212 */
        g_add_chain(g, seq, W, 0, len);
        g->nseq++;
        g_toposort(g);
        return 0;
    }
    uint32_t np = A->n / 2;
    int32_t vfront = -1, vback = -1;
    for (uint32_t k = 0; k < np; ++k) {                     /* graph.cpp:214-232 */
        int32_t q = A->d[2 * k + 1];
        if (q != -1) {
            if (q < 0 || q >= (int32_t)len) return -1;
            if (vfront == -1) vfront = q;
            vback = q;
        }
    }
    if (vfront == -1) return -1;

    uint32_t begin = g_add_chain(g, seq, W, 0, (uint32_t)vfront);      /* :234 */
    uint32_t prev = (begin != NONE) ? g->n_nodes - 1 : NONE;           /* :235 */
    uint32_t last = g_add_chain(g, seq, W, (uint32_t)vback + 1, len);  /* :236 */

    for (uint32_t k = 0; k < np; ++k) {                                /* :239-291 */
        int32_t n = A->d[2 * k], q = A->d[2 * k + 1];
        if (q == -1) continue;
        uint32_t c = (uint32_t)g->coder[seq[q]];
        uint32_t curr = NONE;
        if (n == -1) {
            curr = g_add_node(g, c);
        } else {
            uint32_t j = (uint32_t)n;
            if (g->code[j] == c) {
                curr = j;
            } else {
                for (uint32_t t = 0; t < g->aligned[j].n; ++t) {
                    uint32_t a = g->aligned[j].d[t];
                    if (g->code[a] == c) { curr = a; break; }
                }
                if (curr == NONE) {
                    curr = g_add_node(g, c);
                    uint32_t na = g->aligned[j].n;          /* snapshot: list grows below */
                    for (uint32_t t = 0; t < na; ++t) {
                        uint32_t a = g->aligned[j].d[t];
                        v_push(&g->aligned[a], curr);
                        v_push(&g->aligned[curr], a);
                    }
                    v_push(&g->aligned[j], curr);
                    v_push(&g->aligned[curr], j);
                }
            }
        }
        if (begin == NONE) begin = curr;
        if (prev != NONE) g_add_edge(g, prev, curr, W[q - 1] + W[q]);
        prev = curr;
    }
    if (last != NONE) g_add_edge(g, prev, last, W[vback] + W[vback + 1]);   /* :292-295 */
    g->nseq++;                                                           /* :296 */
    g_toposort(g);
    return 0;
}

/* ------------------------------------------------------------------ alignment (linear gap) */
#define KNEG (INT32_MIN + 1024)   /* sisd_alignment_engine.cpp:13-14 */

/* AlignmentEngine::WorstCaseAlignmentScore, alignment_engine.cpp:101-110 with e=g, q=g, c=g */
static int64_t worst_case(int64_t m, int64_t gp, int64_t i, int64_t j) {
    int64_t d = i > j ? i - j : j - i, mn = i < j ? i : j;
    int64_t gs_d = d == 0 ? 0 : gp + (d - 1) * gp;
    int64_t gs_i = i == 0 ? 0 : gp + (i - 1) * gp;
    int64_t gs_j = j == 0 ? 0 : gp + (j - 1) * gp;
    int64_t a = -1 * (m * mn + gs_d), b = gs_i + gs_j;
    return a < b ? a : b;
}

/* SisdAlignmentEngine::Align + Initialize + Linear, sisd_alignment_engine.cpp:256-460,118-254.
 * type: 0 = kSW, 1 = kNW.  Returns -1 where the reference throws (possible overflow). */
static int g_align(const graph* g, int type, int m, int n, int gp,
                   const uint8_t* seq, uint32_t len, alnvec* out, vco_stats* stats) {
    out->n = 0;
    uint32_t N = g->n_nodes;
    if (N == 0 || len == 0) return 0;                                   /* :265-267 */
    /* the production (SIMD) engine checks len+8 (simd impl:699-706) and throws below int32 range */
    if (worst_case(m, gp, (int64_t)len + 8, N) < (int64_t)KNEG) return -1;
    if (stats) { stats->cells += (uint64_t)N * len; stats->alignments++; }

    size_t w = (size_t)len + 1;
    int32_t* H = (int32_t*)malloc(((size_t)N + 1) * w * sizeof(int32_t));
    uint32_t* node_rank = (uint32_t*)malloc((size_t)N * sizeof(uint32_t));
    for (uint32_t r = 0; r < N; ++r) node_rank[g->rank.d[r]] = r;

    /* Initialize, :118-254 */
    H[0] = 0;
    if (type == 0) {
        for (size_t j = 1; j < w; ++j) H[j] = 0;
        for (size_t i = 1; i <= N; ++i) H[i * w] = 0;
    } else {
        for (size_t j = 1; j < w; ++j) H[j] = (int32_t)j * gp;
        for (uint32_t i = 1; i <= N; ++i) {
            const u32vec* in = &g->in[g->rank.d[i - 1]];
            int32_t pen = in->n == 0 ? 0 : KNEG;
            for (uint32_t k = 0; k < in->n; ++k) {
                size_t pi = node_rank[g->tail[in->d[k]]] + 1;
                if (H[pi * w] > pen) pen = H[pi * w];
            }
            H[(size_t)i * w] = pen + gp;
        }
    }

    int32_t max_score = type == 0 ? 0 : KNEG;
    uint32_t max_i = 0, max_j = 0;
    for (uint32_t r = 0; r < N; ++r) {                                  /* Linear, :315-360 */
        uint32_t v = g->rank.d[r];
        size_t i = (size_t)r + 1;
        const u32vec* in = &g->in[v];
        char c = (char)g->decoder[g->code[v]];
        size_t pi = in->n == 0 ? 0 : node_rank[g->tail[in->d[0]]] + 1;
        int32_t* Hr = H + i * w;
        const int32_t* Hp = H + pi * w;
        for (size_t j = 1; j < w; ++j) {
            int32_t s = ((char)seq[j - 1] == c) ? m : n;
            int32_t a = Hp[j - 1] + s, b = Hp[j] + gp;
            Hr[j] = a > b ? a : b;
        }
        for (uint32_t k = 1; k < in->n; ++k) {
            pi = node_rank[g->tail[in->d[k]]] + 1;
            Hp = H + pi * w;
            for (size_t j = 1; j < w; ++j) {
                int32_t s = ((char)seq[j - 1] == c) ? m : n;
                int32_t a = Hp[j - 1] + s, b = Hp[j] + gp;
                int32_t x = Hr[j] > b ? Hr[j] : b;
                Hr[j] = a > x ? a : x;
            }
        }
        int sink = g->out[v].n == 0;
        for (size_t j = 1; j < w; ++j) {
            int32_t a = Hr[j - 1] + gp;
            if (a > Hr[j]) Hr[j] = a;
            if (type == 0) {
                if (Hr[j] < 0) Hr[j] = 0;
                if (max_score < Hr[j]) { max_score = Hr[j]; max_i = (uint32_t)i; max_j = (uint32_t)j; }
            } else if (sink && j == w - 1) {
                if (max_score < Hr[j]) { max_score = Hr[j]; max_i = (uint32_t)i; max_j = (uint32_t)j; }
            }
        }
    }

    if (!(max_i == 0 && max_j == 0)) {                                  /* :362-459 */
        uint32_t i = max_i, j = max_j, pi_ = 0, pj_ = 0;
        for (;;) {
            if (type == 0) { if (H[(size_t)i * w + j] == 0) break; }
            else           { if (i == 0 && j == 0) break; }
            int32_t Hij = H[(size_t)i * w + j];
            int found = 0;
            const u32vec* in = i ? &g->in[g->rank.d[i - 1]] : NULL;
            if (i != 0 && j != 0) {
                uint32_t v = g->rank.d[i - 1];
                int32_t s = ((char)seq[j - 1] == (char)g->decoder[g->code[v]]) ? m : n;
                uint32_t np = in->n ? in->n : 1;
                for (uint32_t k = 0; k < np; ++k) {
                    uint32_t p = in->n ? node_rank[g->tail[in->d[k]]] + 1 : 0;
                    if (Hij == H[(size_t)p * w + (j - 1)] + s) { pi_ = p; pj_ = j - 1; found = 1; break; }
                }
            }
            if (!found && i != 0) {
                uint32_t np = in->n ? in->n : 1;
                for (uint32_t k = 0; k < np; ++k) {
                    uint32_t p = in->n ? node_rank[g->tail[in->d[k]]] + 1 : 0;
                    if (Hij == H[(size_t)p * w + j] + gp) { pi_ = p; pj_ = j; found = 1; break; }
                }
            }
            if (!found && j != 0 && Hij == H[(size_t)i * w + j - 1] + gp) { pi_ = i; pj_ = j - 1; found = 1; }
            if (!found) { free(H); free(node_rank); return -1; }   /* cannot happen on a DAG */
            a_push(out, i == pi_ ? -1 : (int32_t)g->rank.d[i - 1], j == pj_ ? -1 : (int32_t)j - 1);
            i = pi_; j = pj_;
        }
        /* std::reverse, :458 */
        uint32_t npairs = out->n / 2;
        for (uint32_t a = 0; a < npairs / 2; ++a) {
            uint32_t b = npairs - 1 - a;
            int32_t t0 = out->d[2 * a], t1 = out->d[2 * a + 1];
            out->d[2 * a] = out->d[2 * b]; out->d[2 * a + 1] = out->d[2 * b + 1];
            out->d[2 * b] = t0; out->d[2 * b + 1] = t1;
        }
    }
    free(H); free(node_rank);
    return 0;
}

/* ------------------------------------------------------------------ subgraph (partial-span layers) */
/* Graph::ExtractSubgraph + Graph::Subgraph, graph.cpp:640-732.  map[new id] = old id. */
static void g_subgraph(const graph* g, uint32_t begin, uint32_t end, graph* sub, u32vec* map) {
    uint32_t N = g->n_nodes;
    uint8_t* in_sub = (uint8_t*)calloc(N, 1);
    u32vec st = {0, 0, 0};
    v_push(&st, end);                                     /* ExtractSubgraph(nodes_[end], nodes_[begin]) */
    while (st.n) {
        uint32_t c = st.d[--st.n];
        if (!in_sub[c] && c >= begin) {
            for (uint32_t k = 0; k < g->in[c].n; ++k) v_push(&st, g->tail[g->in[c].d[k]]);
            for (uint32_t k = 0; k < g->aligned[c].n; ++k) v_push(&st, g->aligned[c].d[k]);
            in_sub[c] = 1;
        }
    }
    g_init(sub);
    sub->num_codes = g->num_codes;
    memcpy(sub->coder, g->coder, sizeof(g->coder));
    memcpy(sub->decoder, g->decoder, sizeof(g->decoder));
    uint32_t* g2s = (uint32_t*)malloc((size_t)N * sizeof(uint32_t));
    map->n = 0;
    for (uint32_t v = 0; v < N; ++v) {
        g2s[v] = NONE;
        if (!in_sub[v]) continue;
        g2s[v] = g_add_node(sub, g->code[v]);
        v_push(map, v);
    }
    for (uint32_t v = 0; v < N; ++v) {
        if (!in_sub[v]) continue;
        uint32_t jt = g2s[v];
        for (uint32_t k = 0; k < g->in[v].n; ++k) {
            uint32_t e = g->in[v].d[k];
            if (g2s[g->tail[e]] != NONE) g_add_edge(sub, g2s[g->tail[e]], jt, (uint32_t)g->weight[e]);
        }
        for (uint32_t k = 0; k < g->aligned[v].n; ++k) {
            uint32_t a = g->aligned[v].d[k];
            if (g2s[a] != NONE) v_push(&sub->aligned[jt], g2s[a]);
        }
    }
    g_toposort(sub);
    free(in_sub); free(g2s); v_free(&st);
}

/* ------------------------------------------------------------------ VeChat additions */
/* Graph::PruneGraph, graph.cpp:811-982 (min_weight==0 never triggers, window.cpp:311) */
static void g_prune(graph* g, int64_t min_weight, double d, double s, double avg) {
    uint32_t E = g->n_edges;
    uint8_t* prune = (uint8_t*)calloc(E ? E : 1, 1);
    for (uint32_t e = 0; e < E; ++e) {
        if (!g->alive[e]) continue;
        if (g->weight[e] < min_weight) { prune[e] = 1; continue; }
        int64_t tot = 0;
        const u32vec* o = &g->out[g->tail[e]];
        for (uint32_t k = 0; k < o->n; ++k) tot += g->weight[o->d[k]];
        double conf_uv = (double)g->weight[e] / (double)tot;
        double support = (double)g->weight[e] / avg;
        tot = 0;
        const u32vec* in = &g->in[g->head[e]];
        for (uint32_t k = 0; k < in->n; ++k) tot += g->weight[in->d[k]];
        double conf_vu = (double)g->weight[e] / (double)tot;
        if (conf_uv >= d && conf_vu >= d && support >= s) prune[e] = 0; else prune[e] = 1;
    }
    for (uint32_t e = 0; e < E; ++e) if (prune[e]) g->alive[e] = 0;   /* tombstones, :940-981 */
    free(prune);
}

/* Graph::DfsUtil (recursive preorder), graph.cpp:984-1019 -- explicit frames, neighbours =
 * live in-edge tails then live out-edge heads, `visited` tested when the loop reaches u */
static void g_dfs_component(const graph* g, uint32_t v0, uint8_t* visited, u32vec* comp) {
    typedef struct { uint32_t v, k; } frame;
    uint32_t cap = 64, sp = 0;
    frame* fr = (frame*)malloc(cap * sizeof(frame));
    visited[v0] = 1; v_push(comp, v0);
    fr[sp].v = v0; fr[sp].k = 0; sp++;
    while (sp) {
        frame* f = &fr[sp - 1];
        uint32_t v = f->v, nin = g->in[v].n, nout = g->out[v].n;
        uint32_t u = NONE;
        while (f->k < nin + nout) {
            uint32_t k = f->k++;
            uint32_t e = k < nin ? g->in[v].d[k] : g->out[v].d[k - nin];
            if (!g->alive[e]) continue;
            uint32_t cand = k < nin ? g->tail[e] : g->head[e];
            if (!visited[cand]) { u = cand; break; }
        }
        if (u == NONE) { sp--; continue; }
        visited[u] = 1; v_push(comp, u);
        if (sp == cap) { cap *= 2; fr = (frame*)realloc(fr, cap * sizeof(frame)); }
        fr[sp].v = u; fr[sp].k = 0; sp++;
    }
    free(fr);
}

/* Graph::LargestSubgraph, graph.cpp:1021-1089 (+ AddNodeForSubgraph/AddEdgeForSubgraph :1091-1102) */
static void g_largest_subgraph(const graph* g, graph* sub) {
    uint32_t N = g->n_nodes;
    uint8_t* visited = (uint8_t*)calloc(N ? N : 1, 1);
    u32vec comp = {0, 0, 0}, best = {0, 0, 0};
    uint32_t best_size = 0;
    for (uint32_t v = 0; v < N; ++v) {
        if (visited[v]) continue;
        comp.n = 0;
        g_dfs_component(g, v, visited, &comp);
        if (comp.n >= best_size) {                      /* `>=`: later component wins ties, :1049 */
            best_size = comp.n;
            best.n = 0;
            for (uint32_t k = 0; k < comp.n; ++k) v_push(&best, comp.d[k]);
        }
    }
    g_init(sub);
    sub->num_codes = g->num_codes;
    memcpy(sub->coder, g->coder, sizeof(g->coder));
    memcpy(sub->decoder, g->decoder, sizeof(g->decoder));
    uint32_t* v2s = (uint32_t*)malloc((size_t)(N ? N : 1) * sizeof(uint32_t));
    for (uint32_t k = 0; k < best.n; ++k) v2s[best.d[k]] = g_add_node(sub, g->code[best.d[k]]);
    for (uint32_t k = 0; k < best.n; ++k) {
        uint32_t v = best.d[k];
        for (uint32_t t = 0; t < g->out[v].n; ++t) {
            uint32_t e = g->out[v].d[t];
            if (!g->alive[e]) continue;
            g_new_edge(sub, v2s[v], v2s[g->head[e]], 0, 0);   /* weight 0, label 0, no dedup */
        }
    }
    g_toposort(sub);
    free(visited); free(v2s); v_free(&comp); v_free(&best);
}

/* Graph::AddWeights, graph.cpp:1104-1165.  Returns 1 when the alignment was empty (skipped). */
static int g_add_weights(graph* g, const alnvec* A, uint32_t len, const uint32_t* W) {
    if (len == 0) return 0;
    if (A->n == 0) return 1;
    uint32_t prev = NONE;
    for (uint32_t k = 0; k < A->n / 2; ++k) {
        int32_t n = A->d[2 * k], q = A->d[2 * k + 1];
        if (n == -1 || q == -1) { prev = NONE; continue; }
        uint32_t curr = (uint32_t)n;
        if (prev != NONE) g_add_edge(g, prev, curr, W[q - 1] + W[q]);
        prev = curr;
    }
    return 0;
}

/* ------------------------------------------------------------------ racon consensus (round 2) */
/* Graph::BranchCompletion, graph.cpp:590-638 */
static uint32_t g_branch_completion(const graph* g, uint32_t rank, int64_t* scores, uint32_t* pred) {
    uint32_t start = g->rank.d[rank];
    for (uint32_t k = 0; k < g->out[start].n; ++k) {
        uint32_t h = g->head[g->out[start].d[k]];
        for (uint32_t t = 0; t < g->in[h].n; ++t) {
            uint32_t tl = g->tail[g->in[h].d[t]];
            if (tl != start) scores[tl] = -1;
        }
    }
    uint32_t max = NONE;
    for (uint32_t i = rank + 1; i < g->rank.n; ++i) {
        uint32_t it = g->rank.d[i];
        scores[it] = -1; pred[it] = NONE;
        for (uint32_t t = 0; t < g->in[it].n; ++t) {
            uint32_t e = g->in[it].d[t], tl = g->tail[e];
            if (scores[tl] == -1) continue;
            if (scores[it] < g->weight[e] ||
                (scores[it] == g->weight[e] && scores[pred[it]] <= scores[tl])) {
                scores[it] = g->weight[e]; pred[it] = tl;
            }
        }
        if (pred[it] != NONE) scores[it] += scores[pred[it]];
        if (max == NONE || scores[max] < scores[it]) max = it;
    }
    return max;
}

/* Graph::TraverseHeaviestBundle, graph.cpp:534-588 */
static void g_heaviest_bundle(graph* g) {
    g->consensus.n = 0;
    if (g->rank.n == 0) return;
    uint32_t N = g->n_nodes;
    uint32_t* pred = (uint32_t*)malloc((size_t)N * sizeof(uint32_t));
    int64_t* scores = (int64_t*)malloc((size_t)N * sizeof(int64_t));
    for (uint32_t i = 0; i < N; ++i) { pred[i] = NONE; scores[i] = -1; }
    uint32_t max = NONE;
    for (uint32_t r = 0; r < g->rank.n; ++r) {
        uint32_t it = g->rank.d[r];
        for (uint32_t t = 0; t < g->in[it].n; ++t) {
            uint32_t e = g->in[it].d[t], tl = g->tail[e];
            if (scores[it] < g->weight[e] ||
                (scores[it] == g->weight[e] && scores[pred[it]] <= scores[tl])) {
                scores[it] = g->weight[e]; pred[it] = tl;
            }
        }
        if (pred[it] != NONE) scores[it] += scores[pred[it]];
        if (max == NONE || scores[max] < scores[it]) max = it;
    }
    if (g->out[max].n != 0) {
        uint32_t* n2r = (uint32_t*)malloc((size_t)N * sizeof(uint32_t));
        for (uint32_t r = 0; r < g->rank.n; ++r) n2r[g->rank.d[r]] = r;
        while (g->out[max].n != 0) max = g_branch_completion(g, n2r[max], scores, pred);
        free(n2r);
    }
    while (pred[max] != NONE) { v_push(&g->consensus, max); max = pred[max]; }
    v_push(&g->consensus, max);
    for (uint32_t a = 0, b = g->consensus.n - 1; a < b; ++a, --b) {
        uint32_t t = g->consensus.d[a]; g->consensus.d[a] = g->consensus.d[b]; g->consensus.d[b] = t;
    }
    free(pred); free(scores);
}

/* Node::Coverage, graph.cpp:38-56: number of distinct labels on in- and out-edges */
static uint32_t g_coverage(const graph* g, uint32_t v, uint32_t* stamp, uint32_t tick) {
    uint32_t cnt = 0;
    for (int dir = 0; dir < 2; ++dir) {
        const u32vec* l = dir ? &g->out[v] : &g->in[v];
        for (uint32_t k = 0; k < l->n; ++k) {
            const u32vec* lb = &g->labels[l->d[k]];
            for (uint32_t t = 0; t < lb->n; ++t) {
                if (stamp[lb->d[t]] != tick) { stamp[lb->d[t]] = tick; cnt++; }
            }
        }
    }
    return cnt;
}

/* ------------------------------------------------------------------ windows */
static double g_qlut_d[256];
static uint32_t g_qlut_w[256];
static int g_lut_ready = 0;

static void lut_init(void) {
    if (g_lut_ready) return;
    for (int c = 0; c < 256; ++c) {
        int q = (int)(signed char)c;                      /* `char` is signed on the reference's target */
        double p = 1 - pow(10, (33 - q) / 10.0);          /* window.cpp:235,295 */
        g_qlut_d[c] = p;
        double w = (1 - pow(10, (33 - q) / 10.)) * 1000;  /* graph.cpp:169, window.cpp:366 */
        g_qlut_w[c] = (w >= 0 && w < 4294967296.0) ? (uint32_t)w : (uint32_t)(int64_t)w;
    }
    g_lut_ready = 1;
}
void vco_weight_lut(uint32_t lut[256]) { lut_init(); memcpy(lut, g_qlut_w, sizeof(g_qlut_w)); }

typedef struct {
    const uint8_t* seq; const uint8_t* qual; uint32_t len, begin, end; int has_qual;
} seqview;

static uint32_t* make_weights(const seqview* s, int use_qual) {
    uint32_t* W = (uint32_t*)malloc((size_t)(s->len ? s->len : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < s->len; ++i) W[i] = use_qual ? g_qlut_w[s->qual[i]] : 1u;
    return W;
}

static void stage_emit(uint64_t kind, uint64_t index, const graph* g, const alnvec* A) {
    if (!g_stages) return;
    if (g_stages->n + 8 > g_stages->cap) { g_stages->overflow = 1; return; }
    const uint64_t seed = 1469598103934665603ull;
    uint64_t hn = seed, he = seed, hp = seed, np = 0;
    if (g) {
        for (uint32_t v = 0; v < g->n_nodes; ++v) {
            hn = fnv_u8(hn, (uint8_t)g->decoder[g->code[v]]);
            hn = fnv_u32(hn, g->aligned[v].n);
            for (uint32_t k = 0; k < g->aligned[v].n; ++k) hn = fnv_u32(hn, g->aligned[v].d[k]);
        }
        for (uint32_t e = 0; e < g->n_edges; ++e) { he = fnv_u32(he, g->tail[e]); he = fnv_u32(he, g->head[e]); he = fnv_u64(he, (uint64_t)g->weight[e]); }
    }
    if (A) { np = A->n / 2; for (uint32_t k = 0; k < A->n; ++k) hp = fnv_u32(hp, (uint32_t)A->d[k]); }
    uint64_t* r = g_stages->d + g_stages->n;
    r[0] = kind; r[1] = index; r[2] = g ? g->n_nodes : 0; r[3] = g ? g->n_edges : 0; r[4] = g ? hn : 0; r[5] = g ? he : 0; r[6] = np; r[7] = A ? hp : 0;
    g_stages->n += 8;
}

/* the build loop shared by both overloads, window.cpp:100-136 / :239-298 */
static int build_graph(graph* G, const seqview* sv, uint32_t nseq, uint32_t L, const vco_params* p,
                       double* total, int fasta, vco_stats* stats) {
    alnvec A = {0, 0, 0};
    int rc = 0;
    uint32_t* W = make_weights(&sv[0], 1);               /* backbone always goes through the quality overload */
    rc = g_add_alignment(G, NULL, sv[0].seq, sv[0].len, W);
    free(W);
    if (total) {
        if (fasta) *total += (double)sv[0].len;                                    /* window.cpp:225 */
        else for (uint32_t q = 0; q < sv[0].len; ++q) *total += g_qlut_d[sv[0].qual[q]];  /* :232-236 */
    }
    uint32_t offset = (uint32_t)(0.01 * L);               /* window.cpp:212 */
    for (uint32_t j = 1; j < nseq && rc == 0; ++j) {
        const seqview* s = &sv[j];
        if (s->begin < offset && s->end > L - offset) {   /* :253-254 */
            rc = g_align(G, 1, p->match, p->mismatch, p->gap, s->seq, s->len, &A, stats);
        } else {
            graph sub; u32vec map = {0, 0, 0};
            g_subgraph(G, s->begin, s->end, &sub, &map);
            rc = g_align(&sub, 1, p->match, p->mismatch, p->gap, s->seq, s->len, &A, stats);
            for (uint32_t k = 0; k < A.n / 2; ++k)          /* UpdateAlignment, graph.cpp:734-745 */
                if (A.d[2 * k] != -1) A.d[2 * k] = (int32_t)map.d[A.d[2 * k]];
            g_free(&sub); v_free(&map);
        }
        if (rc) break;
        W = make_weights(s, s->has_qual);
        rc = g_add_alignment(G, &A, s->seq, s->len, W);
        free(W);
        if (rc == 0) stage_emit(1, j, G, &A);
        if (total) {
            if (!s->has_qual) *total += (double)s->len;                              /* :283 */
            else for (uint32_t q = 0; q < s->len; ++q) *total += g_qlut_d[s->qual[q]];  /* :292-296 */
        }
    }
    free(A.d);
    return rc;
}

typedef struct { uint8_t* d; uint64_t n, cap; } bytes;
static void b_push(bytes* b, uint8_t c) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 1024; b->d = (uint8_t*)realloc(b->d, b->cap); }
    b->d[b->n++] = c;
}

/* Window::generate_consensus, haplotype overload, window.cpp:176-428 */
static int window_hap(const seqview* sv, uint32_t nseq, int fasta, const vco_params* p,
                      bytes* cons, int* polished, vco_stats* stats) {
    uint32_t L = sv[0].len;
    cons->n = 0;
    if (nseq < 3) {                                       /* :188-192 */
        for (uint32_t i = 0; i < L; ++i) b_push(cons, sv[0].seq[i]);
        *polished = 0;
        return 0;
    }
    if (p->num_prune == 0) return -1;                     /* reference would spin for 2^32 rounds */
    graph G; g_init(&G);
    double total = 0.0;
    int rc = build_graph(&G, sv, nseq, L, p, &total, fasta, stats);
    if (rc) { g_free(&G); return rc; }
    if (stats) {
        if (G.n_nodes > stats->max_nodes) stats->max_nodes = G.n_nodes;
        if (G.n_edges > stats->max_edges) stats->max_edges = G.n_edges;
    }
    uint16_t window_len = (uint16_t)L;                    /* :216 */
    double avg = fasta ? 2.0 * total / window_len : 2.0 * total / window_len * 1000;   /* :301-309 */

    g_prune(&G, 0, p->min_confidence, p->min_support, avg);   /* :318 */
    graph* P = (graph*)malloc(sizeof(graph));
    g_largest_subgraph(&G, P);                            /* :319 */
    g_free(&G);
    stage_emit(2, 0, P, NULL);

    uint32_t offset = (uint32_t)(0.01 * L);
    alnvec A = {0, 0, 0};
    for (uint32_t k = 0; k + 1 < p->num_prune && rc == 0; ++k) {      /* :329-386 */
        for (uint32_t j = 0; j < nseq; ++j) {
            const seqview* s = &sv[j];
            if (j == 0 || (s->begin < offset && s->end > L - offset))
                rc = g_align(P, 1, p->match, p->mismatch, p->gap, s->seq, s->len, &A, stats);
            else
                rc = g_align(P, 0, p->sw_match, p->sw_mismatch, p->sw_gap, s->seq, s->len, &A, stats);
            if (rc) break;
            /* backbone: qualities_[0].first is never nullptr => quality branch (dummy '!' gives 0) */
            uint32_t* W = make_weights(s, j == 0 ? 1 : s->has_qual);
            g_add_weights(P, &A, s->len, W);
            free(W);
        }
        if (rc) break;
        stage_emit(3, k, P, NULL);
        g_prune(P, 0, p->min_confidence, p->min_support, avg);
        graph* Q = (graph*)malloc(sizeof(graph));
        g_largest_subgraph(P, Q);
        g_free(P); free(P);
        P = Q;
        stage_emit(2, k + 1, P, NULL);
    }
    if (rc == 0) {
        rc = g_align(P, 0, p->sw_match, p->sw_mismatch, p->sw_gap, sv[0].seq, sv[0].len, &A, stats);  /* :391 */
        if (rc == 0) {
            stage_emit(4, 0, P, &A);
            for (uint32_t k = 0; k < A.n / 2; ++k) {      /* GenerateCorrectedSequence, graph.cpp:1167-1179 */
                if (A.d[2 * k] == -1) continue;
                b_push(cons, (uint8_t)P->decoder[P->code[A.d[2 * k]]]);
            }
        }
    }
    free(A.d);
    g_free(P); free(P);
    *polished = 1;
    return rc;
}

/* Window::generate_consensus(engine, trim), window.cpp:74-174 */
static int window_linear(const seqview* sv, uint32_t nseq, const vco_params* p,
                         bytes* cons, int* polished, vco_stats* stats) {
    uint32_t L = sv[0].len;
    cons->n = 0;
    if (nseq < 3) {
        for (uint32_t i = 0; i < L; ++i) b_push(cons, sv[0].seq[i]);
        *polished = 0;
        return 0;
    }
    graph G; g_init(&G);
    int rc = build_graph(&G, sv, nseq, L, p, NULL, 0, stats);
    if (rc) { g_free(&G); return rc; }
    g_heaviest_bundle(&G);                                /* GenerateConsensus, graph.cpp:450-486 */
    uint32_t n = G.consensus.n;
    uint32_t* cov = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t* stamp = (uint32_t*)calloc((size_t)G.nseq + 1, sizeof(uint32_t));
    uint32_t tick = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = G.consensus.d[i];
        cov[i] = g_coverage(&G, v, stamp, ++tick);
        for (uint32_t t = 0; t < G.aligned[v].n; ++t) cov[i] += g_coverage(&G, G.aligned[v].d[t], stamp, ++tick);
    }
    int32_t begin = 0, end = (int32_t)n - 1;
    if (p->window_type == 1 && p->trim) {                 /* window.cpp:141-171 */
        uint32_t avgc = (nseq - 1) / 2;
        for (; begin < (int32_t)n; ++begin) if (cov[begin] >= avgc) break;
        for (; end >= 0; --end) if (cov[end] >= avgc) break;
        if (begin >= end) { begin = 0; end = (int32_t)n - 1; }   /* chimeric warning: untrimmed */
    }
    for (int32_t i = begin; i <= end; ++i) b_push(cons, (uint8_t)G.decoder[G.code[G.consensus.d[i]]]);
    free(cov); free(stamp);
    g_free(&G);
    *polished = 1;
    return 0;
}

int vco_run(const vco_batch* b, const vco_params* p, uint32_t w0, uint32_t w1,
            uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint8_t* polished,
            vco_stats* stats) {
    lut_init();
    bytes out = {0, 0, 0};
    int rc = 0;
    for (uint32_t w = w0; w < w1 && rc == 0; ++w) {
        uint32_t s0 = b->win_seq_off[w], s1 = b->win_seq_off[w + 1], ns = s1 - s0;
        seqview* sv = (seqview*)malloc((size_t)ns * sizeof(seqview));
        for (uint32_t i = 0; i < ns; ++i) {
            sv[i].seq = b->bases + b->seq_off[s0 + i];
            sv[i].qual = b->quals + b->seq_off[s0 + i];
            sv[i].len = (uint32_t)(b->seq_off[s0 + i + 1] - b->seq_off[s0 + i]);
            sv[i].begin = b->seq_begin[s0 + i];
            sv[i].end = b->seq_end[s0 + i];
            sv[i].has_qual = b->seq_has_qual[s0 + i];
        }
        int pol = 0;
        if (p->mode == 0) rc = window_hap(sv, ns, b->win_fasta[w], p, &out, &pol, stats);
        else              rc = window_linear(sv, ns, p, &out, &pol, stats);
        free(sv);
        if (rc) break;
        if (cons_off[w] + out.n > cons_cap) { rc = -2; break; }
        memcpy(cons + cons_off[w], out.d, out.n);
        cons_off[w + 1] = cons_off[w] + out.n;
        polished[w] = (uint8_t)pol;
    }
    free(out.d);
    return rc;
}

/* ------------------------------------------------------------------ spoa KAT flow */
/* Stage digests of window w (haplotype overload), see stage_emit; returns the number of records through n_rec. */
int vco_window_stages(const vco_batch* b, const vco_params* p, uint32_t w, uint64_t* rec, uint32_t rec_cap, uint32_t* n_rec) {
    stagevec sv = {rec, 0, rec_cap * 8, 0};
    uint64_t* off = (uint64_t*)calloc((size_t)w + 2, sizeof(uint64_t));      /* vco_run indexes both by window number */
    uint8_t* pol = (uint8_t*)calloc((size_t)w + 1, 1);
    const uint32_t s0 = b->win_seq_off[w], s1 = b->win_seq_off[w + 1];
    uint64_t cap = 4096;
    for (uint32_t s = s0; s < s1; ++s) cap += b->seq_off[s + 1] - b->seq_off[s];
    uint8_t* cons = (uint8_t*)malloc(cap);
    g_stages = &sv;
    int rc = vco_run(b, p, w, w + 1, off, cons, cap, pol, NULL);
    g_stages = NULL;
    free(cons); free(off); free(pol);
    *n_rec = sv.n / 8;
    if (rc) return rc;
    return sv.overflow ? -2 : 0;
}

static int spoa_build(graph* G, uint32_t n_seqs, const uint8_t* const* seqs, const uint32_t* lens,
                      const uint8_t* const* quals, int type, int m, int n, int g) {
    lut_init();
    alnvec A = {0, 0, 0};
    int rc = 0;
    for (uint32_t i = 0; i < n_seqs && rc == 0; ++i) {
        rc = g_align(G, type, m, n, g, seqs[i], lens[i], &A, NULL);
        if (rc) break;
        seqview s; s.seq = seqs[i]; s.qual = quals ? quals[i] : NULL; s.len = lens[i];
        s.begin = s.end = 0; s.has_qual = (quals && quals[i]) ? 1 : 0;
        uint32_t* W = make_weights(&s, s.has_qual);
        rc = g_add_alignment(G, &A, seqs[i], lens[i], W);
        free(W);
    }
    free(A.d);
    return rc;
}

int vco_spoa_consensus(uint32_t n_seqs, const uint8_t* const* seqs, const uint32_t* lens,
                       const uint8_t* const* quals, int type, int m, int n, int g,
                       uint8_t* out, uint32_t out_cap, uint32_t* out_len) {
    graph G; g_init(&G);
    int rc = spoa_build(&G, n_seqs, seqs, lens, quals, type, m, n, g);
    if (rc == 0) {
        g_heaviest_bundle(&G);
        *out_len = G.consensus.n;
        if (G.consensus.n > out_cap) rc = -2;
        else for (uint32_t i = 0; i < G.consensus.n; ++i) out[i] = (uint8_t)G.decoder[G.code[G.consensus.d[i]]];
    }
    g_free(&G);
    return rc;
}

int vco_spoa_align_probe(uint32_t n_seqs, const uint8_t* const* seqs, const uint32_t* lens,
                         const uint8_t* const* quals, int build_type, int m, int n, int g,
                         const uint8_t* query, uint32_t query_len, int query_type,
                         int32_t* pairs, uint32_t pairs_cap, uint32_t* n_pairs,
                         uint32_t* rank_to_node, uint32_t rank_cap, uint32_t* n_nodes) {
    graph G; g_init(&G);
    int rc = spoa_build(&G, n_seqs, seqs, lens, quals, build_type, m, n, g);
    if (rc == 0) {
        alnvec A = {0, 0, 0};
        rc = g_align(&G, query_type, m, n, g, query, query_len, &A, NULL);
        *n_pairs = A.n / 2; *n_nodes = G.n_nodes;
        if (rc == 0) {
            if (A.n / 2 > pairs_cap || G.n_nodes > rank_cap) rc = -2;
            else {
                memcpy(pairs, A.d, (size_t)A.n * sizeof(int32_t));
                memcpy(rank_to_node, G.rank.d, (size_t)G.rank.n * sizeof(uint32_t));
            }
        }
        free(A.d);
    }
    g_free(&G);
    return rc;
}
