// Window assembly and stitching on the host side of the C-ABI: what sits between the
// overlaps of a racon-style polisher and the batch of windows the device path consumes, and between the
// per-window results and the corrected sequences.  No device code; built into libvechat_hip.so and
// libvechat_host.so.  The reference's own implementation of this layer (src/polisher.cpp, src/overlap.cpp)
// cannot be compiled here (thread_pool / edlib are fetched at configure time), so parity of this
// file is UNPINNED: it is a restatement, cross-checked against an independent Python restatement in
// tests/test_windows.py (whose reverse complement rule is itself checked against the reference's
// racon::Sequence, tests/test_seqio.py).
//
//   vc_wb_add_overlap + breaking points  <- src/overlap.cpp:222-292  (find_breaking_points_from_cigar)
//   vc_wb_build                          <- src/polisher.cpp:389-462 (windows, layer filters, add_layer)
//                                           src/window.cpp:17-72     (createWindow / add_layer checks)
//                                           src/sequence.cpp:50-83   (reverse complement / reverse quality)
//   vc_wb_stitch                         <- src/polisher.cpp:520-547 (per-target concatenation, LN/RC/XC tags)
#include "vechat_hip.h"
#include "vc_hostbuf.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <sched.h>

namespace {

struct Seq {
    std::string name;
    const char* d = nullptr; const char* q = nullptr; size_t n = 0;   // bases / quality (nullptr: none): a view ...
    std::vector<char> data, qual;                  // ... of these when the sequence was added by copy (vc_wb_add_sequence; a moved
                                                   //     vector keeps its heap buffer, so the views survive the table's growth)
    std::string rc, rq;                            // built on demand (sequence.cpp:50-83)
};

struct Ovl {
    uint32_t q_id, t_id, q_begin, q_end, q_length, t_begin, t_end;
    int strand;
    std::vector<std::pair<uint32_t, uint32_t>> bp;  // (t_pos, q_pos), pairs of [first match, one past last match]
};

struct Layer { uint32_t q_id; int strand; uint32_t q0, q1, begin, end; };

struct Win { uint32_t target, rank, start, length; std::vector<Layer> layers; };

}  // namespace

struct vc_wb {
    uint32_t window_length = 500;
    double quality_threshold = 10.0;
    uint32_t n_targets = 0;
    std::vector<Seq> seqs;
    std::vector<Ovl> ovls;
    std::vector<Win> wins;
    std::vector<uint32_t> coverage;                 // targets_coverages_
    std::vector<uint64_t> wbytes, wseqs;            // layout of the last build: first byte / first sequence of every window (vc_wb_build_begin / _fill)
    std::string err;
    // flattened batch (owned here, handed out through vc_batch)
    std::vector<uint32_t> win_seq_off, seq_begin, seq_end;
    std::vector<uint64_t> seq_off;
    std::vector<uint8_t> seq_has_qual, win_fasta;
    VcHostBuf bases, quals;                        // written once by the flatten threads: no zero fill, huge pages on request
    std::vector<uint32_t> seq_orig;                 // add_layer() index of every stored sequence (0 = backbone)
    // stitched output
    std::vector<std::string> out_name, out_data;
};

namespace {

int fail(vc_wb* b, const char* msg) { b->err = msg; return VC_ERR_ARG; }

void make_reverse(Seq& s) {
    if (!s.rc.empty() || s.n == 0) return;
    s.rc.resize(s.n);
    for (size_t i = 0; i < s.n; ++i) {
        const char c = s.d[s.n - 1 - i];
        s.rc[i] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
    }
    if (s.q) s.rq.assign(std::reverse_iterator<const char*>(s.q + s.n), std::reverse_iterator<const char*>(s.q));
}

unsigned wb_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("VC_IO_THREADS")) n = (unsigned)std::max(1, atoi(e));
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // a container may see 256 cores and own 16
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(q) / period));
        fclose(f);
    }
    return std::max(1u, std::min(n, 32u));
}

// fn(k) for k in [0, n), on several threads when there is enough to do (contiguous ranges: results keep their order)
template <typename F>
void parallel_for(size_t n, size_t min_per_thread, F fn) {
    const size_t T = std::min<size_t>(wb_threads(), std::max<size_t>(1, n / std::max<size_t>(1, min_per_thread)));
    if (T <= 1) { for (size_t k = 0; k < n; ++k) fn(k); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t)
        th.emplace_back([=]() { for (size_t k = n * t / T; k < n * (t + 1) / T; ++k) fn(k); });
    for (auto& x : th) x.join();
}

// Breaking points of an overlap from its CIGAR (what src/overlap.cpp:222-292 computes base by base): for every
// window of the target the overlap touches, the first aligned (target, query) position inside it and the position
// one past the last aligned pair.  Done run by run here: a run of matches or deletions is cut at the window ends
// it crosses.  Window ends are the last target position of each window, and the overlap's own end.
// Returns false when the CIGAR does not consume exactly the query and target spans of the overlap record (then the
// breaking points would index past the sequences).
bool breaking_points_from_cigar(Ovl& o, const char* cigar, uint32_t window_length) {
    std::vector<int64_t> ends;
    for (uint64_t e = window_length; e < o.t_end; e += window_length)
        if (e > o.t_begin) ends.push_back((int64_t)e - 1);
    ends.push_back((int64_t)o.t_end - 1);

    size_t w = 0;
    bool open = false;                                   // a first aligned pair of the current window has been seen
    std::pair<uint32_t, uint32_t> first(0, 0), last(0, 0);
    int64_t tpos = (int64_t)o.t_begin - 1;               // last consumed target / query position
    int64_t qpos = (int64_t)(o.strand ? o.q_length - o.q_end : o.q_begin) - 1;
    auto close_window = [&]() {
        if (open) { o.bp.push_back(first); o.bp.push_back(last); }
        open = false;
        ++w;
    };
    // consume `len` target positions (aligned to query positions or not), stopping at every window end on the way
    auto advance = [&](uint64_t len, bool aligned) {
        while (len) {
            uint64_t k = len;
            if (w < ends.size() && ends[w] >= tpos) k = std::min<uint64_t>(len, (uint64_t)std::max<int64_t>(ends[w] - tpos, 1));
            if (aligned) {
                if (!open) { open = true; first = {(uint32_t)(tpos + 1), (uint32_t)(qpos + 1)}; }
                qpos += (int64_t)k;
            }
            tpos += (int64_t)k;
            if (aligned) last = {(uint32_t)(tpos + 1), (uint32_t)(qpos + 1)};
            len -= k;
            if (w < ends.size() && tpos == ends[w]) close_window();
        }
    };
    o.bp.reserve(2 * ends.size());
    for (const char* p = cigar; *p;) {
        const char* e = p;
        uint64_t len = 0;
        while (*e >= '0' && *e <= '9') { len = len * 10 + (uint64_t)(*e - '0'); ++e; }
        if (e == p || !*e) break;                        // no run length / no operation: stop like atoi-driven parsing would
        switch (*e) {
            case 'M': case '=': case 'X': advance(len, true); break;
            case 'D': case 'N': advance(len, false); break;
            case 'I': qpos += (int64_t)len; break;
            default: break;                              // S, H, P consume nothing here (clips are in q_begin / q_end)
        }
        p = e + 1;
    }
    const int64_t q_last = (int64_t)(o.strand ? o.q_length - o.q_begin : o.q_end) - 1;      // last query / target position of the span
    return qpos == q_last && tpos == (int64_t)o.t_end - 1;
}

}  // namespace

extern "C" {

vc_wb* vc_wb_create(uint32_t window_length, double quality_threshold) {
    if (window_length == 0) return nullptr;
    vc_wb* b = new vc_wb();
    b->window_length = window_length;
    b->quality_threshold = quality_threshold;
    return b;
}

void vc_wb_destroy(vc_wb* b) { delete b; }

const char* vc_wb_last_error(const vc_wb* b) { return b ? b->err.c_str() : "null builder"; }

int vc_wb_add_sequence(vc_wb* b, const char* name, const char* data, uint32_t length, const char* quality) {
    if (!b || !data || length == 0) return -1;
    b->seqs.emplace_back();
    Seq& s = b->seqs.back();
    s.name = name ? name : "";
    s.data.assign(data, data + length);
    if (quality) s.qual.assign(quality, quality + length);
    s.d = s.data.data(); s.q = quality ? s.qual.data() : nullptr; s.n = length;
    return (int)b->seqs.size() - 1;
}

// the same without a copy: `data` / `quality` stay owned by the caller and must outlive the builder (vc_io_load hands in the
// buffers of its vc_seqset)
int vc_wb_add_sequence_view(vc_wb* b, const char* name, uint32_t name_len, const char* data, uint32_t length, const char* quality) {
    if (!b || !data || length == 0) return -1;
    b->seqs.emplace_back();
    Seq& s = b->seqs.back();
    s.name.assign(name ? name : "", name ? name_len : 0);
    s.d = data; s.q = quality; s.n = length;
    return (int)b->seqs.size() - 1;
}

int vc_wb_set_targets(vc_wb* b, uint32_t n_targets) {
    if (!b || n_targets > b->seqs.size()) return VC_ERR_ARG;
    b->n_targets = n_targets;
    return VC_OK;
}

namespace {
// the checks of one overlap record (overlap.cpp:139-146,161-167) and its breaking points; nullptr or the complaint
const char* make_overlap(const vc_wb* b, Ovl& o, const char* cigar) {
    if (o.q_id >= b->seqs.size() || o.t_id >= b->n_targets) return "overlap refers to an unknown sequence";
    if (o.q_length != b->seqs[o.q_id].n) return "unequal lengths in sequence and overlap record";
    if (o.q_begin > o.q_end || o.q_end > o.q_length || o.t_begin > o.t_end || o.t_end > b->seqs[o.t_id].n) return "overlap coordinates out of range";
    if (!breaking_points_from_cigar(o, cigar, b->window_length)) return "CIGAR does not match the overlap's query / target spans";
    return nullptr;
}
}  // namespace

int vc_wb_add_overlap(vc_wb* b, uint32_t q_id, uint32_t t_id, int strand, uint32_t q_begin, uint32_t q_end,
                      uint32_t q_length, uint32_t t_begin, uint32_t t_end, const char* cigar) {
    if (!b || !cigar) return VC_ERR_ARG;
    Ovl o{q_id, t_id, q_begin, q_end, q_length, t_begin, t_end, strand ? 1 : 0, {}};
    if (const char* m = make_overlap(b, o, cigar)) return fail(b, m);
    b->ovls.emplace_back(std::move(o));
    return VC_OK;
}

// n overlaps at once, in the order given (the breaking points of different overlaps are independent: several threads)
int vc_wb_add_overlaps(vc_wb* b, uint64_t n, const uint32_t* q_id, const uint32_t* t_id, const uint8_t* strand, const uint32_t* q_begin,
                       const uint32_t* q_end, const uint32_t* q_length, const uint32_t* t_begin, const uint32_t* t_end, const char* const* cigar) {
    if (!b || (n && (!q_id || !t_id || !strand || !q_begin || !q_end || !q_length || !t_begin || !t_end || !cigar))) return VC_ERR_ARG;
    const size_t base = b->ovls.size();
    b->ovls.resize(base + n);
    std::vector<const char*> msg(n, nullptr);
    parallel_for(n, 256, [&](size_t k) {
        Ovl& o = b->ovls[base + k];
        o = Ovl{q_id[k], t_id[k], q_begin[k], q_end[k], q_length[k], t_begin[k], t_end[k], strand[k] ? 1 : 0, {}};
        msg[k] = cigar[k] ? make_overlap(b, o, cigar[k]) : "overlap without a CIGAR";
    });
    for (size_t k = 0; k < n; ++k) if (msg[k]) { b->ovls.resize(base); return fail(b, msg[k]); }
    return VC_OK;
}

uint32_t vc_wb_n_breaking_points(const vc_wb* b, uint32_t overlap) {
    return b && overlap < b->ovls.size() ? (uint32_t)b->ovls[overlap].bp.size() : 0;
}

void vc_wb_breaking_points(const vc_wb* b, uint32_t overlap, uint32_t* t_pos, uint32_t* q_pos) {
    if (!b || overlap >= b->ovls.size()) return;
    const auto& bp = b->ovls[overlap].bp;
    for (size_t i = 0; i < bp.size(); ++i) { t_pos[i] = bp[i].first; q_pos[i] = bp[i].second; }
}

// vc_wb_build in two steps, for a caller that hands the windows to the device in slices ("files in, FASTA out"): _begin lays
// the batch out -- every offset, every buffer -- and _fill writes the bytes of the windows [w_lo, w_hi); slice i + 1 is filled while the
// device works on slice i, as the reference's accelerated polisher fills its next batch while one computes (src/cuda/cudapolisher.cpp:246-277).
static int wb_build_impl(vc_wb* b, vc_batch* out, bool fill_all);
int vc_wb_build(vc_wb* b, vc_batch* out) { return wb_build_impl(b, out, true); }
int vc_wb_build_begin(vc_wb* b, vc_batch* out) { return wb_build_impl(b, out, false); }

static void wb_fill(vc_wb* b, size_t w_lo, size_t w_hi);
int vc_wb_build_fill(vc_wb* b, uint32_t w_lo, uint32_t w_hi) {
    if (!b || w_lo > w_hi || w_hi > b->wins.size() || b->wbytes.size() != b->wins.size() + 1) return VC_ERR_ARG;
    wb_fill(b, w_lo, w_hi);
    return VC_OK;
}

static int wb_build_impl(vc_wb* b, vc_batch* out, bool fill_all) {
    if (!b || !out) return VC_ERR_ARG;
    const uint32_t W = b->window_length;
    b->wins.clear();
    // polisher.cpp:389-404: windows of every target in order
    std::vector<uint64_t> first_window(b->n_targets + 1, 0);
    for (uint32_t t = 0; t < b->n_targets; ++t) {
        const uint32_t len = (uint32_t)b->seqs[t].n;
        uint32_t k = 0;
        for (uint32_t j = 0; j < len; j += W, ++k) b->wins.push_back(Win{t, k, j, std::min(j + W, len) - j, {}});
        first_window[t + 1] = first_window[t] + k;
    }
    b->coverage.assign(b->n_targets, 0);
    // reverse complements of the reads some overlap needs on the other strand (sequence.cpp:50-83), each made once
    {
        std::vector<uint32_t> need;
        std::vector<uint8_t> mark(b->seqs.size(), 0);
        for (const auto& o : b->ovls) if (o.strand && !mark[o.q_id]) { mark[o.q_id] = 1; need.push_back(o.q_id); }
        parallel_for(need.size(), 16, [&](size_t k) { make_reverse(b->seqs[need[k]]); });
    }
    // polisher.cpp:408-459: the layers every overlap contributes (independent of each other: several threads) ...
    struct Made { uint64_t wid; Layer l; };
    std::vector<std::vector<Made>> made(b->ovls.size());
    std::vector<const char*> msg(b->ovls.size(), nullptr);
    parallel_for(b->ovls.size(), 64, [&](size_t oi) {
        const Ovl& o = b->ovls[oi];
        const Seq& s = b->seqs[o.q_id];
        for (size_t j = 0; j + 1 < o.bp.size(); j += 2) {
            const uint32_t q0 = o.bp[j].second, q1 = o.bp[j + 1].second;
            if (q1 < q0 || q1 > s.n) { msg[oi] = "breaking point beyond the end of the read"; return; }
            if ((double)(q1 - q0) < 0.02 * W) continue;                                   // :416
            if (s.q) {                                                                    // :420-434
                const char* q = o.strand ? s.rq.data() : s.q;
                double average_quality = 0;
                for (uint32_t k = q0; k < q1; ++k) average_quality += (uint32_t)(uint8_t)q[k] - 33;
                average_quality /= q1 - q0;
                if (average_quality < b->quality_threshold) continue;
            }
            const uint64_t wid = first_window[o.t_id] + o.bp[j].first / W;                // :436-439
            const uint32_t wstart = (o.bp[j].first / W) * W;
            const uint32_t begin = o.bp[j].first - wstart, end = o.bp[j + 1].first - wstart - 1;
            const Win& win = b->wins[wid];
            // window.cpp:47-72 (add_layer): empty or single-column layers are dropped, bad positions are fatal
            if (q1 == q0 || begin == end) continue;
            if (begin >= end || begin > win.length || end > win.length) { msg[oi] = "layer begin and end positions are invalid"; return; }
            made[oi].push_back(Made{wid, Layer{o.q_id, o.strand, q0, q1, begin, end}});
        }
    });
    // ... appended to their windows in overlap order
    for (size_t oi = 0; oi < b->ovls.size(); ++oi) {
        if (msg[oi]) return fail(b, msg[oi]);
        ++b->coverage[b->ovls[oi].t_id];
        for (const Made& m : made[oi]) b->wins[m.wid].layers.push_back(m.l);
    }
    // flatten, layers in the reference's rank order (window.cpp:203-210 via vc_rank_layers): sizes first, then the bytes of the
    // windows side by side
    auto& B = *b;
    const size_t nw = b->wins.size();
    std::vector<uint64_t>& wbytes = b->wbytes; std::vector<uint64_t>& wseqs = b->wseqs;
    wbytes.assign(nw + 1, 0); wseqs.assign(nw + 1, 0);
    for (size_t w = 0; w < nw; ++w) {
        const Win& win = b->wins[w];
        uint64_t by = win.length;
        for (const Layer& l : win.layers) by += l.q1 - l.q0;
        wbytes[w + 1] = wbytes[w] + by;
        wseqs[w + 1] = wseqs[w] + win.layers.size() + 1;
    }
    const uint64_t ns = wseqs[nw], nb = wbytes[nw];
    B.win_seq_off.assign(nw + 1, 0); B.seq_off.assign(ns + 1, 0);
    B.seq_begin.assign(ns, 0); B.seq_end.assign(ns, 0); B.seq_has_qual.assign(ns, 0); B.seq_orig.assign(ns, 0);
    B.bases.alloc(nb); B.quals.alloc(nb); B.win_fasta.assign(nw, 0);
    for (size_t w = 0; w <= nw; ++w) B.win_seq_off[w] = (uint32_t)wseqs[w];
    B.seq_off[ns] = nb;
    // (sequence offsets of every window now, so that a slice of the batch is complete as soon as its own windows are filled)
    // (the order of the layers inside a window is settled in wb_fill, and the offsets of its sequences with it; here only every window's
    // FIRST offset, which is what the slice in front of it ends on)
    for (size_t w = 0; w < nw; ++w) B.seq_off[wseqs[w]] = wbytes[w];
    if (fill_all) wb_fill(b, 0, nw);
    out->n_windows = (uint32_t)nw;
    out->win_seq_off = B.win_seq_off.data(); out->seq_off = B.seq_off.data();
    out->seq_begin = B.seq_begin.data(); out->seq_end = B.seq_end.data(); out->seq_has_qual = B.seq_has_qual.data();
    out->bases = (const uint8_t*)B.bases.data(); out->quals = (const uint8_t*)B.quals.data(); out->win_fasta = B.win_fasta.data();
    return VC_OK;
}

static void wb_fill(vc_wb* b, size_t w_lo, size_t w_hi) {
    auto& B = *b;
    const uint32_t W = b->window_length;
    const std::vector<uint64_t>& wbytes = b->wbytes; const std::vector<uint64_t>& wseqs = b->wseqs;
    parallel_for(w_hi - w_lo, 8, [&](size_t wi) {
        const size_t w = w_lo + wi;
        const Win& win = b->wins[w];
        const Seq& t = b->seqs[win.target];
        const uint32_t n = (uint32_t)win.layers.size() + 1;
        std::vector<uint32_t> begins(n, 0), rank(n);
        for (uint32_t i = 1; i < n; ++i) begins[i] = win.layers[i - 1].begin;
        vc_rank_layers(begins.data(), n, rank.data());
        // window.cpp:223 on the pointers polisher.cpp:397-400 hands over: a FASTA target gets the shared
        // dummy string of window_length '!' (equal only for a full-length window); a FASTQ target gets a
        // pointer into its own quality string, whose C-string runs to the end of that read
        bool fasta;
        if (!t.q) fasta = win.length == W;
        else {
            fasta = win.start + win.length == t.n;
            for (uint32_t i = 0; fasta && i < win.length; ++i) fasta = t.q[win.start + i] == '!';
        }
        B.win_fasta[w] = fasta ? 1 : 0;
        uint64_t off = wbytes[w], si = wseqs[w];
        for (uint32_t k = 0; k < n; ++k, ++si) {
            const uint32_t i = rank[k];
            B.seq_orig[si] = i;
            B.seq_off[si] = off;
            if (i == 0) {
                std::memcpy(B.bases.data() + off, t.d + win.start, win.length);
                if (!t.q) std::memset(B.quals.data() + off, '!', win.length);
                else std::memcpy(B.quals.data() + off, t.q + win.start, win.length);
                B.seq_has_qual[si] = 1;
                off += win.length;
            } else {
                const Layer& l = win.layers[i - 1];
                const Seq& s = b->seqs[l.q_id];
                const uint32_t len = l.q1 - l.q0;
                std::memcpy(B.bases.data() + off, (l.strand ? s.rc.data() : s.d) + l.q0, len);
                if (!s.q) std::memset(B.quals.data() + off, '!', len);
                else std::memcpy(B.quals.data() + off, (l.strand ? s.rq.data() : s.q) + l.q0, len);
                B.seq_begin[si] = l.begin; B.seq_end[si] = l.end; B.seq_has_qual[si] = s.q ? 1 : 0;
                off += len;
            }
        }
    });
}

const uint32_t* vc_wb_seq_orig(const vc_wb* b) { return b ? b->seq_orig.data() : nullptr; }
uint32_t vc_wb_n_windows(const vc_wb* b) { return b ? (uint32_t)b->wins.size() : 0; }
uint32_t vc_wb_window_target(const vc_wb* b, uint32_t w) { return b && w < b->wins.size() ? b->wins[w].target : 0; }
// (target, rank) of every window of the last build at once
void vc_wb_window_ids(const vc_wb* b, uint32_t* target, uint32_t* rank) {
    if (!b) return;
    for (size_t w = 0; w < b->wins.size(); ++w) { if (target) target[w] = b->wins[w].target; if (rank) rank[w] = b->wins[w].rank; }
}
uint32_t vc_wb_window_rank(const vc_wb* b, uint32_t w) { return b && w < b->wins.size() ? b->wins[w].rank : 0; }

// polisher.cpp:520-547.  status[w] == VC_WIN_OK counts as polished; anything above VC_WIN_UNPOLISHED is an error
// of the caller (such windows have to be rerun or computed elsewhere before stitching).
int vc_wb_stitch(vc_wb* b, const vc_result* res, int drop_unpolished, int fragment_correction) {
    if (!b || !res || !res->cons_off || !res->cons || !res->status) return VC_ERR_ARG;
    b->out_name.clear(); b->out_data.clear();
    std::string polished;
    uint32_t num_polished = 0;
    for (size_t i = 0; i < b->wins.size(); ++i) {
        if (res->status[i] > VC_WIN_UNPOLISHED) return fail(b, "a window without a result cannot be stitched");
        num_polished += res->status[i] == VC_WIN_OK ? 1 : 0;
        polished.append((const char*)res->cons + res->cons_off[i], res->cons_off[i + 1] - res->cons_off[i]);
        if (i == b->wins.size() - 1 || b->wins[i + 1].rank == 0) {
            const double ratio = num_polished / (double)(b->wins[i].rank + 1);
            if (!drop_unpolished || ratio > 0) {
                std::string tags = fragment_correction ? "r" : "";
                tags += " LN:i:" + std::to_string(polished.size());
                tags += " RC:i:" + std::to_string(b->coverage[b->wins[i].target]);
                tags += " XC:f:" + std::to_string(ratio);
                b->out_name.emplace_back(b->seqs[b->wins[i].target].name + tags);
                b->out_data.emplace_back(polished);
            }
            num_polished = 0;
            polished.clear();
        }
    }
    return VC_OK;
}

uint32_t vc_wb_n_polished(const vc_wb* b) { return b ? (uint32_t)b->out_name.size() : 0; }
const char* vc_wb_polished_name(const vc_wb* b, uint32_t i) { return b && i < b->out_name.size() ? b->out_name[i].c_str() : ""; }
const char* vc_wb_polished_data(const vc_wb* b, uint32_t i, uint64_t* length) {
    if (!b || i >= b->out_data.size()) { if (length) *length = 0; return ""; }
    if (length) *length = b->out_data[i].size();
    return b->out_data[i].c_str();
}

}  // extern "C"
