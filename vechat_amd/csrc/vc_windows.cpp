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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace {

struct Seq {
    std::string name, data, qual, rc, rq;          // rc / rq built on demand (sequence.cpp:50-83)
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
    std::string err;
    // flattened batch (owned here, handed out through vc_batch)
    std::vector<uint32_t> win_seq_off, seq_begin, seq_end;
    std::vector<uint64_t> seq_off;
    std::vector<uint8_t> seq_has_qual, bases, quals, win_fasta;
    std::vector<uint32_t> seq_orig;                 // add_layer() index of every stored sequence (0 = backbone)
    // stitched output
    std::vector<std::string> out_name, out_data;
};

namespace {

int fail(vc_wb* b, const char* msg) { b->err = msg; return VC_ERR_ARG; }

void make_reverse(Seq& s) {
    if (!s.rc.empty() || s.data.empty()) return;
    s.rc.reserve(s.data.size());
    for (size_t i = s.data.size(); i-- > 0;) {
        switch (s.data[i]) {
            case 'A': s.rc += 'T'; break;
            case 'T': s.rc += 'A'; break;
            case 'C': s.rc += 'G'; break;
            case 'G': s.rc += 'C'; break;
            default:  s.rc += s.data[i]; break;
        }
    }
    s.rq.assign(s.qual.rbegin(), s.qual.rend());
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
    for (const char* p = cigar; *p;) {
        char* e = nullptr;
        const uint64_t len = std::strtoull(p, &e, 10);
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
    Seq s;
    s.name = name ? name : "";
    s.data.assign(data, length);
    if (quality) s.qual.assign(quality, length);
    b->seqs.emplace_back(std::move(s));
    return (int)b->seqs.size() - 1;
}

int vc_wb_set_targets(vc_wb* b, uint32_t n_targets) {
    if (!b || n_targets > b->seqs.size()) return VC_ERR_ARG;
    b->n_targets = n_targets;
    return VC_OK;
}

int vc_wb_add_overlap(vc_wb* b, uint32_t q_id, uint32_t t_id, int strand, uint32_t q_begin, uint32_t q_end,
                      uint32_t q_length, uint32_t t_begin, uint32_t t_end, const char* cigar) {
    if (!b || !cigar) return VC_ERR_ARG;
    if (q_id >= b->seqs.size() || t_id >= b->n_targets) return fail(b, "overlap refers to an unknown sequence");
    // overlap.cpp:139-146,161-167: the lengths in the overlap record must match the sequences
    if (q_length != b->seqs[q_id].data.size()) return fail(b, "unequal lengths in sequence and overlap record");
    if (q_begin > q_end || q_end > q_length || t_begin > t_end || t_end > b->seqs[t_id].data.size())
        return fail(b, "overlap coordinates out of range");
    Ovl o{q_id, t_id, q_begin, q_end, q_length, t_begin, t_end, strand ? 1 : 0, {}};
    if (!breaking_points_from_cigar(o, cigar, b->window_length))
        return fail(b, "CIGAR does not match the overlap's query / target spans");
    b->ovls.emplace_back(std::move(o));
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

int vc_wb_build(vc_wb* b, vc_batch* out) {
    if (!b || !out) return VC_ERR_ARG;
    const uint32_t W = b->window_length;
    b->wins.clear();
    // polisher.cpp:389-404: windows of every target in order
    std::vector<uint64_t> first_window(b->n_targets + 1, 0);
    for (uint32_t t = 0; t < b->n_targets; ++t) {
        const uint32_t len = (uint32_t)b->seqs[t].data.size();
        uint32_t k = 0;
        for (uint32_t j = 0; j < len; j += W, ++k) b->wins.push_back(Win{t, k, j, std::min(j + W, len) - j, {}});
        first_window[t + 1] = first_window[t] + k;
    }
    b->coverage.assign(b->n_targets, 0);
    // polisher.cpp:408-459: layers, in overlap order
    for (auto& o : b->ovls) {
        ++b->coverage[o.t_id];
        Seq& s = b->seqs[o.q_id];
        if (o.strand) make_reverse(s);
        for (size_t j = 0; j + 1 < o.bp.size(); j += 2) {
            const uint32_t q0 = o.bp[j].second, q1 = o.bp[j + 1].second;
            if (q1 < q0 || q1 > b->seqs[o.q_id].data.size()) return fail(b, "breaking point beyond the end of the read");
            if ((double)(q1 - q0) < 0.02 * W) continue;                                   // :416
            if (!s.qual.empty()) {                                                        // :420-434
                const std::string& q = o.strand ? s.rq : s.qual;
                double average_quality = 0;
                for (uint32_t k = q0; k < q1; ++k) average_quality += (uint32_t)(uint8_t)q[k] - 33;
                average_quality /= q1 - q0;
                if (average_quality < b->quality_threshold) continue;
            }
            const uint64_t wid = first_window[o.t_id] + o.bp[j].first / W;                // :436-439
            const uint32_t wstart = (o.bp[j].first / W) * W;
            const uint32_t begin = o.bp[j].first - wstart, end = o.bp[j + 1].first - wstart - 1;
            Win& win = b->wins[wid];
            // window.cpp:47-72 (add_layer): empty or single-column layers are dropped, bad positions are fatal
            if (q1 == q0 || begin == end) continue;
            if (begin >= end || begin > win.length || end > win.length) return fail(b, "layer begin and end positions are invalid");
            win.layers.push_back(Layer{o.q_id, o.strand, q0, q1, begin, end});
        }
    }
    // flatten, layers in the reference's rank order (window.cpp:203-210 via vc_rank_layers)
    auto& B = *b;
    B.win_seq_off.assign(1, 0); B.seq_off.assign(1, 0);
    B.seq_begin.clear(); B.seq_end.clear(); B.seq_has_qual.clear(); B.bases.clear(); B.quals.clear(); B.win_fasta.clear(); B.seq_orig.clear();
    std::vector<uint32_t> begins, rank;
    for (const Win& win : b->wins) {
        const Seq& t = b->seqs[win.target];
        const uint32_t n = (uint32_t)win.layers.size() + 1;
        begins.assign(n, 0);
        for (uint32_t i = 1; i < n; ++i) begins[i] = win.layers[i - 1].begin;
        rank.resize(n);
        vc_rank_layers(begins.data(), n, rank.data());
        // window.cpp:223 on the pointers polisher.cpp:397-400 hands over: a FASTA target gets the shared
        // dummy string of window_length '!' (equal only for a full-length window); a FASTQ target gets a
        // pointer into its own quality string, whose C-string runs to the end of that read
        bool fasta;
        if (t.qual.empty()) fasta = win.length == W;
        else {
            fasta = win.start + win.length == t.qual.size();
            for (uint32_t i = 0; fasta && i < win.length; ++i) fasta = t.qual[win.start + i] == '!';
        }
        B.win_fasta.push_back(fasta ? 1 : 0);
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t i = rank[k];
            B.seq_orig.push_back(i);
            if (i == 0) {
                B.bases.insert(B.bases.end(), t.data.begin() + win.start, t.data.begin() + win.start + win.length);
                if (t.qual.empty()) B.quals.insert(B.quals.end(), win.length, (uint8_t)'!');
                else B.quals.insert(B.quals.end(), t.qual.begin() + win.start, t.qual.begin() + win.start + win.length);
                B.seq_begin.push_back(0); B.seq_end.push_back(0); B.seq_has_qual.push_back(1);
                B.seq_off.push_back(B.seq_off.back() + win.length);
            } else {
                const Layer& l = win.layers[i - 1];
                const Seq& s = b->seqs[l.q_id];
                const std::string& d = l.strand ? s.rc : s.data;
                B.bases.insert(B.bases.end(), d.begin() + l.q0, d.begin() + l.q1);
                if (s.qual.empty()) B.quals.insert(B.quals.end(), l.q1 - l.q0, (uint8_t)'!');
                else { const std::string& q = l.strand ? s.rq : s.qual; B.quals.insert(B.quals.end(), q.begin() + l.q0, q.begin() + l.q1); }
                B.seq_begin.push_back(l.begin); B.seq_end.push_back(l.end); B.seq_has_qual.push_back(s.qual.empty() ? 0 : 1);
                B.seq_off.push_back(B.seq_off.back() + (l.q1 - l.q0));
            }
        }
        B.win_seq_off.push_back((uint32_t)B.seq_begin.size());
    }
    out->n_windows = (uint32_t)b->wins.size();
    out->win_seq_off = B.win_seq_off.data(); out->seq_off = B.seq_off.data();
    out->seq_begin = B.seq_begin.data(); out->seq_end = B.seq_end.data(); out->seq_has_qual = B.seq_has_qual.data();
    out->bases = B.bases.data(); out->quals = B.quals.data(); out->win_fasta = B.win_fasta.data();
    return VC_OK;
}

const uint32_t* vc_wb_seq_orig(const vc_wb* b) { return b ? b->seq_orig.data() : nullptr; }
uint32_t vc_wb_n_windows(const vc_wb* b) { return b ? (uint32_t)b->wins.size() : 0; }
uint32_t vc_wb_window_target(const vc_wb* b, uint32_t w) { return b && w < b->wins.size() ? b->wins[w].target : 0; }
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
