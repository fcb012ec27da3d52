// File formats either side of the hot path, on the host side of the C ABI: what the reference's Polisher
// does with bioparser before a single window exists.  No device code; built into libvechat_hip.so and libvechat_host.so.
//
//   vc_io_read_sequences  <- src/polisher.cpp:77-138 (parser selection by extension) + vendor/spoa/vendor/bioparser
//                            (FASTA / FASTQ records, names cut at the first whitespace) + src/sequence.cpp:19-42
//                            (upper-casing; a quality string that is all '!' counts as none)
//   vc_io_read_overlaps   <- src/overlap.cpp:14-27 (MHAP), :29-42 (PAF), :44-110 (SAM: unmapped flag, strand, clips, lengths,
//                            error) -- the record constructors; a PAF `cg:Z:` tag supplies the CIGAR, without one the record
//                            is aligned on the device first (vc_align; the reference calls edlib there, overlap.cpp:205-220)
//   vc_io_load            <- src/polisher.cpp:207-352 (Polisher::initialize: reads that are also targets share one record,
//                            self-overlaps and overlaps above the error threshold are dropped, window type from the mean
//                            read length, name -> id resolution of overlap.cpp:129-177)
//
// Sequence ingest is pinned against the reference's own bioparser + racon::Sequence (oracle/ref_seqparse.cpp,
// tests/test_seqio.py); the overlap constructors are restatements (src/overlap.cpp needs edlib.h, which is not in the tree),
// cross-checked against the independent Python restatement in vechat_amd/seqio.py on every format.
// Files are read whole (mmap, or zlib for .gz) and cut into records by memchr; plain files are parsed by several threads.
#include "vechat_hip.h"
#include "vc_hostbuf.h"

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <chrono>
// development: VC_IO_TIMING=1 prints the time of every stage of a call to stderr
#define VC_IO_T0 auto io_t0_ = std::chrono::steady_clock::now(); const bool io_tm_ = getenv("VC_IO_TIMING") != nullptr
#define VC_IO_TICK(what) do { if (io_tm_) { auto n_ = std::chrono::steady_clock::now(); fprintf(stderr, "  [vc_io] %-10s %.3f s\n", what, std::chrono::duration<double>(n_ - io_t0_).count()); io_t0_ = n_; } } while (0)

namespace {

// ------------------------------------------------------------------------------------------------ file contents
struct Blob {
    const char* p = nullptr; size_t n = 0;
    std::string owned;          // .gz: inflated here
    void* map = nullptr; size_t map_n = 0;
    ~Blob() { if (map) munmap(map, map_n); }
    // a thread about to read [a, b) of a mapped file asks for its pages in one call (threads faulting page by page queue up behind
    // each other); a kernel without MADV_POPULATE_READ just says no and the pages come as they are touched
    void populate(const char* a, const char* b) const {
        if (!map || b <= a) return;
        const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE), lo = (uintptr_t)a & ~(pg - 1), hi = std::min(((uintptr_t)b + pg - 1) & ~(pg - 1), (uintptr_t)map + ((map_n + pg - 1) & ~(pg - 1)));
#ifdef MADV_POPULATE_READ
        if (hi > lo) (void)madvise((void*)lo, (size_t)(hi - lo), MADV_POPULATE_READ);
#endif
    }
};

bool ends_with(const std::string& s, const char* suf) {
    const size_t k = std::strlen(suf);
    return s.size() >= k && s.compare(s.size() - k, k, suf) == 0;
}

bool load_file(const char* path, Blob& b, std::string& err) {
    const std::string ps(path);
    if (ends_with(ps, ".gz")) {
        gzFile f = gzopen(path, "rb");
        if (!f) { err = ps + ": cannot open"; return false; }
        gzbuffer(f, 1 << 20);
        std::string& o = b.owned;
        std::vector<char> buf(1 << 22);
        for (;;) {
            const int k = gzread(f, buf.data(), (unsigned)buf.size());
            if (k < 0) { err = ps + ": gzip read error"; gzclose(f); return false; }
            if (k == 0) break;
            o.append(buf.data(), (size_t)k);
        }
        gzclose(f);
        b.p = o.data(); b.n = o.size();
        return true;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = ps + ": cannot open"; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); err = ps + ": cannot stat"; return false; }
    if (st.st_size == 0) { close(fd); b.p = ""; b.n = 0; return true; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);                  // (pages: Blob::populate, by the threads that read them)
    close(fd);
    if (m == MAP_FAILED) { err = ps + ": cannot map"; return false; }
    b.map = m; b.map_n = (size_t)st.st_size; b.p = (const char*)m; b.n = b.map_n;
    return true;
}

inline const char* line_end(const char* p, const char* e) {
    const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
    return q ? q : e;
}
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
// [p, e) without leading / trailing whitespace (Python's bytes.strip())
inline void strip(const char*& p, const char*& e) {
    while (p < e && is_space(*p)) ++p;
    while (e > p && is_space(e[-1])) --e;
}
unsigned n_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("VC_IO_THREADS")) n = (unsigned)std::max(1, atoi(e));
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    // cgroup CPU quota (a container may see 256 cores and own 16)
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(q) / period));
        fclose(f);
    }
    return std::max(1u, std::min(n, 32u));
}

}  // namespace

// ------------------------------------------------------------------------------------------------ sequences
namespace {
using RawBuf = VcHostBuf;                            // bytes without the zero fill of a std::string (every thread first-touches its own part)
}  // namespace

struct vc_seqset {
    std::string names;
    RawBuf data, qual;                               // concatenated; qual has the data's offsets ('!' where a record has none)
    std::vector<uint64_t> name_off{0}, data_off{0};
    std::vector<uint8_t> has_qual;
    std::vector<uint64_t> length;                    // == data_off differences when the data is kept
    bool names_only = false;
    std::string err;
    size_t size() const { return has_qual.size(); }
    std::string name(size_t i) const { return names.substr(name_off[i], name_off[i + 1] - name_off[i]); }
};

namespace {

struct RecRef {                                      // one record as pass 1 found it: spans into the file image, nothing copied yet
    const char* name; uint32_t name_len;
    char kind;                                       // '>' or '@'
    const char* seq; const char* seq_end;            // '@': the stripped sequence line; '>': first sequence line .. start of the next record
    const char* qual;                                // '@': the stripped quality line (dlen bytes)
    uint64_t dlen;
};

struct SeqIndex {                                    // what one thread made of its byte range of the file
    std::vector<RecRef> recs;
    const char* first = nullptr;                     // where its first record starts (nullptr: none starts inside the range)
    const char* end = nullptr;                       // where the record behind its last one starts (blank lines skipped)
    std::string err;
};

using KeepSet = std::unordered_map<std::string, char>;

inline const char* skip_blank_lines(const char* p, const char* e) {
    while (p < e) {
        const char* le = line_end(p, e);
        const char* a = p; const char* b = le;
        strip(a, b);
        if (a != b) break;
        p = le < e ? le + 1 : e;
    }
    return p;
}

// pass 1: the records that START in [p, range_end), p being a record start (or blank lines before one); reads on to the end of the
// last one.  Only memchr and strip: no byte of sequence or quality is looked at.
void index_records(const char* p, const char* range_end, const char* e, const std::string& path, const KeepSet* keep, SeqIndex& o) {
    p = skip_blank_lines(p, e);
    while (p < range_end && p < e) {
        if (*p != '>' && *p != '@') { o.err = path + ": unrecognised record"; return; }
        const char* le = line_end(p, e);
        RecRef r;
        r.kind = *p;
        // name: the header line without its first byte, cut at the first whitespace (b"...".split()[0])
        const char* ns = p + 1; const char* ne = le;
        strip(ns, ne);
        const char* nc = ns;
        while (nc < ne && !is_space(*nc)) ++nc;
        r.name = ns; r.name_len = (uint32_t)(nc - ns);
        const bool want = !keep || keep->count(std::string(ns, (size_t)(nc - ns))) != 0;
        const char* q = le < e ? le + 1 : e;
        r.qual = nullptr; r.dlen = 0;
        if (r.kind == '>') {
            // sequence lines up to the next line that starts with '>' (or the end); every line stripped, blank lines contribute nothing
            r.seq = q;
            while (q < e && *q != '>') {
                const char* l2 = line_end(q, e);
                const char* a = q; const char* b = l2;
                strip(a, b);
                r.dlen += (uint64_t)(b - a);
                q = l2 < e ? l2 + 1 : e;
            }
            r.seq_end = q;
        } else {
            // four-line FASTQ: sequence, separator, quality
            const char* l2 = line_end(q, e);
            const char* a = q; const char* b = l2;
            strip(a, b);
            r.seq = a; r.seq_end = b; r.dlen = (uint64_t)(b - a);
            q = l2 < e ? l2 + 1 : e;
            q = line_end(q, e); q = q < e ? q + 1 : e;                       // the '+' line
            const char* l4 = line_end(q, e);
            const char* qa = q; const char* qb = l4;
            strip(qa, qb);
            if ((uint64_t)(qb - qa) != r.dlen) { o.err = path + ": quality length differs from sequence length for " + std::string(ns, (size_t)(nc - ns)); return; }
            r.qual = qa;
            q = l4 < e ? l4 + 1 : e;
        }
        if (want) o.recs.push_back(r);
        p = skip_blank_lines(q, e);
    }
    o.end = p;
}

// A position at or behind x where a record probably starts, for a thread that begins in the middle of the file.  FASTA: a line
// starting with '>' (exact: the parser itself ends a record there).  FASTQ: an '@' line whose line after next starts with '+' ('@'
// may open a quality line too, but then the line after next is a sequence line).  A guess only: the caller accepts the pieces only
// when every piece ends exactly where the next one starts, and reads the file in one piece otherwise.
const char* guess_record_start(const char* base, const char* x, const char* e, bool fastq) {
    const char* q = x;
    if (q > base && q[-1] != '\n') { q = line_end(q, e); q = q < e ? q + 1 : e; }
    while (q < e) {
        const char* le = line_end(q, e);
        if (!fastq) { if (*q == '>') return q; }
        else if (*q == '@') {
            const char* l1 = le < e ? le + 1 : e;
            const char* l2 = line_end(l1, e); l2 = l2 < e ? l2 + 1 : e;
            if (l2 < e && *l2 == '+') return q;
        }
        q = le < e ? le + 1 : e;
    }
    return e;
}

// sequence.cpp:19-42: the sum of (c - '!') decides whether a quality string counts; all '!' -> none
inline bool quality_counts(const char* qa, uint64_t n) {
    unsigned long long sum = 0; bool below = false;
    for (uint64_t i = 0; i < n; ++i) { sum += (unsigned long long)((unsigned char)qa[i] - 33u); below |= (unsigned char)qa[i] < 33u; }
    if (below) { long long s2 = 0; for (uint64_t i = 0; i < n; ++i) s2 += (long long)(unsigned char)qa[i] - 33; sum = (unsigned long long)(s2 != 0); }
    return n > 0 && sum != 0;
}
inline void copy_upper(char* o, const char* a, size_t n) {                            // sequence.cpp:19-42 upper-cases the bases
    for (size_t i = 0; i < n; ++i) { const char c = a[i]; o[i] = (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
}

}  // namespace

extern "C" {

vc_seqset* vc_io_read_sequences(const char* path, const char* keep_names, int names_only) {
    vc_seqset* s = new vc_seqset();
    s->names_only = names_only != 0;
    if (!path) { s->err = "null path"; return s; }
    VC_IO_T0;
    Blob f;
    if (!load_file(path, f, s->err)) return s;
    VC_IO_TICK("map");
    KeepSet keep;
    if (keep_names) {
        for (const char* p = keep_names; *p;) {
            const char* e = strchr(p, '\n');
            if (!e) e = p + strlen(p);
            if (e > p) keep.emplace(std::string(p, (size_t)(e - p)), 1);
            p = *e ? e + 1 : e;
        }
    }
    const KeepSet* kp = keep_names ? &keep : nullptr;
    const std::string ps(path);
    const char* const base = f.p; const char* const fe = f.p + f.n;
    // ---- pass 1: where the records are.  Byte ranges side by side; a thread that starts mid-file guesses its first record start,
    // and the pieces count only if they chain (piece k ends exactly where piece k+1 begins) -- then they are what one thread
    // walking the whole file finds.  Otherwise (a file the guess does not fit, or an error anywhere) one thread walks it.
    const char* s0 = base;
    while (s0 < fe && is_space(*s0)) ++s0;
    const bool fastq = s0 < fe && *s0 == '@';
    unsigned parts = f.n < (1u << 22) ? 1u : n_threads();
    std::vector<SeqIndex> idx(parts);
    if (parts > 1) {
        const size_t step = f.n / parts;
        std::vector<std::thread> th;
        for (unsigned k = 0; k < parts; ++k)
            th.emplace_back([&, k]() {
                const char* lo = base + (size_t)k * step; const char* hi = k + 1 == parts ? fe : base + (size_t)(k + 1) * step;
                f.populate(lo, hi);
                const char* st = k == 0 ? base : guess_record_start(base, lo, fe, fastq);
                idx[k].first = st;
                if (st < hi || k == 0) index_records(st, hi, fe, ps, kp, idx[k]); else idx[k].end = st;
            });
        for (auto& t : th) t.join();
        bool chained = true;
        for (unsigned k = 0; k < parts && chained; ++k) {
            if (!idx[k].err.empty()) chained = false;
            else if (k + 1 < parts && idx[k].end != idx[k + 1].first) {
                // a piece in which no record starts hands its neighbour's start on; anything else is a seam that does not fit
                chained = false;
            }
        }
        if (!chained) { parts = 1; idx.assign(1, SeqIndex()); }
    }
    if (parts == 1) index_records(base, fe, fe, ps, kp, idx[0]);
    if (io_tm_) fprintf(stderr, "  [vc_io] %u piece(s)\n", parts);
    VC_IO_TICK("index");
    for (auto& p : idx) if (!p.err.empty()) { s->err = p.err; return s; }
    size_t nr = 0;
    for (auto& p : idx) nr += p.recs.size();
    std::vector<RecRef> recs;
    recs.reserve(nr);
    for (auto& p : idx) { recs.insert(recs.end(), p.recs.begin(), p.recs.end()); std::vector<RecRef>().swap(p.recs); }
    // ---- offsets
    s->name_off.assign(nr + 1, 0); s->data_off.assign(nr + 1, 0); s->has_qual.assign(nr, 0); s->length.assign(nr, 0);
    std::vector<uint64_t> at(nr + 1, 0);                                   // data offsets as if the data were kept (work split of pass 2)
    bool any_fastq = false;
    for (size_t i = 0; i < nr; ++i) {
        s->name_off[i + 1] = s->name_off[i] + recs[i].name_len;
        at[i + 1] = at[i] + recs[i].dlen;
        s->length[i] = recs[i].dlen;
        any_fastq |= recs[i].kind == '@' && recs[i].dlen > 0;
    }
    const uint64_t nd = at[nr];
    if (!names_only) { s->data_off = at; s->data.alloc(nd); if (any_fastq) s->qual.alloc(nd); }
    s->names.resize(s->name_off[nr]);
    VC_IO_TICK("alloc");
    // ---- pass 2: the bytes, records side by side (pieces of about equal size), every thread straight into the final buffers
    const unsigned T = (unsigned)std::min<size_t>(nd + nr < (1u << 20) ? 1u : n_threads(), std::max<size_t>(nr, 1));
    std::vector<uint8_t> anyq(T, 0);
    auto fill = [&](unsigned t) {
        size_t i0 = (size_t)(std::lower_bound(at.begin(), at.end(), nd / T * t) - at.begin()), i1 = (size_t)(std::lower_bound(at.begin(), at.end(), nd / T * (t + 1)) - at.begin());
        if (t == 0) i0 = 0;
        if (t + 1 == T) i1 = nr;
        i0 = std::min(i0, nr); i1 = std::min(i1, nr);
        for (size_t i = i0; i < i1; ++i) {
            const RecRef& r = recs[i];
            if (r.name_len) std::memcpy(&s->names[s->name_off[i]], r.name, r.name_len);
            char* d = names_only ? nullptr : s->data.data() + at[i];
            if (r.kind == '@') {
                if (d) copy_upper(d, r.seq, (size_t)r.dlen);
                const bool hq = quality_counts(r.qual, r.dlen);
                s->has_qual[i] = hq ? 1 : 0;
                if (hq) anyq[t] = 1;
                if (d && any_fastq) { if (hq) std::memcpy(s->qual.data() + at[i], r.qual, (size_t)r.dlen); else std::memset(s->qual.data() + at[i], '!', (size_t)r.dlen); }
            } else {
                if (d) {
                    for (const char* q = r.seq; q < r.seq_end;) {
                        const char* l2 = line_end(q, r.seq_end);
                        const char* a = q; const char* b = l2;
                        strip(a, b);
                        copy_upper(d, a, (size_t)(b - a)); d += b - a;
                        q = l2 < r.seq_end ? l2 + 1 : r.seq_end;
                    }
                    if (any_fastq) std::memset(s->qual.data() + at[i], '!', (size_t)r.dlen);
                }
            }
        }
    };
    if (T <= 1) fill(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fill, t);
        for (auto& t : th) t.join();
    }
    bool some = false;
    for (uint8_t a : anyq) some |= a != 0;
    if (!some) s->qual.alloc(0);                             // no record with a quality that counts: no quality buffer at all
    VC_IO_TICK("copy");
    return s;
}

void vc_seqset_free(vc_seqset* s) { delete s; }
const char* vc_seqset_error(const vc_seqset* s) { return s && !s->err.empty() ? s->err.c_str() : nullptr; }
uint64_t vc_seqset_size(const vc_seqset* s) { return s ? s->size() : 0; }
const uint64_t* vc_seqset_name_off(const vc_seqset* s) { return s->name_off.data(); }
const char* vc_seqset_names(const vc_seqset* s) { return s->names.data(); }
const uint64_t* vc_seqset_data_off(const vc_seqset* s) { return s->data_off.data(); }
const char* vc_seqset_data(const vc_seqset* s) { return s->data.data(); }
const char* vc_seqset_qual(const vc_seqset* s) { return s->qual.empty() ? nullptr : s->qual.data(); }
const uint8_t* vc_seqset_has_qual(const vc_seqset* s) { return s->has_qual.data(); }
const uint64_t* vc_seqset_lengths(const vc_seqset* s) { return s->length.data(); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------ overlaps
struct vc_ovlset {
    // names as (offset, length) into `text`; MHAP records carry file positions instead (by_index)
    std::string text;
    std::vector<uint64_t> qn_off, tn_off; std::vector<uint32_t> qn_len, tn_len;
    std::vector<uint8_t> by_index, strand, has_cigar, dropped;
    std::vector<uint32_t> q_begin, q_end, q_length, t_begin, t_end, length, q_index, t_index;
    std::vector<double> error;
    std::vector<std::string> cigar;                  // per record (SAM / PAF cg:Z: / set by vc_ovlset_set_cigar)
    std::string err;
    size_t size() const { return strand.size(); }
    void push_common(bool st, uint32_t qb, uint32_t qe, uint32_t ql, uint32_t tb, uint32_t te, uint32_t len, double er) {
        strand.push_back(st); q_begin.push_back(qb); q_end.push_back(qe); q_length.push_back(ql); t_begin.push_back(tb); t_end.push_back(te);
        length.push_back(len); error.push_back(er); dropped.push_back(0);
    }
};

namespace {

struct Fields {                                      // tab- (or whitespace-) separated fields of a line
    std::vector<std::pair<const char*, const char*>> f;
    void tabs(const char* p, const char* e) {
        f.clear();
        for (;;) {
            const char* t = (const char*)memchr(p, '\t', (size_t)(e - p));
            f.emplace_back(p, t ? t : e);
            if (!t) break;
            p = t + 1;
        }
    }
    void blanks(const char* p, const char* e) {
        f.clear();
        while (p < e) {
            while (p < e && is_space(*p)) ++p;
            if (p == e) break;
            const char* s = p;
            while (p < e && !is_space(*p)) ++p;
            f.emplace_back(s, p);
        }
    }
};

bool to_u64(const char* p, const char* e, uint64_t& v) {
    if (p == e) return false;
    v = 0;
    for (; p < e; ++p) { if (*p < '0' || *p > '9') return false; v = v * 10 + (uint64_t)(*p - '0'); }
    return true;
}

// src/overlap.cpp:44-110 on a CIGAR string: every <count><operation> token counts, anything else is passed over (the Python
// restatement's `(\d+)([MIDNSHP=X])` scan); the leading clip is the FIRST token when that one is S or H
void sam_spans(const char* c, const char* e, uint64_t& q_begin_clip, uint64_t& q_aln, uint64_t& t_aln, uint64_t& clip) {
    q_begin_clip = q_aln = t_aln = clip = 0;
    bool first = true;
    while (c < e) {
        if (*c < '0' || *c > '9') { ++c; continue; }
        uint64_t n = 0;
        while (c < e && *c >= '0' && *c <= '9') { n = n * 10 + (uint64_t)(*c - '0'); ++c; }
        if (c == e) break;
        switch (*c) {
            case 'M': case '=': case 'X': q_aln += n; t_aln += n; break;
            case 'I': q_aln += n; break;
            case 'D': case 'N': t_aln += n; break;
            case 'S': case 'H': clip += n; if (first) q_begin_clip = n; break;
            case 'P': break;
            default: continue;                       // not an operation: the digits before it were no token
        }
        first = false;
        ++c;
    }
}

enum OvlFmt { MHAP, PAF, SAM };

// the records of the lines in [p, e) of `base` (names as offsets into base); first_line: number of the first line, for messages
void parse_overlap_lines(vc_ovlset* o, OvlFmt fmt, const std::string& ps, const char* base, const char* p, const char* e) {
    Fields fl;
    auto name = [&](std::vector<uint64_t>& off, std::vector<uint32_t>& len, const std::pair<const char*, const char*>& x) {
        off.push_back((uint64_t)(x.first - base)); len.push_back((uint32_t)(x.second - x.first));
    };
    // coordinates out of order or beyond 32 bits (the fields are u32 from here on): refused here, with the place, instead of wrapping
    // around and surfacing much later as a generic range error (seqio.py: _check_spans is the same test)
    auto bad_spans = [](uint64_t qb, uint64_t qe, uint64_t ql, uint64_t tb, uint64_t te) {
        return qb > qe || tb > te || std::max(std::max(qe, ql), te) > 0xFFFFFFFFull;
    };
    while (p < e) {
        const char* le = line_end(p, e);
        const char* a = p; const char* b = le;
        p = le < e ? le + 1 : e;
        const char* sa = a; const char* sb = b;
        strip(sa, sb);
        if (sa == sb) continue;
        const std::string where = " in the record that starts at byte " + std::to_string((size_t)(a - base));
        const std::string spans = ": coordinates out of order or beyond 32 bits";
        if (fmt == SAM) {
            if (*a == '@') continue;
            if (b > a && b[-1] == '\r') --b;
            fl.tabs(a, b);
            uint64_t flag = 0, pos = 0;
            if (fl.f.size() < 6 || !to_u64(fl.f[1].first, fl.f[1].second, flag) || !to_u64(fl.f[3].first, fl.f[3].second, pos)) {
                o->err = ps + ": malformed SAM record" + where; return;
            }
            if (flag & 0x4) continue;                                          // unmapped
            const char* cs = fl.f[5].first; const char* ce = fl.f[5].second;
            if (ce - cs < 2) { o->err = "missing alignment from SAM object"; return; }
            uint64_t qbc, qa, ta, clip;
            sam_spans(cs, ce, qbc, qa, ta, clip);
            const bool st = (flag & 0x10) != 0;
            uint64_t qb = qbc, qe = qbc + qa;
            const uint64_t ql = clip + qa;
            if (st) { const uint64_t nb = ql - qe, ne = ql - qb; qb = nb; qe = ne; }
            if (pos == 0 || bad_spans(qb, qe, ql, pos - 1, pos - 1 + ta)) { o->err = ps + ": malformed SAM record" + spans + where; return; }
            const uint64_t tb = pos - 1, te = tb + ta, len = std::max(qa, ta);
            name(o->qn_off, o->qn_len, fl.f[0]); name(o->tn_off, o->tn_len, fl.f[2]);
            o->by_index.push_back(0); o->q_index.push_back(0); o->t_index.push_back(0);
            o->push_common(st, (uint32_t)qb, (uint32_t)qe, (uint32_t)ql, (uint32_t)tb, (uint32_t)te, (uint32_t)len,
                           len ? 1 - (double)std::min(qa, ta) / (double)len : 1.0);
            o->has_cigar.push_back(1); o->cigar.emplace_back(cs, (size_t)(ce - cs));
        } else if (fmt == PAF) {
            if (b > a && b[-1] == '\r') --b;
            fl.tabs(a, b);
            uint64_t ql, qb, qe, tb, te;
            if (fl.f.size() < 9 || !to_u64(fl.f[1].first, fl.f[1].second, ql) || !to_u64(fl.f[2].first, fl.f[2].second, qb) ||
                !to_u64(fl.f[3].first, fl.f[3].second, qe) || !to_u64(fl.f[7].first, fl.f[7].second, tb) || !to_u64(fl.f[8].first, fl.f[8].second, te)) {
                o->err = ps + ": malformed PAF record" + where; return;
            }
            if (bad_spans(qb, qe, ql, tb, te)) { o->err = ps + ": malformed PAF record" + spans + where; return; }
            const uint64_t len = std::max(qe - qb, te - tb);
            name(o->qn_off, o->qn_len, fl.f[0]); name(o->tn_off, o->tn_len, fl.f[5]);
            o->by_index.push_back(0); o->q_index.push_back(0); o->t_index.push_back(0);
            o->push_common(fl.f[4].second - fl.f[4].first == 1 && *fl.f[4].first == '-', (uint32_t)qb, (uint32_t)qe, (uint32_t)ql, (uint32_t)tb, (uint32_t)te,
                           (uint32_t)len, len ? 1 - (double)std::min(qe - qb, te - tb) / (double)len : 1.0);
            bool got = false;
            for (size_t k = 12; k < fl.f.size() && !got; ++k)
                if (fl.f[k].second - fl.f[k].first >= 5 && memcmp(fl.f[k].first, "cg:Z:", 5) == 0) { o->cigar.emplace_back(fl.f[k].first + 5, (size_t)(fl.f[k].second - fl.f[k].first - 5)); got = true; }
            if (!got) o->cigar.emplace_back();
            o->has_cigar.push_back(got ? 1 : 0);
        } else {
            fl.blanks(a, b);
            uint64_t v[12];
            bool ok = fl.f.size() >= 12;
            for (int k : {0, 1, 4, 5, 6, 7, 8, 9, 10}) ok = ok && to_u64(fl.f[k].first, fl.f[k].second, v[k]);
            if (!ok) { o->err = ps + ": malformed MHAP record" + where; return; }
            if (v[0] < 1 || v[1] < 1 || v[0] > 0xFFFFFFFFull || v[1] > 0xFFFFFFFFull || bad_spans(v[5], v[6], v[7], v[9], v[10])) { o->err = ps + ": malformed MHAP record" + spans + where; return; }
            const uint64_t len = std::max(v[6] - v[5], v[10] - v[9]);
            o->qn_off.push_back(0); o->qn_len.push_back(0); o->tn_off.push_back(0); o->tn_len.push_back(0);
            o->by_index.push_back(1); o->q_index.push_back((uint32_t)(v[0] - 1)); o->t_index.push_back((uint32_t)(v[1] - 1));
            o->push_common((v[4] ^ v[8]) != 0, (uint32_t)v[5], (uint32_t)v[6], (uint32_t)v[7], (uint32_t)v[9], (uint32_t)v[10], (uint32_t)len,
                           len ? 1 - (double)std::min(v[6] - v[5], v[10] - v[9]) / (double)len : 1.0);
            o->has_cigar.push_back(0); o->cigar.emplace_back();
        }
    }
}

template <typename V> void append(V& a, V& b) { a.insert(a.end(), std::make_move_iterator(b.begin()), std::make_move_iterator(b.end())); }

}  // namespace

extern "C" {

vc_ovlset* vc_io_read_overlaps(const char* path) {
    vc_ovlset* o = new vc_ovlset();
    if (!path) { o->err = "null path"; return o; }
    const std::string ps(path);
    OvlFmt fmt;
    if (ends_with(ps, ".mhap") || ends_with(ps, ".mhap.gz")) fmt = MHAP;
    else if (ends_with(ps, ".sam") || ends_with(ps, ".sam.gz")) fmt = SAM;
    else if (ends_with(ps, ".paf") || ends_with(ps, ".paf.gz")) fmt = PAF;
    else { o->err = ps + ": unsupported overlap format (valid extensions: .mhap, .mhap.gz, .paf, .paf.gz, .sam, .sam.gz)"; return o; }
    {
        Blob f;
        if (!load_file(path, f, o->err)) return o;
        o->text.assign(f.p, f.n);                    // names point into this copy (the mapping goes away with this call)
    }
    const char* base = o->text.data();
    const size_t n = o->text.size();
    // pieces at line boundaries, one per thread; the records of the pieces in file order
    std::vector<size_t> cut{0};
    const unsigned T = n < (1u << 22) ? 1u : n_threads();
    for (unsigned k = 1; k < T; ++k) {
        const char* q = (const char*)memchr(base + n * k / T, '\n', n - n * k / T);
        const size_t at = q ? (size_t)(q - base) + 1 : n;
        if (at > cut.back() && at < n) cut.push_back(at);
    }
    cut.push_back(n);
    std::vector<vc_ovlset> part(cut.size() - 1);
    std::vector<std::thread> th;
    for (size_t k = 0; k + 1 < cut.size(); ++k)
        th.emplace_back([&, k]() { parse_overlap_lines(&part[k], fmt, ps, base, base + cut[k], base + cut[k + 1]); });
    for (auto& t : th) t.join();
    for (auto& p : part) {
        if (!p.err.empty()) { if (o->err.empty()) o->err = p.err; continue; }
        append(o->qn_off, p.qn_off); append(o->tn_off, p.tn_off); append(o->qn_len, p.qn_len); append(o->tn_len, p.tn_len);
        append(o->by_index, p.by_index); append(o->strand, p.strand); append(o->has_cigar, p.has_cigar); append(o->dropped, p.dropped);
        append(o->q_begin, p.q_begin); append(o->q_end, p.q_end); append(o->q_length, p.q_length); append(o->t_begin, p.t_begin); append(o->t_end, p.t_end);
        append(o->length, p.length); append(o->q_index, p.q_index); append(o->t_index, p.t_index); append(o->error, p.error); append(o->cigar, p.cigar);
    }
    return o;
}

void vc_ovlset_free(vc_ovlset* o) { delete o; }
const char* vc_ovlset_error(const vc_ovlset* o) { return o && !o->err.empty() ? o->err.c_str() : nullptr; }
uint64_t vc_ovlset_size(const vc_ovlset* o) { return o ? o->size() : 0; }

int vc_ovlset_get(const vc_ovlset* o, uint64_t i, vc_overlap_rec* r) {
    if (!o || !r || i >= o->size()) return VC_ERR_ARG;
    r->q_name = o->text.data() + o->qn_off[i]; r->q_name_len = o->qn_len[i];
    r->t_name = o->text.data() + o->tn_off[i]; r->t_name_len = o->tn_len[i];
    r->by_index = o->by_index[i]; r->q_index = o->q_index[i]; r->t_index = o->t_index[i];
    r->strand = o->strand[i]; r->q_begin = o->q_begin[i]; r->q_end = o->q_end[i]; r->q_length = o->q_length[i];
    r->t_begin = o->t_begin[i]; r->t_end = o->t_end[i]; r->length = o->length[i]; r->error = o->error[i];
    r->cigar = o->has_cigar[i] ? o->cigar[i].c_str() : nullptr;
    r->dropped = o->dropped[i];
    return VC_OK;
}

// give record i its CIGAR (the device aligner's result); NULL: the record cannot be aligned, drop it
int vc_ovlset_set_cigar(vc_ovlset* o, uint64_t i, const char* cigar) {
    if (!o || i >= o->size()) return VC_ERR_ARG;
    if (cigar) { o->cigar[i] = cigar; o->has_cigar[i] = 1; }
    else { o->cigar[i].clear(); o->has_cigar[i] = 1; o->dropped[i] = 1; }
    return VC_OK;
}

// Polisher::initialize (src/polisher.cpp:207-352) in fragment-correction mode: sequences and overlaps into the window builder.
// Returns the number of overlaps kept, or -1 (message through err).  window_type: 0 NGS (mean read length <= 1000), 1 TGS.
// ---- planning for one rank of a multi-GPU run (vechat_amd/polish.py): nothing but names, lengths and the overlap records are in
// memory yet.  Stands in for the part of Polisher::initialize that sizes the work (src/polisher.cpp:207-352) before anything large
// is loaded; the reference has no ranks -- it fans windows out over its devices inside one process (src/cuda/cudapolisher.cpp:229-241).
// cost[k] = length of target k + the target bases every overlap lays on it (windows x depth x length).
// Overlaps that name their sequences by file position (MHAP) are resolved through `reads` / `targets` (both may be names-only sets).
int vc_io_target_cost(const vc_ovlset* o, const vc_seqset* targets, double* cost) {
    if (!o || !targets || !cost) return VC_ERR_ARG;
    std::unordered_map<std::string, uint32_t> t_id;
    t_id.reserve(targets->size() * 2);
    for (size_t i = 0; i < targets->size(); ++i) { t_id[targets->name(i)] = (uint32_t)i; cost[i] = (double)targets->length[i]; }
    for (size_t k = 0; k < o->size(); ++k) {
        uint32_t t;
        if (o->by_index[k]) { if (o->t_index[k] >= targets->size()) continue; t = t_id[targets->name(o->t_index[k])]; }
        else {
            auto ti = t_id.find(std::string(o->text.data() + o->tn_off[k], o->tn_len[k]));
            if (ti == t_id.end()) continue;
            t = ti->second;
        }
        cost[t] += (double)(o->t_end[k] - o->t_begin[k]);
    }
    return VC_OK;
}

// '\n'-separated names a rank that owns targets [t_lo, t_hi) has to load: those targets and the query of every overlap on one of
// them (each once).  The buffer is malloc'ed: vc_io_free() it.  NULL on error.
char* vc_io_rank_names(const vc_ovlset* o, const vc_seqset* targets, const vc_seqset* reads, uint64_t t_lo, uint64_t t_hi, uint64_t* n_names) {
    if (!o || !targets || t_lo > t_hi || t_hi > targets->size()) return nullptr;
    std::unordered_map<std::string, uint8_t> mine;
    mine.reserve((t_hi - t_lo) * 2);
    std::string out;
    uint64_t n = 0;
    for (uint64_t i = t_lo; i < t_hi; ++i) {
        const std::string nm = targets->name(i);
        if (mine.emplace(nm, 1).second) { out += nm; out += '\n'; n++; }
    }
    std::unordered_map<std::string, uint8_t> seen;
    for (size_t k = 0; k < o->size(); ++k) {
        std::string tn, qn;
        if (o->by_index[k]) {
            if (!reads || o->t_index[k] >= targets->size() || o->q_index[k] >= reads->size()) continue;
            tn = targets->name(o->t_index[k]); qn = reads->name(o->q_index[k]);
        } else {
            tn.assign(o->text.data() + o->tn_off[k], o->tn_len[k]); qn.assign(o->text.data() + o->qn_off[k], o->qn_len[k]);
        }
        if (!mine.count(tn) || mine.count(qn)) continue;
        if (seen.emplace(qn, 1).second) { out += qn; out += '\n'; n++; }
    }
    char* buf = (char*)malloc(out.size() + 1);
    if (!buf) return nullptr;
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    if (n_names) *n_names = n;
    return buf;
}
void vc_io_free(void* p) { free(p); }

int64_t vc_io_load(vc_wb* wb, const vc_seqset* targets, const vc_seqset* reads, vc_ovlset* ovl, double error_threshold, int allow_empty,
                   int* window_type, char* err, uint64_t err_cap) {
    auto fail = [&](const std::string& m) -> int64_t { if (err && err_cap) { snprintf(err, (size_t)err_cap, "%s", m.c_str()); } return -1; };
    if (!wb || !targets || !reads || !ovl) return fail("null argument");
    VC_IO_T0;
    if (targets->size() == 0) return fail("empty target sequences set");
    if (reads->size() == 0 && !allow_empty) return fail("empty sequences set");
    auto add = [&](const vc_seqset* s, size_t i) -> int {           // no copy: the record buffers outlive the builder (they are the caller's sets)
        const uint64_t o0 = s->data_off[i], len = s->data_off[i + 1] - o0;
        return vc_wb_add_sequence_view(wb, s->names.data() + s->name_off[i], (uint32_t)(s->name_off[i + 1] - s->name_off[i]), s->data.data() + o0,
                                       (uint32_t)len, s->has_qual[i] ? s->qual.data() + o0 : nullptr);
    };
    std::unordered_map<std::string, uint32_t> t_id, q_id;
    t_id.reserve(targets->size() * 2); q_id.reserve(reads->size() * 2);
    std::unordered_map<uint32_t, size_t> t_rec;      // builder id of a target -> its record
    for (size_t i = 0; i < targets->size(); ++i) {
        const int id = add(targets, i);
        if (id < 0) return fail("empty sequence");
        t_id[targets->name(i)] = (uint32_t)id;       // (a repeated name: the last record wins, as a dict does)
        t_rec[(uint32_t)id] = i;
    }
    unsigned long long total = 0;
    for (size_t i = 0; i < reads->size(); ++i) {
        const uint64_t len = reads->data_off[i + 1] - reads->data_off[i];
        total += len;
        const std::string nm = reads->name(i);
        auto it = t_id.find(nm);
        if (it != t_id.end()) {                      // a read that is also a target shares its record (polisher.cpp:262-283)
            const size_t ti = t_rec[it->second];
            const uint64_t tlen = targets->data_off[ti + 1] - targets->data_off[ti];
            const uint64_t tq = targets->has_qual[ti] ? tlen : 0, rq = reads->has_qual[i] ? len : 0;
            if (tlen != len || tq != rq) return fail("duplicate sequence " + nm + " with unequal data");
            q_id[nm] = it->second;
        } else {
            const int id = add(reads, i);
            if (id < 0) return fail("empty sequence");
            q_id[nm] = (uint32_t)id;
        }
    }
    if (vc_wb_set_targets(wb, (uint32_t)targets->size()) != VC_OK) return fail("vc_wb_set_targets failed");
    VC_IO_TICK("sequences");
    std::vector<uint32_t> oq, ot, oqb, oqe, oql, otb, ote;
    std::vector<uint8_t> ost;
    std::vector<const char*> ocg;
    for (size_t k = 0; k < ovl->size(); ++k) {
        uint32_t q, t;
        if (ovl->by_index[k]) {                      // MHAP: positions in the reads / targets files (overlap.cpp:129-166)
            if (ovl->q_index[k] >= reads->size() || ovl->t_index[k] >= targets->size()) continue;
            // ... resolved through the NAMES, like the other formats (a read that is also a target has one id)
            auto qi = q_id.find(reads->name(ovl->q_index[k])); auto ti = t_id.find(targets->name(ovl->t_index[k]));
            if (qi == q_id.end() || ti == t_id.end()) continue;
            q = qi->second; t = ti->second;
        } else {
            auto qi = q_id.find(std::string(ovl->text.data() + ovl->qn_off[k], ovl->qn_len[k]));
            auto ti = t_id.find(std::string(ovl->text.data() + ovl->tn_off[k], ovl->tn_len[k]));
            if (qi == q_id.end() || ti == t_id.end()) continue;
            q = qi->second; t = ti->second;
        }
        if (ovl->dropped[k] || ovl->error[k] > error_threshold || q == t) continue;
        if (!ovl->has_cigar[k]) return fail("overlap without a CIGAR: align it first (vc_align / vc_ovlset_set_cigar)");
        oq.push_back(q); ot.push_back(t); ost.push_back(ovl->strand[k]); oqb.push_back(ovl->q_begin[k]); oqe.push_back(ovl->q_end[k]);
        oql.push_back(ovl->q_length[k]); otb.push_back(ovl->t_begin[k]); ote.push_back(ovl->t_end[k]); ocg.push_back(ovl->cigar[k].c_str());
    }
    VC_IO_TICK("resolve");
    const int64_t kept = (int64_t)oq.size();
    if (kept && vc_wb_add_overlaps(wb, (uint64_t)kept, oq.data(), ot.data(), ost.data(), oqb.data(), oqe.data(), oql.data(), otb.data(), ote.data(),
                                   ocg.data()) != VC_OK)
        return fail(vc_wb_last_error(wb));
    VC_IO_TICK("overlaps");
    if (kept == 0 && !allow_empty) return fail("empty overlap set");
    if (window_type) *window_type = (double)total / (double)std::max<size_t>(reads->size(), 1) <= 1000 ? 0 : 1;
    return kept;
}

}  // extern "C"
