/* libvechat_hip.so -- C ABI of the MI355X-native replacement for VeChat's per-window hot path
 * (SPOA partial-order alignment + graph prune + consensus).
 *
 * Each entry point replaces a piece of the reference's C++ interface for this path; the
 * reference file:line it stands in for is cited next to it (paths relative to the reference
 * tree).  No C++ or torch types cross this boundary: plain pointers and sizes only.
 *
 * Threading and streams.  A vc_ctx is bound to one HIP device and must be driven by ONE host thread at a
 * time (the reference gives each worker its own spoa engine, src/polisher.cpp:186-190; its GPU shim gives
 * each batch processor its own stream, src/cuda/cudabatch.cpp:54).  Different contexts may be driven from
 * different threads at the same time.  What a context owns and what contexts share:
 *   - its own stream (vc_stream): H2D of vc_submit, D2H of vc_collect, fills;
 *   - the CHUNK STREAMS the kernels run on belong to the process: one set per device (up to 16, made on
 *     first use, never destroyed), used by every context of that device.  Two contexts that run at the
 *     same time interleave their chunks on those streams -- correct, results unchanged, but they share
 *     the device; nothing is gained over one context with batches queued behind each other;
 *   - host threads: one per chunk stream and context, started by the first vc_run, kept until vc_destroy
 *     (blocked on a condition variable while the context has nothing queued).
 * Pipelining inside one context.  A context holds TWO batches.  vc_run only queues the batch staged last
 * and returns; a vc_submit that follows copies the next batch in while that one runs; vc_collect hands out
 * the oldest run nobody has collected and waits for that run only.  So the loop of the reference's
 * accelerated polisher (fill the next batch while one computes, src/cuda/cudapolisher.cpp:246-277) is
 *     submit(b0) run   submit(b1) run   collect -> b0   submit(b2) run   collect -> b1   ...
 * on one thread, with H2D, kernels and D2H overlapping and no gap on the device between batches.  The
 * serial order (submit, run, [sync,] collect, submit ...) works as before and uses one batch slot.
 */
#ifndef VECHAT_HIP_H_
#define VECHAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vc_ctx vc_ctx;

/* status codes returned by every call */
enum {
    VC_OK            = 0,
    VC_ERR_ARG       = -1,   /* bad argument / malformed batch                                  */
    VC_ERR_HIP       = -2,   /* a HIP runtime call failed (vc_last_error has the text)          */
    VC_ERR_NO_DEVICE = -3,   /* no gfx950 device visible: the product path has NO CPU fallback  */
    VC_ERR_STATE     = -4,   /* call out of order (e.g. vc_run before vc_submit)                */
    VC_ERR_CAPACITY  = -5    /* caller buffer too small                                         */
};

/* per-window status written by vc_run (vc_result.status) */
enum {
    VC_WIN_OK          = 0,  /* consensus produced; generate_consensus() would return true      */
    VC_WIN_UNPOLISHED  = 1,  /* < 3 sequences: backbone copied, returns false (window.cpp:188-192) */
    VC_WIN_OVERFLOW    = 2,  /* graph outgrew max_nodes/max_edges/stack: resubmit with larger caps */
    VC_WIN_UNSUPPORTED = 3,  /* reserved: no longer produced (every valid window is computed on the device) */
    VC_WIN_INVALID     = 4   /* input the reference would throw on (graph.cpp:191-231)          */
};

/* Replaces the constructor arguments the reference threads from main.cpp:46-61 through
 * createPolisher (src/polisher.cpp:59,140-157) into spoa::AlignmentEngine::Create
 * (polisher.cpp:186-190) and Window::generate_consensus (polisher.cpp:501-515). */
typedef struct vc_params {
    int32_t  device;                        /* HIP device ordinal                                  */
    int32_t  match, mismatch, gap;          /* -m -x -g (3,-5,-4)                                  */
    int32_t  sw_match, sw_mismatch, sw_gap; /* local engine hard-coded at window.cpp:326 (3,-5,-4) */
    double   min_confidence, min_support;   /* -d -s                                               */
    uint32_t num_prune;                     /* -k                                                  */
    int32_t  mode;                          /* 0 = haplotype overload (window.cpp:176)             */
    int32_t  trim;                          /* ignored in mode 0 (window.cpp:399)                  */
    int32_t  window_type;                   /* 0 kNGS / 1 kTGS (window.hpp:21-24)                  */
    uint32_t max_nodes;                     /* per-window graph capacity; 0 = derive from batch    */
    uint32_t max_edges;                     /* 0 = derive                                          */
    uint32_t chunk_windows;                 /* windows resident per pass; 0 = derive from memory   */
    uint64_t scratch_bytes;                 /* device scratch budget; 0 = 60 % of free memory, at most 128 GiB (VC_SCRATCH_CAP_GB) */
    int32_t  profile;                       /* 1 = bracket every kernel launch with HIP events, 2 = only the forward kernel's */
    uint32_t n_streams;                     /* chunks in flight on separate chunk streams (<= 16); 0 = chosen per batch: 8 from 12 288 windows up, else 4 */
} vc_params;

/* A batch of windows, the unit the reference's accelerated path fills with
 * CUDABatchProcessor::addWindow (src/cuda/cudabatch.hpp:39-59, cudabatch.cpp:79-135).
 * Sequence 0 of every window is the backbone (with its quality or the dummy '!' string,
 * polisher.cpp:397-400); sequences 1.. are the layers in the reference's `rank` order
 * (window.cpp:203-210) -- use vc_rank_layers() on the host to obtain it. */
typedef struct vc_batch {
    uint32_t        n_windows;
    const uint32_t* win_seq_off;   /* [n_windows+1]                                            */
    const uint64_t* seq_off;       /* [n_seqs+1] byte offsets into bases/quals                 */
    const uint32_t* seq_begin;     /* [n_seqs] positions_.first  (backbone: 0)                 */
    const uint32_t* seq_end;       /* [n_seqs] positions_.second (backbone: 0)                 */
    const uint8_t*  seq_has_qual;  /* [n_seqs] 0 = qualities_[i].first == nullptr              */
    const uint8_t*  bases;
    const uint8_t*  quals;         /* same offsets as bases; don't-care where has_qual == 0    */
    const uint8_t*  win_fasta;     /* [n_windows] window.cpp:223's `if_fasta` (vc_backbone_is_fasta) */
} vc_batch;

/* Result of a batch: Window::consensus() (window.hpp:43-45) of every window, concatenated, plus
 * the bool generate_consensus() returns (status VC_WIN_OK <=> true, VC_WIN_UNPOLISHED <=> false). */
typedef struct vc_result {
    uint64_t* cons_off;    /* [n_windows+1] out                         */
    uint8_t*  cons;        /* out, capacity cons_cap bytes              */
    uint64_t  cons_cap;
    uint8_t*  status;      /* [n_windows] out                           */
} vc_result;

typedef struct vc_stats {
    uint64_t cells;          /* sum over Align calls of graph_nodes * sequence_len  */
    uint64_t alignments;
    uint64_t dp_rows;
    uint64_t far_row_reads;  /* predecessor rows older than the LDS ring, read back from the H matrix */
    uint64_t trace_steps;    /* backtrack moves emitted                                               */
    uint64_t trace_spec;     /* ... of which confirmed in bulk by the first-in-edge speculation       */
    uint64_t trace_rounds;   /* speculation rounds (each: one batch of loads)                         */
    uint32_t n_classes;      /* kernel classes below                                            */
    double   ms[16];         /* accumulated HIP-event time per kernel class (profile=1)         */
    uint64_t launches[16];
    char     names[16][24];
    uint32_t max_nodes, max_edges, chunk_windows, n_streams;   /* what the context actually used */
    double   busy_ms[16];    /* per class: time during which at least one launch of it was running (the chunk streams overlap,  */
                             /* so a class's launches overlap each other and `ms` counts such time once per launch)             */
    uint64_t band_redo;      /* alignments whose backtrack left the stored band and were run again with whole rows              */
    uint64_t device_bytes;   /* device memory this context holds (workspaces + batch buffers)                                   */
    uint64_t fwd_shader_cycles, fwd_wall_ticks;   /* summed over the forward waves' row loops: shader cycles and 100 MHz ticks -- their ratio  */
                             /* x 100 is the shader clock in MHz the chip sustained under the job                                    */
} vc_stats;

/* -- lifecycle: stands in for createCUDABatch / ~CUDABatchProcessor (cudabatch.hpp:26,33) ------ */
int  vc_create(vc_ctx** out, const vc_params* p);
void vc_destroy(vc_ctx* ctx);
const char* vc_last_error(const vc_ctx* ctx);   /* ctx may be NULL: error of a failed vc_create */

/* -- batch: addWindow()... / generateConsensus() / reset() (cudabatch.hpp:39-59) --------------- */
int vc_submit(vc_ctx* ctx, const vc_batch* b);        /* validates, copies to HBM (H2D), retains nothing of b; does not wait for a batch
                                                         that is running unless the workspaces must grow for this one                 */
int vc_run(vc_ctx* ctx);                              /* queues the whole hot path for the batch staged last; returns at once          */
int vc_sync(vc_ctx* ctx);                             /* waits for every queued run                                                    */
/* The three calls below speak of the OLDEST run whose results have not been collected (none such: the latest run) and wait
 * for that run only; vc_collect / vc_collect_device mark it collected. */
int vc_result_windows(vc_ctx* ctx, uint32_t* n_windows);  /* windows of that batch                                 */
int vc_result_size(vc_ctx* ctx, uint64_t* cons_bytes);/* total consensus bytes                                 */
int vc_collect(vc_ctx* ctx, vc_result* r);            /* D2H of consensus + status                             */
/* device-side hand-over for the multi-GPU gather (RCCL lives in the caller, e.g. torch.distributed):
 * compacts the consensus bytes into caller-owned DEVICE memory. */
int vc_collect_device(vc_ctx* ctx, void* d_cons, uint64_t cons_cap, void* d_cons_off /*u64[n+1]*/,
                      void* d_status /*u8[n]*/);
int vc_get_stats(vc_ctx* ctx, vc_stats* s);
/* diagnostics: per-window (site << 16) | detail of the kernel that took the window out of VC_WIN_OK */
int vc_debug_errinfo(vc_ctx* ctx, uint32_t* out /*[n_windows]*/);
/* Test hooks for localising a divergence (tests/golden/stages.json): stop vc_run after a stage (kind 1 build layer `index`
 * added, 2 prune `index` done, 3 AddWeights round `index` done, 0 run to the end) and digest a window's graph / last alignment
 * where it stands, in the record format of the oracle's vco_window_stages.  Single-chunk batches. */
int vc_debug_stop_after(vc_ctx* ctx, uint32_t kind, uint32_t index);
int vc_debug_stage_digest(vc_ctx* ctx, uint32_t window, int with_pairs, uint64_t* out /*[8], [0..1] untouched*/);
/* development: the row records (16 B each) the next alignment of window w will use, after a stopped run (tools/gpu_rowstats.py);
 * the persistent pipeline's counters / per-window words, and its waves' phase clocks (tools/gpu_pipe_dbg.py) */
int vc_debug_rows(vc_ctx* ctx, uint32_t window, uint32_t* out, uint32_t cap_rows, uint32_t* nrows);
int vc_debug_pipe_state(vc_ctx* ctx, uint32_t* out, uint32_t n);
int vc_debug_pipe_prof(vc_ctx* ctx, unsigned long long* out);
void* vc_stream(vc_ctx* ctx);                         /* the context's own hipStream_t (copies, fills); the kernels run on the process's chunk streams */
int   vc_set_profile(vc_ctx* ctx, int profile);       /* change vc_params.profile of a live context (0, 1, 2)  */
/* Execution plan of the build loop (src/window.cpp:239-298).  0 (default): lock-step -- one launch per kernel per layer for a
 * whole chunk.  1: persistent pipeline -- two resident kernels per chunk (forward + AddAlignment waves, backtrack waves) that
 * hand windows to each other through device-side queues, no launch and no lock-step per layer (vechat_amd/csrc/vc_pipe.h).
 * Results are identical either way; the environment variable VC_PIPE=0/1 sets the default of a new context.
 * forward_waves / backtrack_waves: resident workgroups of the two kernels (0: 15 and 5 per CU). */
int   vc_set_pipeline(vc_ctx* ctx, int on, uint32_t forward_waves, uint32_t backtrack_waves);
/* 1 when the library was built with -DVC_EXPERIMENTS (VC_EXPERIMENTS=1 python __graft_entry__.py): the persistent pipeline above is a
 * measured experiment (bit-identical, slower than the lock-step plan) and is left out of the default build -- vc_set_pipeline(ctx, 1, ..)
 * then returns VC_ERR_ARG, as do vc_debug_pipe_state / vc_debug_pipe_prof. */
int   vc_has_experiments(void);
/* Allocates the workspaces' memory now, in one piece (bytes = 0: the default budget, vc_params.scratch_bytes or 60 % of the free
 * memory up to 128 GiB), instead of under the first vc_submit; batches of any shape are then laid out inside it without further
 * allocations.  The counterpart of createCUDABatch sizing a batch's device memory at construction (mem_per_batch, src/cuda/cudapolisher.cpp:229-243):
 * a caller does it while its input is still being parsed.  Optional; without it vc_submit allocates what a batch needs. */
int   vc_reserve(vc_ctx* ctx, uint64_t bytes);
/* Changes the polishing parameters of a live context -- overload (mode), thresholds, prune rounds, trim, window type, scores: what differs
 * between the two rounds of the driver (scripts/vechat:59-93: round 1 `vechat_racon -f -p -d D -s S`, round 2 `-f [-u]`).  Device,
 * capacities, scratch budget and streams stay as created, workspaces stay where they are: one warm context serves both rounds and every
 * --split chunk, where a process per invocation (scripts/vechat:371-393) pays the start-up each time.  A batch staged before the call
 * must be submitted again. */
int   vc_set_polish_params(vc_ctx* ctx, const vc_params* params);
/* Gives the workspaces (and a reservation) back to the device; batch buffers and results stay.  The next vc_submit lays them out again.
 * For a caller that needs the memory for something else in between -- e.g. a second context for windows that overflowed. */
int   vc_release(vc_ctx* ctx);
/* The window type is known only after the reads are (Polisher::initialize, src/polisher.cpp:300-306); a context created before that
 * takes it here.  0 = kNGS, 1 = kTGS. */
int   vc_set_window_type(vc_ctx* ctx, int window_type);

/* -- host helpers that keep reference semantics on the host side of the boundary ---------------- */
/* rank[] as produced by window.cpp:203-210: rank[0]=0, rank[1..] = std::sort of 1..n-1 by begin
 * (same libstdc++ introsort => same permutation for ties). */
void vc_rank_layers(const uint32_t* begins, uint32_t n_seqs, uint32_t* rank_out);
/* window.cpp:223: `qualities_.front().first == std::string(len,'!')` -- a C-string comparison that
 * reads up to the NUL of the buffer `quality` points into. */
int  vc_backbone_is_fasta(const char* quality_cstr, uint32_t backbone_len);
/* graph.cpp:165-170 quality -> weight table as computed by this host's libm */
void vc_weight_lut(uint32_t lut[256]);

/* -- synthetic windows (SURVEY 8d generator; used by bench.py and the tests) -------------------- */
typedef struct vc_synth_cfg {
    uint64_t seed;
    uint32_t backbone_len;      /* L                                   */
    uint32_t n_layers;          /* D                                   */
    double   error_rate;        /* total per-base error                */
    double   frac_ins, frac_del, frac_sub;   /* split of the error      */
    double   frac_partial;      /* fraction of layers that are partial-span */
    int32_t  fastq;             /* 1 = layers carry qualities          */
    int32_t  backbone_fastq;    /* 1 = backbone has real quality, 0 = dummy '!' (FASTA target) */
    int32_t  n_haplotypes;      /* 1 or 2: reads drawn from this many variants of the truth    */
    double   snp_rate;          /* divergence between haplotypes       */
} vc_synth_cfg;

typedef struct vc_synth vc_synth;
/* generates windows [first, first+n) of the stream defined by cfg; layers are stored in rank order */
vc_synth* vc_synth_generate(const vc_synth_cfg* cfg, uint64_t first, uint32_t n, uint32_t n_threads);
void      vc_synth_batch(const vc_synth* s, vc_batch* out);   /* pointers stay owned by s */
uint64_t  vc_synth_n_seqs(const vc_synth* s);
/* [n_seqs] index each stored sequence had in add_layer() order (before the rank sort) */
const uint32_t* vc_synth_orig_index(const vc_synth* s);
uint64_t  vc_synth_n_bytes(const vc_synth* s);
void      vc_synth_free(vc_synth* s);

/* ------------------------------------------------------------------------------------------------
 * Window assembly and stitching (host only).  Stands in for the part of
 * Polisher::initialize that turns overlaps into windows (src/polisher.cpp:389-462, with the breaking
 * points of src/overlap.cpp:222-292 computed from a CIGAR string) and for the stitching loop of
 * Polisher::polish (src/polisher.cpp:520-547).  Sequences 0..n_targets-1 are the targets; overlaps are
 * taken in the order given (it decides the order of a window's layers before the rank sort).
 * ------------------------------------------------------------------------------------------------ */
typedef struct vc_wb vc_wb;
vc_wb*      vc_wb_create(uint32_t window_length, double quality_threshold);
void        vc_wb_destroy(vc_wb* b);
const char* vc_wb_last_error(const vc_wb* b);
/* returns the sequence id (>= 0) or -1; quality may be NULL (FASTA) */
int         vc_wb_add_sequence(vc_wb* b, const char* name, const char* data, uint32_t length, const char* quality);
/* the same without a copy: data / quality stay the caller's and must outlive the builder (vc_io_load passes its record buffers) */
int         vc_wb_add_sequence_view(vc_wb* b, const char* name, uint32_t name_len, const char* data, uint32_t length, const char* quality);
int         vc_wb_set_targets(vc_wb* b, uint32_t n_targets);
/* coordinates as in a PAF/SAM record: q_* on the read's forward strand, strand != 0 = reverse complement */
int         vc_wb_add_overlap(vc_wb* b, uint32_t q_id, uint32_t t_id, int strand, uint32_t q_begin, uint32_t q_end,
                              uint32_t q_length, uint32_t t_begin, uint32_t t_end, const char* cigar);
/* n overlaps at once, kept in the order given (their breaking points are computed on several threads) */
int         vc_wb_add_overlaps(vc_wb* b, uint64_t n, const uint32_t* q_id, const uint32_t* t_id, const uint8_t* strand, const uint32_t* q_begin,
                               const uint32_t* q_end, const uint32_t* q_length, const uint32_t* t_begin, const uint32_t* t_end,
                               const char* const* cigar);
uint32_t    vc_wb_n_breaking_points(const vc_wb* b, uint32_t overlap);
void        vc_wb_breaking_points(const vc_wb* b, uint32_t overlap, uint32_t* t_pos, uint32_t* q_pos);
/* fills `out` with arrays owned by the builder (valid until the next build / destroy): every window of every
 * target in order, layers in the reference's rank order */
int         vc_wb_build(vc_wb* b, vc_batch* out);
/* The same in two steps, for a caller that hands a large input to the device in slices (vc_submit of slice i + 1 while slice i runs): _begin
 * lays the whole batch out (every offset and buffer of `out` is final, the windows' bytes are not written yet), _fill writes the windows
 * [w_lo, w_hi).  A slice may be submitted once its windows are filled; filling the next one overlaps the device, as the reference's
 * accelerated polisher fills its next batch while one computes (src/cuda/cudapolisher.cpp:246-277). */
int         vc_wb_build_begin(vc_wb* b, vc_batch* out);
int         vc_wb_build_fill(vc_wb* b, uint32_t w_lo, uint32_t w_hi);
/* add_layer() index (0 = backbone) of every stored sequence of the last build, i.e. the rank permutation */
const uint32_t* vc_wb_seq_orig(const vc_wb* b);
uint32_t    vc_wb_n_windows(const vc_wb* b);
uint32_t    vc_wb_window_target(const vc_wb* b, uint32_t w);
uint32_t    vc_wb_window_rank(const vc_wb* b, uint32_t w);
void        vc_wb_window_ids(const vc_wb* b, uint32_t* target /*[n_windows]*/, uint32_t* rank /*[n_windows]*/);
/* per-target concatenation of the window results with the LN/RC/XC tags; fragment_correction adds the "r" */
int         vc_wb_stitch(vc_wb* b, const vc_result* res, int drop_unpolished, int fragment_correction);
uint32_t    vc_wb_n_polished(const vc_wb* b);
const char* vc_wb_polished_name(const vc_wb* b, uint32_t i);
const char* vc_wb_polished_data(const vc_wb* b, uint32_t i, uint64_t* length);

/* ------------------------------------------------------------------------------------------------
 * File formats (host only).  Stands in for what Polisher::initialize does with bioparser before a window
 * exists (src/polisher.cpp:77-138 parser selection, :207-352 loading and filtering) and for the record constructors of
 * src/sequence.cpp:19-42 and src/overlap.cpp:14-110.  vechat_amd/csrc/vc_io.cpp.
 * ------------------------------------------------------------------------------------------------ */
typedef struct vc_seqset vc_seqset;      /* the records of one FASTA / FASTQ (.gz) file */
typedef struct vc_ovlset vc_ovlset;      /* the records of one MHAP / PAF / SAM (.gz) file */
/* Always returns a set; vc_seqset_error() != NULL says what went wrong.  keep_names: NULL, or '\n'-separated names -- the other
 * records are passed over.  names_only: names and lengths, no data (a rank of a multi-GPU run plans with that). */
vc_seqset*      vc_io_read_sequences(const char* path, const char* keep_names, int names_only);
void            vc_seqset_free(vc_seqset* s);
const char*     vc_seqset_error(const vc_seqset* s);
uint64_t        vc_seqset_size(const vc_seqset* s);
const uint64_t* vc_seqset_name_off(const vc_seqset* s);   /* [n+1] into vc_seqset_names */
const char*     vc_seqset_names(const vc_seqset* s);
const uint64_t* vc_seqset_data_off(const vc_seqset* s);   /* [n+1] into vc_seqset_data / vc_seqset_qual */
const char*     vc_seqset_data(const vc_seqset* s);       /* upper-cased (sequence.cpp:19-42) */
const char*     vc_seqset_qual(const vc_seqset* s);       /* NULL when no record has a quality string */
const uint8_t*  vc_seqset_has_qual(const vc_seqset* s);   /* [n]; an all-'!' quality string counts as none */
const uint64_t* vc_seqset_lengths(const vc_seqset* s);    /* [n] */
typedef struct {
    const char* q_name; uint32_t q_name_len;              /* not NUL-terminated */
    const char* t_name; uint32_t t_name_len;
    uint8_t  by_index;                                    /* MHAP: q_index / t_index are positions in the reads / targets files */
    uint32_t q_index, t_index;
    uint8_t  strand;                                      /* 1 = reverse complement */
    uint32_t q_begin, q_end, q_length, t_begin, t_end, length;
    double   error;                                       /* 1 - min(spans) / max(spans) (overlap.cpp:21-26) */
    const char* cigar;                                    /* NULL: none yet (plain PAF, MHAP) */
    uint8_t  dropped;                                     /* could not be aligned (vc_ovlset_set_cigar(.., NULL)) */
} vc_overlap_rec;
vc_ovlset*  vc_io_read_overlaps(const char* path);        /* format from the extension, like the reference */
void        vc_ovlset_free(vc_ovlset* o);
const char* vc_ovlset_error(const vc_ovlset* o);
uint64_t    vc_ovlset_size(const vc_ovlset* o);
int         vc_ovlset_get(const vc_ovlset* o, uint64_t i, vc_overlap_rec* out);
int         vc_ovlset_set_cigar(vc_ovlset* o, uint64_t i, const char* cigar);
/* Planning for one rank of a multi-GPU run, on names, lengths and overlap records only (targets / reads may be names-only sets):
 * cost[k] = length of target k + the target bases its overlaps cover; and the '\n'-separated names (malloc'ed, vc_io_free) a rank
 * that owns targets [t_lo, t_hi) must load -- those targets and the queries of the overlaps on them. */
int         vc_io_target_cost(const vc_ovlset* o, const vc_seqset* targets, double* cost /*[targets]*/);
char*       vc_io_rank_names(const vc_ovlset* o, const vc_seqset* targets, const vc_seqset* reads, uint64_t t_lo, uint64_t t_hi, uint64_t* n_names);
void        vc_io_free(void* p);
/* Polisher::initialize, fragment-correction mode: every target, every read (a read that is also a target shares its record),
 * every overlap that survives the filters, into the window builder.  Returns the number of overlaps kept, -1 on error. */
int64_t     vc_io_load(vc_wb* builder, const vc_seqset* targets, const vc_seqset* reads, vc_ovlset* overlaps, double error_threshold,
                       int allow_empty, int* window_type, char* err, uint64_t err_cap);

/* ------------------------------------------------------------------------------------------------
 * Overlap alignment on the device.  Stands in for the edlib call the reference makes
 * for overlaps without a CIGAR (src/overlap.cpp:205-220): global unit-cost alignment with path, returned
 * as an edlib-standard CIGAR (M / I / D).  The path is optimal; which of several optimal paths is returned
 * is this library's choice (diagonal, then insertion, then deletion, from the end), not edlib's.
 * q / t hold the pieces to align back to back (the query piece already oriented as it aligns);
 * Any length: scores are kept relative per 2048-column tile.  An overlap whose stored matrix would not fit the
 * device memory is not aligned: its distance comes back as -1 and its CIGAR empty; the others are unaffected.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n;
    const uint64_t* q_off;   /* [n+1] */
    const uint8_t*  q;
    const uint64_t* t_off;   /* [n+1] */
    const uint8_t*  t;
} vc_align_batch;
int         vc_align(int device, const vc_align_batch* b, char* cigar, uint64_t cigar_cap, uint64_t* cigar_off,
                     int32_t* edit_distance);
const char* vc_align_last_error(void);
/* the aligner keeps its (large) matrix buffer between calls; this gives it back */
void        vc_align_release(void);

#ifdef __cplusplus
}
#endif
#endif
