/* compseed_amd.h -- C ABI of the MI355X-native compressive SMEM seeding engine.
 *
 * This is the drop-in boundary for ONE path of i-xiaohu/CompSeed: per-read SMEM collection over the FM-index
 * plus the suffix-array lookup that turns the intervals into seeds, i.e. the two blocks
 *     "Collect exact matches"  mapping/comp_seed.cpp:2247-2304   (== mem_collect_intv, mapping/bwamem.c:218-272)
 *     "SAL"                    mapping/comp_seed.cpp:2306-2347
 * of seed_and_extend(), called once per chunk from mem_process_seqs (comp_seed.cpp:2527).  The reference has no
 * FFI of its own (everything is statically linked); INTEGRATION.md shows the 20-line patch that makes
 * seed_and_extend() consume this API.  Plain C types only: no torch, no HIP types, no C++.
 *
 * All functions return 0 on success or a negative CS_E* code; cs_last_error() gives the message for the calling
 * thread.  Nothing in here aborts the process (the reference aborts/exits on error: bwalib/utils.c:92-124).
 * An engine is bound to one GPU and must be driven by one host thread at a time; engines are independent.
 */
#ifndef COMPSEED_AMD_H
#define COMPSEED_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_OK          0
#define CS_EINVAL     -1   /* bad argument                                   */
#define CS_EIO        -2   /* index file missing / truncated / inconsistent  */
#define CS_ENOMEM     -3   /* host or device allocation failed               */
#define CS_EDEVICE    -4   /* HIP runtime error (no GPU, launch failure ...) */
#define CS_ERANGE     -5   /* index or read too large for the device layout  */

/* bi-interval; identical layout and meaning to bwtintv_t (FM_index/bwt.h:62-64):
 * x0 = SA start of the match, x1 = SA start of its reverse complement, x2 = occurrences,
 * info = (query begin << 32) | query end for a finished mem (comp_seed.cpp:127). */
typedef struct { uint64_t x0, x1, x2, info; } cs_intv_t;

/* seed = the fields of mem_seed_t that seeding fills (mapping/comp_seed.h:77-83; score == len, aln unused) */
typedef struct { int64_t rbeg; int32_t qbeg, len; } cs_seed_t;

/* read-only view of the FM-index arrays a bwt_t holds (FM_index/bwt.h:48-60).  `bwt` is the interleaved
 * Occ/BWT array exactly as stored in <prefix>.bwt (64-byte blocks, bwt.h:73-80); `sa` is the sampled suffix
 * array with sa[0] == (uint64_t)-1 as bwt_restore_sa leaves it (bwt.c:437). */
typedef struct {
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	uint64_t bwt_size;      /* number of 32-bit words in bwt */
	const uint32_t *bwt;
	uint64_t sa_intv;       /* power of two, 32 for bwaidx-built indexes */
	uint64_t n_sa;
	const uint64_t *sa;
} cs_index_view_t;

/* the mem_opt_t fields the seeding path reads (mapping/comp_seed.h:50-59), CLI flags in brackets */
typedef struct {
	int32_t  min_seed_len;   /* [-k] 19   */
	float    split_factor;   /* [-r] 1.5  */
	int32_t  split_width;    /* [-s] 10   */
	int32_t  max_occ;        /* [-c] 500  */
	uint64_t max_mem_intv;   /* [-y] 20   */
	int32_t  want_sal;       /* 1: also produce seeds (SAL block), 0: mems only */
	int32_t  sst_mode;       /* 1: on-device SST (LDS-resident memo of bwt_extend, mapping/SST.h) on [default], 0: off.
	                          * Results are identical either way; only speed and the bwt_calls counter differ */
	uint32_t disable;        /* bit mask of CS_DISABLE_*: switch single exact shortcuts off for this call (A/B parity tests,
	                          * profiling).  0 [default] = everything the engine was created with is used */
	uint32_t count_traffic;  /* 1: run the instantiations of the SMEM kernels that count their index-side accesses
	                          * (cs_engine_traffic_model); same results, a few per cent slower.  0 [default] */
} cs_params_t;

/* cs_params_t.disable: every mechanism below is exact; switching one off changes speed and counters only */
#define CS_DISABLE_TEXT_MODE    0x01u  /* unique forward matches compared against the 2-bit text instead of the FM index      */
#define CS_DISABLE_R2_TEXT      0x02u  /* re-seeding calls of unique SMEMs answered from rep[] / lcp[] / inverse SA           */
#define CS_DISABLE_TEXT_SWEEP   0x04u  /* backward sweeps bounded by the previous pivot read off the text                     */
#define CS_DISABLE_WINDOW       0x08u  /* window scheme for the backward sweeps (jump table + k-mer filter)                   */
#define CS_DISABLE_R3_TEXT      0x10u  /* round-3 seeds inside a mem taken from the text arrays                               */
#define CS_DISABLE_KMER_FILTER  0x20u  /* k-mer filter in front of the window lanes                                           */
#define CS_DISABLE_FWD0         0x40u  /* separate lean kernel for the calls at the first base of each read                   */

/* engine construction options: which derived arrays are materialised in HBM and how the working buffers are sized.
 * cs_engine_options_default() fills the defaults (in brackets).  Everything optional is also skipped automatically when
 * it does not fit next to the index; the mechanisms that need it are then off.  Results never depend on any of this. */
typedef struct {
	int32_t full_sa;          /* [1] full suffix array in HBM: SAL is one gather instead of a bwt_sa walk (bwt.c:86-96)       */
	int32_t sa64;             /* [0] 8-byte suffix-array / inverse-SA entries even below 2^32 rows (what hg19 scale uses)      */
	int32_t text_mode;        /* [1] 2-bit text + inverse suffix array (needs full_sa)                                        */
	int32_t text_arrays;      /* [1] lcp[] / rep[] byte arrays (needs text_mode): re-seeding and round 3 from the text        */
	int32_t jump_k;           /* [15] bi-interval table of every jump_k-mer, 6..15; 0 = none                                  */
	int32_t kmer_filter;      /* [1] filter over all min_seed_len-mers of the text for the window lanes                       */
	int32_t fused;            /* [0] 1: the fused one-lane-per-read kernel instead of the split forward/backward kernels      */
	int32_t mem_cap;          /* [64] mems per read held in the first-pass arena (more go through the overflow records)       */
	int64_t lep_arena_mb;     /* [16384] arena of the forward passes' left-extension points, per pass context; smaller = more chunks per pass */
	int64_t max_raw_mb;       /* [24576] first-pass mem arena; smaller = a batch is processed in more sub-batches             */
	int32_t r3_text_iter;     /* [5] forward launch after which round 3 starts on its side stream                             */
	int32_t pipeline_reads;   /* [5000000] reads per sub-batch of the host variants (cs_engine_seed_batch, _packed): upload of the next,
	                           *     seeding of the current and download of the previous sub-batch overlap; 0 = the whole batch at once */
	int32_t expand_threads;   /* [16] host threads that expand packed results in cs_engine_seed_batch                         */
	int32_t count_sal_merged; /* [0] 1: also count the distinct SA slots per 512 reads (cs_stats_t.sal_calls as the reference
	                           *     counts them, comp_seed.cpp:2327-2345); costs one sort of the slots per call              */
	int32_t verbose;          /* [0] progress lines on stderr                                                                 */
	int32_t host_pack_threads;/* [8] host variants: the reads are turned into the kernels' 16-byte / 32-base records by this many
	                           *     host threads and cross PCIe as 0.6 bytes per base; 0 = upload the caller's bytes, pack on the GPU */
	int32_t passes_in_flight; /* [2] seeding passes the engine runs at the same time (parts of a blocking call, batches of a stream, device
	                           *     batches of cs_engine_submit_device): the thin tail of one pass is filled by the next.  2 costs a second
	                           *     set of working buffers (allocated on first use); 1 = one pass at a time                          */
	int32_t reserved[3];      /* must be 0 */
} cs_engine_options_t;

/* CSR result of one batch.  Read r owns mems[mem_off[r] .. mem_off[r+1]) sorted by info (comp_seed.cpp:2301) and
 * seeds[seed_off[r] .. seed_off[r+1]) in mem-then-slot order (comp_seed.cpp:2311-2325).  The arrays belong to the
 * engine and stay valid until the next seed call on the same engine or its destruction. */
typedef struct {
	int64_t   n_reads;
	uint64_t  n_mems, n_seeds;
	const uint64_t  *mem_off;   /* n_reads + 1 */
	const cs_intv_t *mems;
	const uint64_t  *seed_off;  /* n_reads + 1; NULL when want_sal == 0 */
	const cs_seed_t *seeds;
} cs_result_t;

/* counters in the spirit of display_profile (main.cpp:203-214), accumulated over the engine's lifetime */
typedef struct {
	uint64_t reads, bases, mems, seeds;
	uint64_t bwt_queries;      /* bwt_extend queries evaluated on the device ("BWT-extend queries"); equals the
	                              reference's count unless re-seeding calls were answered from the text (below)  */
	uint64_t bwt_calls;        /* of those, served from HBM/L2, i.e. not by the on-device SST               */
	uint64_t sal_queries;      /* SA slots requested                                                        */
	uint64_t sal_calls;        /* distinct SA slots per 512 consecutive reads (what CompSeed looks up after merging,
	                              comp_seed.cpp:2327-2345) when the engine was created with count_sal_merged;
	                              otherwise == sal_queries.  The device looks every slot up: one gather each      */
	uint64_t overflow_mems;    /* mems beyond a read's first mem_cap (split kernels), or reads that needed the
	                              large-capacity second pass (fused kernel)                                        */
	double   seed_kernel_ms;   /* accumulated HIP-event time of the first-pass SMEM kernel launches         */
	double   sal_kernel_ms;    /* ... of the SAL kernels                                                    */
	double   total_ms;         /* ... of whole seed calls, first launch to last                             */
	uint64_t seed_kernel_launches;
	double   overflow_kernel_ms;       /* ... of the second-pass SMEM launches over overflowed reads        */
	uint64_t overflow_kernel_launches;
	uint64_t reseed_text_calls;        /* re-seeding calls (bwamem.c:241-249) answered from the text arrays
	                                      instead of the FM index: same SMEMs, their bwt_extend queries never run */
	uint64_t reseed_index_calls;       /* re-seeding calls of unique SMEMs that had to stay on the FM index       */
	uint64_t sweep_text_calls;         /* round-1 calls whose backward sweep (bwt.c:325-345) was read off the text  */
	uint64_t r3_text_seeds;            /* round-3 seeds (bwt.c:357-381) whose bi-interval came from the text arrays  */
} cs_stats_t;

typedef struct cs_index  cs_index_t;   /* host copy of an index loaded from files */
typedef struct cs_engine cs_engine_t;

const char *cs_last_error(void);
const char *cs_version(void);

void cs_params_default(cs_params_t *p);                     /* mem_opt_init, comp_seed.cpp:26-58 */

/* ---- index files: <prefix>.bwt and <prefix>.sa in the reference's formats (bwt_restore_bwt / bwt_restore_sa,
 *      FM_index/bwt.c:421-462; prefix inference as bwa_idx_infer_prefix, bwalib/bwa.c:244) */
int  cs_index_load(const char *prefix, cs_index_t **out);
int  cs_index_view(const cs_index_t *idx, cs_index_view_t *view);
void cs_index_free(cs_index_t *idx);

/* ---- index construction on the GPU (replaces `bwaidx`, FM_index/index_main.c:257-325, for the .bwt/.sa pair):
 *      `fwd_nt4` = forward strand, one base per byte, codes 0..3 (ambiguous bases already replaced, as
 *      bns_fasta2bntseq does with lrand48, bntseq.c:295); the reverse complement is appended internally.
 *      cs_index_save writes <prefix>.bwt / <prefix>.sa byte-identical to bwt_dump_bwt / bwt_dump_sa (bwt.c:385-407). */
int  cs_index_build(const uint8_t *fwd_nt4, uint64_t l_pac, int device, cs_index_t **out);
#define CS_BUILD_FORCE_64BIT 1u   /* run the 64-bit suffix sorter on a genome that would fit 32 bits (tests) */
#define CS_BUILD_VERBOSE     2u
int  cs_index_build_flags(const uint8_t *fwd_nt4, uint64_t l_pac, int device, uint32_t flags, cs_index_t **out);
int  cs_index_save(const cs_index_t *idx, const char *prefix);

/* ---- reference sequences: the FASTA front end of the builder and the other three index files.  cs_refseq_from_fasta reads FASTA / FASTQ
 *      (plain or gzip) as bns_fasta2bntseq does (FM_index/bntseq.c:280-330): contig table, holes (runs of one ambiguity code), and the
 *      forward strand as codes 0..3 with every ambiguous base replaced by lrand48() & 3 after srand48(11) (bntseq.c:266,295);
 *      cs_refseq_save writes <prefix>.pac / .ann / .amb byte-identical to bwaidx (bns_dump, bntseq.c:65-95);
 *      cs_index_build_fasta is bwa_idx_build (index_main.c:257-325): all five files from a FASTA, the suffix sort on the GPU. */
typedef struct cs_refseq cs_refseq_t;
int  cs_refseq_from_fasta(const char *path, cs_refseq_t **out);
int  cs_refseq_codes(const cs_refseq_t *r, const uint8_t **fwd_nt4, uint64_t *l_pac, int32_t *n_seqs, int32_t *n_holes);
int  cs_refseq_save(const cs_refseq_t *r, const char *prefix);
void cs_refseq_free(cs_refseq_t *r);
int  cs_index_build_fasta(const char *fasta, const char *prefix, int device);

/* ---- engine: uploads the index to GPU `device` once (replaces bwa_idx_load_from_shm's role: HBM residency) */
int  cs_device_count(int *n);
void cs_engine_options_default(cs_engine_options_t *o);
int  cs_engine_create(const cs_index_view_t *index, int device, cs_engine_t **out);            /* default options */
int  cs_engine_create_opts(const cs_index_view_t *index, int device, const cs_engine_options_t *opts, cs_engine_t **out);
void cs_engine_destroy(cs_engine_t *e);

/* ---- the hot path.  `bases` holds the reads back to back (ASCII or already nt4-coded 0..4, as CompSeed accepts:
 *      comp_seed.cpp:2258-2260); read r is bases[offsets[r] .. offsets[r+1]).  The caller's buffers are not modified
 *      (the reference overwrites seq in place).  Host variant: pointers are host memory, results land in pinned host
 *      memory.  Device variant: pointers are device memory on the engine's GPU and the result arrays are device
 *      pointers (no PCIe traffic inside the call).  The engine works on streams of its own: device inputs must be
 *      COMPLETE when the call is made (synchronise the stream that produced them first), d_offsets[0] must be 0 and
 *      d_offsets[n_reads] == n_bases (checked: CS_EINVAL); results are complete when the call returns.  Nothing behind
 *      d_bases[n_bases - 1] is read. */
int  cs_engine_seed_batch(cs_engine_t *e, const cs_params_t *par, int64_t n_reads,
                          const uint8_t *bases, const uint64_t *offsets, cs_result_t *out);
int  cs_engine_seed_batch_device(cs_engine_t *e, const cs_params_t *par, int64_t n_reads,
                                 const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_bases, cs_result_t *out);
/* The device variant as a stream of batches, up to TWO in flight (cs_engine_options_t.passes_in_flight): batch n is seeded on pass
 * context n & 1 -- a second set of streams and working buffers over the same index -- so the thin tail of one pass (late iterations,
 * sort, SAL, host round trips) runs beside the dense start of the next; this is how a worker that keeps its reads in HBM drives the engine
 * (the reference's kt_for over chunks of a batch, comp_seed.cpp:2541-2548, is the same overlap on CPU threads).  Inputs as for
 * cs_engine_seed_batch_device, and they must stay untouched until their batch has been collected.  Results come back in submission
 * order; the device pointers of a collected batch stay valid until the SECOND submit after its collect (the next batch on its context).
 * While device batches are in flight the other entry points that use the device return CS_EINVAL.  Threads: as for cs_engine_submit /
 * cs_engine_collect_packed -- one submitting thread, one collecting thread. */
int  cs_engine_submit_device(cs_engine_t *e, const cs_params_t *par, int64_t n_reads,
                             const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_bases);
int  cs_engine_collect_device(cs_engine_t *e, cs_result_t *out);

/* ---- packed results: what actually crosses PCIe.  A 10 M-read batch produces ~80 M mems and ~200 M seeds, 5.9 GB as cs_intv_t /
 *      cs_seed_t -- more transfer time than seeding time -- but half of those bytes are implied: a seed's qbeg and len are its
 *      mem's, a mem's seeds are its first min(x2, max_occ) slots (comp_seed.cpp:2313-2325), and three 33-bit coordinates, a 32-bit
 *      size and two 15-bit query positions fit 16 bytes, a reference position 5.  cs_engine_seed_batch_packed returns that form in pinned host memory;
 *      a consumer that copies per read anyway (the reference does: aux.match[r] / aux.seed[r] are per-read vectors) unpacks on
 *      the fly in its own worker threads with the inline helpers below, and cs_engine_seed_batch is this call followed by a
 *      multi-threaded expansion into cs_intv_t / cs_seed_t arrays.  Both split the batch into sub-batches and overlap the upload
 *      of the next, the seeding of the current and the download (and expansion) of the previous one.
 *      mem_format CS_MEM_PACKED16 needs an index shorter than 2^33 symbols and reads shorter than 2^15 bases (hg19 / T2T, short
 *      reads); anything else comes back as CS_MEM_FULL32 (plain cs_intv_t); seeds as their rbeg in 40 bits either way. */
#define CS_MEM_FULL32   0
#define CS_MEM_PACKED16 1
typedef struct { uint64_t w0, w1; } cs_mem16_t;
/* w0 = x0 | (x2 & 0x7fffffff) << 33;   w1 = x1 | beg << 33 | end << 48 | (x2 >> 31) << 63 */
typedef struct {
	int64_t   n_reads;
	uint64_t  n_mems, n_seeds;
	int32_t   mem_format;        /* CS_MEM_PACKED16 or CS_MEM_FULL32 */
	int32_t   max_occ;           /* the -c the batch was seeded with: a mem owns min(x2, max_occ) consecutive entries of seed_rbeg */
	const uint64_t *mem_off;     /* n_reads + 1 */
	const void     *mems;        /* cs_mem16_t[n_mems] or cs_intv_t[n_mems] */
	const uint64_t *seed_off;    /* n_reads + 1, NULL when want_sal == 0 */
	int32_t   seed_format;       /* CS_SEED_RBEG40: a seed's rbeg as 40 bits in two planes (reference positions are below 2^37) */
	int32_t   reserved;
	const uint32_t *seed_rbeg_lo; /* n_seeds low words, in mem-then-slot order */
	const uint8_t  *seed_rbeg_hi; /* n_seeds fifth bytes: rbeg = lo | (int64_t)hi << 32 (cs_packed_seed_rbeg) */
} cs_packed_result_t;
#define CS_SEED_RBEG40 1
static inline int64_t cs_packed_seed_rbeg(const cs_packed_result_t *r, uint64_t i) { return (int64_t)((uint64_t)r->seed_rbeg_lo[i] | (uint64_t)r->seed_rbeg_hi[i] << 32); }
int  cs_engine_seed_batch_packed(cs_engine_t *e, const cs_params_t *par, int64_t n_reads,
                                 const uint8_t *bases, const uint64_t *offsets, cs_packed_result_t *out);
/* The same as a pipeline across batches, the counterpart of the reference's kt_pipeline (main.cpp:438: read chunk n+1 while chunk n is
 * processed and chunk n-1 written): cs_engine_submit queues a batch and returns at once (at most FOUR in flight; the caller's buffers
 * must stay untouched until the batch is collected), cs_engine_collect_packed blocks until the OLDEST submitted batch is complete.  With
 * batches kept submitted the upload of batch n+1, the seeding of batch n (of two batches at a time, on the engine's two pass contexts) and
 * the download of batch n-1 overlap, and throughput is that of the slowest stage instead of their sum.  How many to keep submitted: a batch
 * spends upload + seeding + download in the engine, and a new one enters only when the oldest has been collected, so the stream runs at
 * (that latency) / (batches in flight) per batch until the slowest stage takes over -- two in flight (what a two-thread kt_pipeline gives)
 * leave the engine idle half of the time, three to four keep it busy.  A collected result stays valid until the next collect or blocking
 * seed call.
 * While batches are in flight the engine's other entry points that use the device return CS_EINVAL (cs_engine_reset_stats does nothing).
 * Threads: cs_engine_submit may be called from ONE thread while cs_engine_collect_packed runs on ONE other thread (the reader and the
 * processing step of a kt_pipeline, main.cpp:60-126); everything else on an engine needs the caller's own serialisation. */
int  cs_engine_submit(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases, const uint64_t *offsets);
int  cs_engine_collect_packed(cs_engine_t *e, cs_packed_result_t *out);
static inline void cs_unpack_mem(const cs_packed_result_t *r, uint64_t i, cs_intv_t *m)
{
	if (r->mem_format == CS_MEM_PACKED16) {
		const cs_mem16_t p = ((const cs_mem16_t *)r->mems)[i];
		m->x0 = p.w0 & 0x1ffffffffull; m->x1 = p.w1 & 0x1ffffffffull;
		m->x2 = (p.w0 >> 33) | ((p.w1 >> 63) << 31);
		m->info = ((p.w1 >> 33) & 0x7fffull) << 32 | ((p.w1 >> 48) & 0x7fffull);
	} else *m = ((const cs_intv_t *)r->mems)[i];
}
static inline uint32_t cs_mem_seed_count(const cs_intv_t *m, int32_t max_occ) { return m->x2 < (uint64_t)max_occ ? (uint32_t)m->x2 : (uint32_t)max_occ; }
/* pinned host memory for the caller's read chunk (uploads from it are asynchronous and run at link speed; pageable memory works
 * too, through a staging thread) */
int  cs_host_alloc(size_t bytes, void **ptr);
int  cs_host_free(void *ptr);
/* What the host variants upload when cs_engine_options_t.host_pack_threads > 0: the reads as the records the seeding kernels read,
 * 16 bytes per 32 bases (words 0-1: the bases, 2 bits each, base j in bits 2j..2j+1; word 2: one bit per base that is ambiguous or lies
 * behind the end of the read; word 3: 0), record k of read r at (offsets[r] >> 5) + r + k, (offsets[n_reads] >> 5) + n_reads records in
 * all; letters as nst_nt4_table maps them (FM_index/bntseq.c:46-63), codes 0..3 as they are (comp_seed.cpp:2259).  Exported for tests. */
int  cs_pack_reads(const uint8_t *bases, const uint64_t *offsets, int64_t n_reads, void *records, int threads, uint32_t flags);
#define CS_PACK_SCALAR 1u         /* do not use the AVX2 / BMI2 code path */

/* ---- reordered-reads ingest: the reader step of the reference's pipeline (input_reorder_reads, main.cpp:36-58; FASTQ when the first
 *      byte is '@', main.cpp:399-406; plain or gzip), cutting chunks as main.cpp:54,437 does (the first even read count that reaches
 *      chunk_bases) and delivering them as the engine takes them: bases back to back in pinned memory + n_reads + 1 offsets.  Two
 *      chunk buffers alternate, so a chunk stays intact while the next one is read (submit chunk n+1, collect chunk n, read chunk n+2).
 *      n_reads == 0: end of input. */
typedef struct cs_reader cs_reader_t;
int  cs_reader_open(const char *path, int64_t chunk_bases, cs_reader_t **out);
int  cs_reader_next(cs_reader_t *r, const uint8_t **bases, const uint64_t **offsets, int64_t *n_reads);
void cs_reader_close(cs_reader_t *r);

/* ---- seed -> chain hand-off: mem_chain of the reference (mapping/comp_seed.cpp:241-285) over the seeds of a whole batch, host code.
 *      A chain is a run of co-linear seeds on one reference sequence (test_and_merge, comp_seed.cpp:182-203); chains come out per read in
 *      the order the reference's B-tree traversal gives them, each with the fraction of the read covered by repetitive mems.  The contig
 *      table comes from <prefix>.ann.  `seeds` is a host-side result with seeds (want_sal = 1); read_offsets are the batch's offsets.
 *      The arrays of the result belong to the chainer and stay valid until its next call. */
typedef struct { int32_t w, max_chain_gap, min_seed_len, max_occ; } cs_chain_params_t;   /* mem_opt_t: -w 100, max_chain_gap 10000, -k, -c */
typedef struct { int64_t pos; int32_t rid, n_seeds; float frac_rep; int32_t is_alt; } cs_chain_t;
typedef struct {
	int64_t n_reads; uint64_t n_chains, n_seeds;
	const uint64_t *chain_off;     /* n_reads + 1 */
	const cs_chain_t *chains;
	const uint64_t *cseed_off;     /* n_chains + 1 */
	const cs_seed_t *cseeds;       /* the seeds of chain c: cseeds[cseed_off[c] .. cseed_off[c + 1]) */
} cs_chain_result_t;
typedef struct cs_chainer cs_chainer_t;
int  cs_chainer_create(const char *prefix, cs_chainer_t **out);
void cs_chainer_destroy(cs_chainer_t *c);
void cs_chain_params_default(cs_chain_params_t *p);
int  cs_chain_batch(cs_chainer_t *c, const cs_chain_params_t *par, const cs_result_t *seeds, const uint64_t *read_offsets, int n_threads,
                    cs_chain_result_t *out);

/* ---- the chain filters between chaining and extension (comp_seed.cpp:2364-2367), host code: mem_chain_flt (comp_seed.cpp:297-360: chains
 *      by descending weight -- klib's introsort, whose order among equal weights is reproduced --, chains shadowed on the read by a much
 *      heavier one dropped, the first shadowed chain of each kept one retained) and mem_flt_chained_seeds (comp_seed.cpp:393-412: for reads
 *      of ~700 bases and more, short seeds whose neighbourhood does not reach a local alignment score of 5.5 ln(read length) are dropped and
 *      the others carry that score; ksw_align2's number, bwalib/ksw.c:343).  `in`: cs_chain_batch's result (or the caller's chains in that
 *      form); `out`: the surviving chains per read in the reference's order, `cseed_score` their seeds' scores -- what cs_extend_chains takes.
 *      `bases` / <prefix>.pac are read only when a read is long enough for the seed test.  The result belongs to the chainer and stays
 *      valid until its next cs_chain_filter. */
typedef struct { int32_t min_chain_weight, max_chain_extend, max_chain_gap, min_seed_len; float mask_level, drop_ratio;   /* mem_opt_t: min_chain_weight (-W), max_chain_extend, max_chain_gap, -k, mask_level, drop_ratio (-D) */
                 int32_t a, b, o_del, e_del, o_ins, e_ins; } cs_flt_params_t;                                              /* ... -A -B -O -E (the seed test) */
void cs_flt_params_default(cs_flt_params_t *p);
int  cs_chain_filter(cs_chainer_t *c, const cs_flt_params_t *par, const cs_chain_result_t *in, const uint8_t *bases, const uint64_t *read_offsets,
                     int n_threads, cs_chain_result_t *out, const int32_t **cseed_score);

/* ---- seed extension: the banded Smith-Waterman extensions of mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319), i.e.
 *      what the reference hands to BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (mapping/bandedSWA.h:117-167; call sites
 *      comp_seed.cpp:1719,1790,1859,1942,2003,2074), computed on the GPU with the exact semantics of ksw_extend2 (bwalib/ksw.c:380-479): band of
 *      +-w diagonals that shrinks to the live columns, Z-drop, end bonus in the band limit only.  A cs_extender_t corresponds to a
 *      BandedPairWiseSW object (its constructor's parameters, bandedSWA.h:117-121); cs_ext_pair_t to the input fields of SeqPair (idq, idr,
 *      len2, len1, h0; bandedSWA.h:90-100) with 64-bit offsets into the two sequence buffers (codes 0..4, one byte per base; the query and
 *      target of a left extension reversed by the caller as comp_seed.cpp:1525,1546 do); cs_ext_result_t to its output fields.  All pairs of
 *      a call share the band width w (the reference calls once per band try: w, then 2w for the pairs whose max_off came close to the band,
 *      comp_seed.cpp:1740-1742).  Scoring: pairs the reference would send to its vectorised code (both lengths and h0 + min(len) * match
 *      below 32768, comp_seed.cpp:1569-1577) compare base codes -- ambiguous (4) on either side scores -1, equal codes mat[0], others mat[1]
 *      (bandedSWA.cpp:286-290) -- longer ones index the 5 x 5 matrix like ksw_extend2; the two only differ for codes above 4.
 *      Known deviation of the REFERENCE from its own definition, not reproduced: its vectorised code gives other numbers than ksw_extend2
 *      in a few pairs per thousand when o_del != o_ins or a gap extension is not 1 (tests/test_oracle_bsw.py); this library computes
 *      ksw_extend2's result for every parameter set.  Host variant: host pointers in, results in `out`; device variant: everything is device
 *      memory on the extender's GPU.  Pairs with qlen < 1, tlen < 0 or offsets outside the buffers get a zero result and the call returns
 *      CS_EINVAL after delivering the others. */
typedef struct { int8_t mat[25]; int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus; uint32_t flags; } cs_ext_params_t;
/* flags.  Default 0: queries of up to 160 columns whose scores fit 15 bits (every extension of a 150-bp read) run one pair per LANE, the rest
 * one wavefront per pair, a query column per lane.  The others are A/B switches; every combination gives the same results. */
#define CS_EXT_PACKED16     1u   /* wave-per-pair only, queries longer than 64 bases through a kernel with two columns per lane (packed int16); exact, slower */
#define CS_EXT_PACKED16_ALL 2u   /* ... all queries of the 16-bit class through it */
#define CS_EXT_NO_LANES     4u   /* do not use the one-pair-per-lane kernel for short queries (A/B tests): every pair goes one wave per pair */
#define CS_EXT_LANES_QIN    8u   /* lane kernel, 8-bit class: the query base inside the score cell (one LDS read per cell, a third more LDS); exact; an experiment */
typedef struct { uint64_t q_off, t_off; int32_t qlen, tlen, h0, reserved; } cs_ext_pair_t;   /* reserved: not looked at (the caller's own tag, e.g. the region a pair belongs to) */
typedef struct { int32_t score, qle, tle, gtle, gscore, max_off; } cs_ext_result_t;
typedef struct { uint64_t pairs, cells, rows, launches; double kernel_ms; } cs_ext_stats_t;   /* cells = DP cells computed (inside the adaptive band) */
typedef struct cs_extender cs_extender_t;
void cs_ext_params_default(cs_ext_params_t *p);               /* mem_opt_init's scoring: a 1, b 4, o 6, e 1, zdrop 100, pen_clip 5 (comp_seed.cpp:26-58) */
int  cs_extender_create(int device, const cs_ext_params_t *par /* NULL: defaults */, cs_extender_t **out);
void cs_extender_destroy(cs_extender_t *x);
int  cs_extend_batch(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *pairs, const uint8_t *qbuf, uint64_t q_bytes,
                     const uint8_t *tbuf, uint64_t t_bytes, int32_t w, cs_ext_result_t *out);
int  cs_extend_batch_device(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *d_pairs, const uint8_t *d_qbuf, uint64_t q_bytes,
                            const uint8_t *d_tbuf, uint64_t t_bytes, int32_t w, cs_ext_result_t *d_out);
int  cs_extender_stats(const cs_extender_t *x, cs_ext_stats_t *st);
/* the band retries extend the same sequences again with another pair list: upload the two buffers once, then run pair lists against them */
int  cs_extender_upload(cs_extender_t *x, const uint8_t *qbuf, uint64_t q_bytes, const uint8_t *tbuf, uint64_t t_bytes);
int  cs_extend_batch_resident(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *pairs, int32_t w, cs_ext_result_t *out);

/* ---- the extension stage as a whole: mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319-2237), chains in, alignment regions out.
 *      For every seed of every chain (as the caller's chain filters left them: mem_chain_flt / mem_flt_chained_seeds, comp_seed.cpp:2364-2367,
 *      are the caller's) a region is opened, extended left and then right by banded Smith-Waterman on the GPU -- band w, once more with 2w
 *      where the alignment came within a quarter of the band's edge (MAX_BAND_TRY 2) --, and regions made redundant by an earlier region
 *      of the read are marked qb = qe = -1 (the caller drops them as comp_seed.cpp:2387-2393 does).  cs_alnreg_t holds the fields of
 *      mem_alnreg_t this stage fills (comp_seed.h:106-125); `chain` is the index of the region's chain within its read.  Regions come in the
 *      reference's order: chain by chain, seeds by descending score.  `chains`: the chains as a cs_chain_result_t -- cs_chain_batch's output or the caller's
 *      own --, `cseed_score` the chained seeds' scores in the same order (NULL: score = len, what mem_chain leaves), `bases` / `read_offsets` the
 *      batch's reads (ASCII or codes; a '-' is code 5 as in nst_nt4_table).  The aligner loads <prefix>.ann / .alt / .pac.  The result arrays
 *      belong to the aligner and stay valid until its next call.  pen_clip5 must equal pen_clip3 for now. */
typedef struct { int32_t a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop; /* mem_opt_t: -A -B -O -E -L -w -d */
                 int32_t threads;   /* [16] host threads of the passes around the kernels (cs_dedup_regions) */
                 uint32_t flags;    /* [0] CS_ALN_* below: A/B switches of cs_extend_chains, every combination gives the same regions */ } cs_aln_params_t;
#define CS_ALN_NO_LIGHT_PATHS 1u   /* every chain through the wave-per-chain region kernel, every read through the LDS purge kernel (default: chains of up to 8 seeds a lane
                                    * each, reads of up to 64 regions a wave each in registers) */
#define CS_ALN_PURGE_FROM_HBM 2u   /* reads of more than 64 regions purged straight from HBM: the fallback for a read whose regions do not fit the LDS (3,000) */
typedef struct { int64_t rb, re; int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0; float frac_rep; int32_t chain; } cs_alnreg_t;
typedef struct { int64_t n_reads; uint64_t n_regs; const uint64_t *reg_off; const cs_alnreg_t *regs; } cs_aln_result_t;
typedef struct { uint64_t reads, regions, pairs, retries, purged, launches;   /* pairs = extensions run incl. retries */
                 uint64_t ext_cells; double ext_kernel_ms; } cs_aln_stats_t;     /* of the extension kernels: DP cells inside the adaptive band, HIP-event time */
typedef struct cs_aligner cs_aligner_t;
void cs_aln_params_default(cs_aln_params_t *p);
int  cs_aligner_create(const char *prefix, int device /* -1: host-side passes only (cs_dedup_regions), no GPU needed */, const cs_aln_params_t *par /* NULL: defaults */, cs_aligner_t **out);
void cs_aligner_destroy(cs_aligner_t *a);
int  cs_extend_chains(cs_aligner_t *a, const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases,
                      const uint64_t *read_offsets, cs_aln_result_t *out);
/* The pass behind the extension stage (comp_seed.cpp:2385-2395), host code: regions marked by the purge (qe <= qb) are dropped and
 * mem_sort_dedup_patch (comp_seed.cpp:629-687) runs over each read's rest -- of two regions that overlap by more than mask_level_redun on
 * read and reference the lower-scoring one goes; colinear neighbours close to one diagonal are merged when a banded global alignment over
 * both (bwa_gen_cigar2 / ksw_global2) scores at least 0.9 of what they promise; the survivors come sorted by score.  `regs`: cs_extend_chains'
 * result (or the caller's regions in that form); `n_comp`: per surviving region, mem_alnreg_t.n_comp (1, more after merges; 0 for a read's only region, which the reference's function does not touch).  `chain` of a
 * merged region is that of its later part.  The result belongs to the aligner and stays valid until its next cs_dedup_regions. */
typedef struct { int32_t max_chain_gap; float mask_level_redun; } cs_dedup_params_t;   /* mem_opt_t: max_chain_gap 10000, mask_level_redun 0.95 */
void cs_dedup_params_default(cs_dedup_params_t *p);
int  cs_dedup_regions(cs_aligner_t *a, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets,
                      cs_aln_result_t *out, const int32_t **n_comp);
int  cs_aligner_stats(const cs_aligner_t *a, cs_aln_stats_t *st);

/* ---- the result of the LAST device-variant call, without moving it: an order-sensitive 64-bit digest per array
 *      (sum over the array's 64-bit words w[i] of splitmix64(w[i] + i * 0x9E3779B97F4A7C15), mod 2^64), so that two runs over
 *      10 M reads can be compared word for word without downloading 6 GB; and the CSR slice of selected reads (any order,
 *      repeats allowed) gathered on the device into the engine's pinned host buffers (what cs_engine_seed_batch would
 *      have returned for exactly those reads -- results do not depend on the batch a read travels in). */
typedef struct { uint64_t mem_off, mems, seed_off, seeds; } cs_digest_t;
int  cs_engine_result_digest(cs_engine_t *e, cs_digest_t *out);
int  cs_engine_gather_reads(cs_engine_t *e, int64_t n_sel, const uint64_t *read_ids, cs_result_t *out);

/* ---- byte model of the SMEM stage as THIS implementation runs it (bench.py's roofline): every access the kernels make to an
 *      index-side array is counted on the device as an event of its kernel; bytes = events x event_bytes.  These are the bytes
 *      the kernels ask for -- a lower bound of what has to come out of HBM/L2 for them -- not the bytes of the reference's
 *      algorithm (most of whose bwt_extend calls are answered without the FM index here).  Accumulated like cs_stats_t. */
#define CS_N_KERNELS 9   /* fwd0, fwd, bwd_win, bwd_win0, bwd_wide, bwd_all, r2text, r3text, fused (smem_kernel) */
#define CS_N_EVENTS 10   /* Occ record 32 B, jump entry 16 B, filter word 8 B, SA entry, inverse-SA entry (4 or 8 B), text word 4 B,
                            rep[] load 8 B, lcp[] byte, LEP entry 16 B (read or written), mem record read back 32 B */
typedef struct {
	uint64_t events[CS_N_KERNELS][CS_N_EVENTS];
	uint64_t event_bytes[CS_N_EVENTS];
	uint64_t stream_bytes;   /* modelled on the host per call: read bases read once, their packed records written and read by
	                            each kernel family, task queue words, backward task records, raw mems written, sorted and
	                            written again */
} cs_traffic_t;
int  cs_engine_traffic_model(cs_engine_t *e, cs_traffic_t *out);

/* ---- validation of the resident index at the size it is used (the byte-for-byte comparisons with bwaidx's files stop at 64 Mbp): counts
 *      violations of (1) recovered text == `d_fwd_nt4` (device memory, codes 0..3, l_pac bases; NULL skips this) and its reverse complement,
 *      (2) suffix order of every neighbouring pair of rows of the full suffix array, decided on the text, (3) ISA[SA[r]] == r, (4) BWT character
 *      of row r == T[SA[r] - 1] and the row of suffix 0 == primary, (5) sampled SA == full SA at the sampled rows (bwt_cal_sa, bwt.c:62-96).
 *      All zero <=> the .bwt / .sa pair is the FM index of that text.  undecided_rows: pairs that share more than 2^20 bases. */
typedef struct {
	uint64_t rows_checked, order_violations, isa_violations, bwt_violations, sampled_sa_violations, undecided_rows, text_violations;
	int32_t  text_checked, reserved;
} cs_index_check_t;
int  cs_engine_check_index(cs_engine_t *e, const uint8_t *d_fwd_nt4, uint64_t l_pac, cs_index_check_t *out);

int  cs_engine_stats(const cs_engine_t *e, cs_stats_t *st);
void cs_engine_reset_stats(cs_engine_t *e);

/* ---- batched primitives on the device index, for parity tests of the building blocks (host pointers):
 *      bwt_occ4 (bwt.c:169), bwt_extend (bwt.c:262; ok is n x 4 intervals, info untouched = 0), bwt_sa (bwt.c:86) */
int  cs_engine_occ4(cs_engine_t *e, int64_t n, const uint64_t *k, uint64_t *cnt4);
int  cs_engine_extend(cs_engine_t *e, int64_t n, const cs_intv_t *ik, const uint8_t *is_back, cs_intv_t *ok4);
int  cs_engine_sa(cs_engine_t *e, int64_t n, const uint64_t *k, uint64_t *sa);

/* ---- diagnostic: ceiling of the path's access shape on the resident index -- every lane of `waves_per_simd` x 4 waves
 *      per CU follows a dependent chain of `steps` random 64-byte Occ-block reads; returns 64-byte lines per second */
int  cs_engine_probe_random_lines(cs_engine_t *e, int waves_per_simd, int steps, double *lines_per_sec);

/* ---- device memory helpers so that a caller without its own HIP code can stage inputs for the device variant */
int  cs_device_alloc(cs_engine_t *e, size_t bytes, void **dptr);
int  cs_device_free(cs_engine_t *e, void *dptr);
int  cs_device_upload(cs_engine_t *e, void *dst, const void *src, size_t bytes);
int  cs_device_download(cs_engine_t *e, void *dst, const void *src, size_t bytes);
int  cs_device_sync(cs_engine_t *e);

#ifdef __cplusplus
}
#endif
#endif /* COMPSEED_AMD_H */
