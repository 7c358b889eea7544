/* oracle/cs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's SMEM seeding path (i-xiaohu/CompSeed), written from the
 * behavioural spec in SURVEY.md Appendix A and the cited reference lines.  It is the CHECKER for the HIP
 * engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * library (compseed_amd/csrc) never links, loads or calls anything in this directory.
 *
 * Pinned (tests/test_oracle.py) against the golden vectors in tests/golden/, which were produced by the
 * real reference compiled in place (oracle/Makefile `ref`, oracle/ref_harness.cpp).
 */
#ifndef CS_ORACLE_H
#define CS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bi-interval, same field meaning as bwtintv_t (FM_index/bwt.h:62-64) */
typedef struct { uint64_t x0, x1, x2, info; } cso_intv_t;

/* seed, the three fields of mem_seed_t that seeding fills (mapping/comp_seed.h:77-83) */
typedef struct { int64_t rbeg; int32_t qbeg, len; } cso_seed_t;

/* index handle, the seeding-relevant part of bwt_t (FM_index/bwt.h:48-60) */
typedef struct {
	uint64_t primary, L2[5], seq_len, bwt_size; /* bwt_size in 32-bit words */
	const uint32_t *bwt;                        /* 64-B blocks: 4 x u64 counts + 128 bases, bwt.h:73-80 */
	uint64_t sa_intv, n_sa;
	const uint64_t *sa;                         /* sa[0] == (uint64_t)-1, bwt.c:83 */
	void *owned_bwt, *owned_sa;
} cso_index_t;

/* the mem_opt_t fields read by seeding (mapping/comp_seed.h:50-59; defaults comp_seed.cpp:26-58) */
typedef struct {
	int32_t  min_seed_len;   /* -k, 19  */
	float    split_factor;   /* -r, 1.5 */
	int32_t  split_width;    /* -s, 10  */
	int32_t  max_occ;        /* -c, 500 */
	uint64_t max_mem_intv;   /* -y, 20  */
} cso_params_t;

typedef struct {
	uint64_t bwt_queries;     /* == CompSeed "BWT-extend queries" == bwamem "BWT-extend calls" (main.cpp:206-209) */
	uint64_t bwt_calls;       /* real bwt_extend calls under the SST policy of the chosen batch size            */
	uint64_t bwt_blocks;      /* 64-B Occ blocks touched by those real calls (1 or 2 each, bwt.c:194)            */
	uint64_t bwt_blocks_uncached; /* the same for every query (no cache)                                         */
	uint64_t sal_queries;     /* SA slots requested (comp_seed.cpp:2322)                                         */
	uint64_t sal_calls;       /* distinct slots per batch (comp_seed.cpp:2339-2341)                              */
	uint64_t sal_steps;       /* bwt_invPsi steps walked by the distinct slots                                    */
	uint64_t sal_steps_uncached; /* ... by every requested slot                                                   */
	uint64_t n_mems, n_seeds;
	/* workload shape (mode 0 only), used to size the device kernels: */
	uint64_t q_fwd, q_bwd, q_r3;   /* bwt_extend queries by phase: forward passes of rounds 1-2, backward sweeps, round 3 */
	uint64_t n_calls;              /* SMEM calls (rounds 1-2)                                                         */
	uint64_t lep_sum, lep_max;     /* LEP list length at the end of the forward pass                                  */
	uint64_t bwd_steps;            /* backward positions visited                                                      */
	uint64_t lep_hist[8];          /* calls with list length <=8, <=16, <=24, <=32, <=48, <=64, <=128, >128            */
	uint64_t q_bwd_hist[8];        /* backward queries spent in calls of those classes                                */
} cso_stats_t;

void cso_params_default(cso_params_t *p);

/* index: read <prefix>.bwt and <prefix>.sa (formats bwt.c:385-462) or wrap caller-owned arrays */
int  cso_index_load(cso_index_t *idx, const char *prefix);
int  cso_index_wrap(cso_index_t *idx, uint64_t primary, const uint64_t L2_1to4[4], const uint32_t *bwt, uint64_t bwt_size,
                    const uint64_t *sa, uint64_t n_sa, uint64_t sa_intv);
void cso_index_free(cso_index_t *idx);
/* cs_index_naive.c: FM-index of a forward-strand genome (codes 0..3) by a plain comparison sort of all suffixes; the same
 * arrays bwaidx writes (index_main.c:257-325).  For CPU-side tests that generate their own genome.  0 on success. */
int  cso_index_build(const uint8_t *fwd_nt4, uint64_t l_pac, int n_threads, cso_index_t *out);

/* primitives (bwt.c:169, 189, 262, 107, 53, 86; bwt.h:82) */
void     cso_occ4(const cso_index_t *idx, uint64_t k, uint64_t cnt[4]);
int      cso_2occ4(const cso_index_t *idx, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4]); /* returns blocks touched */
int      cso_extend(const cso_index_t *idx, const cso_intv_t *ik, cso_intv_t ok[4], int is_back);  /* returns blocks touched */
void     cso_set_intv(const cso_index_t *idx, int c, cso_intv_t *ik);
uint64_t cso_occ(const cso_index_t *idx, uint64_t k, int c);
uint64_t cso_inv_psi(const cso_index_t *idx, uint64_t k);
uint64_t cso_sa(const cso_index_t *idx, uint64_t k, uint64_t *steps);

/* one SMEM round through `x` (bwt_smem1a with max_intv = 0, bwt.c:289-351): returns the next pivot, mems ascending.
 * `mem` must have room for len+1 entries; `scratch` for len+1 entries. */
int cso_smem1(const cso_index_t *idx, int len, const uint8_t *q, int x, uint64_t min_intv,
              cso_intv_t *mem, int *n_mem, cso_intv_t *scratch, cso_stats_t *st);
/* round-3 seed (bwt_seed_strategy1, bwt.c:358-379) */
int cso_seed_strategy1(const cso_index_t *idx, int len, const uint8_t *q, int x, int min_len, uint64_t max_intv,
                       cso_intv_t *mem, cso_stats_t *st);

/* Whole path for a batch of reads: 3 rounds + sort (comp_seed.cpp:2255-2302 / bwamem.c:218-272), then SAL
 * (comp_seed.cpp:2306-2347) when want_sal.  `bases` holds the reads back to back, ASCII or nt4 codes; read r is
 * bases[offsets[r] .. offsets[r+1]).  mode 0 = uncached BWA-MEM flow; mode 1 = CompSeed flow with an emulated SST
 * that is reset every `sst_batch` reads (512 = BATCH_SIZE, comp_seed.h:36) -- results are identical, only
 * st->bwt_calls / bwt_blocks differ.  Outputs are malloc'ed CSR arrays the caller frees with cso_free. */
int cso_seed_batch(const cso_index_t *idx, const cso_params_t *par, int64_t n_reads, const uint8_t *bases,
                   const uint64_t *offsets, int mode, int sst_batch, int want_sal, int n_threads,
                   uint64_t **mem_off, cso_intv_t **mems, uint64_t **seed_off, cso_seed_t **seeds, cso_stats_t *st);
void cso_free(void *p);

/* ---- banded Smith-Waterman seed extension (cs_bsw_oracle.c): ksw_extend2 (bwalib/ksw.c:380) restated.  Sequences are codes 0..4. */
typedef struct { int8_t mat[25]; int32_t o_del, e_del, o_ins, e_ins, zdrop, end_bonus; } cso_bsw_params_t;
typedef struct { uint64_t q_off, t_off; int32_t qlen, tlen, h0, pad; } cso_bsw_pair_t;
typedef struct { int32_t score, qle, tle, gtle, gscore, max_off; } cso_bsw_result_t;
void cso_bsw_params_default(cso_bsw_params_t *P);
int  cso_bsw_score(const cso_bsw_params_t *P, int vec_rule, int t, int q);
int  cso_bsw_uses_vec_rule(const cso_bsw_params_t *P, int qlen, int tlen, int h0);
/* vec_rule 0: ksw_extend2 / scalarBandedSWA (the matrix); 1: getScores8 / getScores16 (compare the codes); cso_extend_pair picks as the reference does */
int  cso_extend_pair_rule(const cso_bsw_params_t *P, int vec_rule, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int h0, cso_bsw_result_t *out);
int  cso_extend_pair(const cso_bsw_params_t *P, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int h0, cso_bsw_result_t *out);
int  cso_extend_batch(const cso_bsw_params_t *P, int64_t n, const cso_bsw_pair_t *pairs, const uint8_t *qbuf, const uint8_t *tbuf, int w,
                      int n_threads, cso_bsw_result_t *out);

extern const uint8_t cso_nt4_table[256]; /* ASCII -> 0..4 ('-' -> 5), same mapping as nst_nt4_table (bntseq.c:46) */

#ifdef __cplusplus
}
#endif
#endif
