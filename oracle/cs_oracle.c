/* oracle/cs_oracle.c -- TEST INFRASTRUCTURE ONLY (see cs_oracle.h).
 *
 * CPU restatement of the reference hot path.  Every function names the reference lines it follows; the code
 * itself is written from the behavioural spec, not transcribed: counting uses 2-bit
 * lane compares + popcount instead of the reference's byte LUT, the backward sweep compacts one list in place
 * instead of swapping two vectors, and the SST is a flat node array.  Results are integers and must be
 * bit-identical to the reference; tests/test_oracle.py checks that against tests/golden/.
 */
#define _GNU_SOURCE
#include "cs_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define NONE64 ((uint64_t)-1)

const uint8_t cso_nt4_table[256] = {
	[0 ... 255] = 4,
	['A'] = 0, ['a'] = 0, ['C'] = 1, ['c'] = 1, ['G'] = 2, ['g'] = 2, ['T'] = 3, ['t'] = 3, ['-'] = 5,
};

void cso_params_default(cso_params_t *p) /* mem_opt_init, mapping/comp_seed.cpp:26-58 */
{
	p->min_seed_len = 19; p->split_factor = 1.5f; p->split_width = 10; p->max_occ = 500; p->max_mem_intv = 20;
}

/* ------------------------------------------------------------------ index IO (FM_index/bwt.c:385-462) */

static void *slurp(const char *fn, size_t skip, size_t *n_bytes)
{
	FILE *fp = fopen(fn, "rb");
	if (!fp) return NULL;
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	if (sz < (long)skip) { fclose(fp); return NULL; }
	fseek(fp, 0, SEEK_SET);
	uint8_t *buf = (uint8_t *)malloc((size_t)sz + 64);
	if (!buf || fread(buf, 1, (size_t)sz, fp) != (size_t)sz) { free(buf); fclose(fp); return NULL; }
	fclose(fp);
	*n_bytes = (size_t)sz;
	return buf;
}

int cso_index_load(cso_index_t *idx, const char *prefix)
{
	char fn[4096];
	size_t nb = 0;
	memset(idx, 0, sizeof(*idx));
	/* .bwt = primary, L2[1..4], then the interleaved words (bwt_dump_bwt, bwt.c:385-394) */
	snprintf(fn, sizeof fn, "%s.bwt", prefix);
	uint8_t *b = (uint8_t *)slurp(fn, 40, &nb);
	if (!b) return -1;
	memcpy(&idx->primary, b, 8);
	memcpy(&idx->L2[1], b + 8, 32);
	idx->L2[0] = 0;
	idx->seq_len = idx->L2[4];
	idx->bwt_size = (nb - 40) >> 2;
	idx->bwt = (const uint32_t *)(b + 40);
	idx->owned_bwt = b;
	/* .sa = primary, L2[1..4], sa_intv, seq_len, then sa[1..n_sa-1] (bwt_dump_sa, bwt.c:396-407) */
	snprintf(fn, sizeof fn, "%s.sa", prefix);
	uint8_t *s = (uint8_t *)slurp(fn, 56, &nb);
	if (!s) { cso_index_free(idx); return -2; }
	uint64_t h[7];
	memcpy(h, s, 56);
	if (h[0] != idx->primary || h[6] != idx->seq_len) { free(s); cso_index_free(idx); return -3; } /* bwt.c:429,433 */
	idx->sa_intv = h[5];
	idx->n_sa = (idx->seq_len + idx->sa_intv) / idx->sa_intv;
	if ((nb - 56) / 8 < idx->n_sa - 1) { free(s); cso_index_free(idx); return -4; }
	/* the file has no sa[0]; slot 6 of the header (seq_len) sits exactly where sa[0] belongs, overwrite it with -1 */
	uint64_t *sa = (uint64_t *)(s + 48);
	sa[0] = NONE64;
	idx->sa = sa;
	idx->owned_sa = s;
	return 0;
}

int cso_index_wrap(cso_index_t *idx, uint64_t primary, const uint64_t L2_1to4[4], const uint32_t *bwt, uint64_t bwt_size,
                   const uint64_t *sa, uint64_t n_sa, uint64_t sa_intv)
{
	memset(idx, 0, sizeof(*idx));
	idx->primary = primary;
	memcpy(&idx->L2[1], L2_1to4, 32);
	idx->seq_len = idx->L2[4];
	idx->bwt = bwt; idx->bwt_size = bwt_size;
	idx->sa = sa; idx->n_sa = n_sa; idx->sa_intv = sa_intv;
	return 0;
}

void cso_index_free(cso_index_t *idx)
{
	free(idx->owned_bwt); free(idx->owned_sa);
	memset(idx, 0, sizeof(*idx));
}

/* ------------------------------------------------------------------ primitives */

/* number of 2-bit lanes of w equal to c among the first nb bases (base 0 = the two top bits, bwt.h:80) */
static inline uint32_t lanes_eq(uint32_t w, uint32_t c, uint32_t nb)
{
	uint32_t x = w ^ (c * 0x55555555u);          /* matching lanes become 00 */
	uint32_t m = ~(x | (x >> 1)) & 0x55555555u; /* low bit of every matching lane */
	if (nb < 16) m &= ~((1u << ((16 - nb) << 1)) - 1u);
	return (uint32_t)__builtin_popcount(m);
}

/* bwt_occ4, FM_index/bwt.c:169-186: counts of A,C,G,T in BWT rows [0, k] ($ skipped) */
void cso_occ4(const cso_index_t *idx, uint64_t k, uint64_t cnt[4])
{
	if (k == NONE64) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= idx->primary);
	const uint32_t *blk = idx->bwt + ((k >> 7) << 4);
	memcpy(cnt, blk, 32);
	uint32_t need = (uint32_t)(k & 127) + 1; /* bases 0..k&127 inclusive */
	for (int w = 0; need > 0 && w < 8; ++w) {
		uint32_t nb = need < 16 ? need : 16, word = blk[8 + w];
		uint32_t c1 = lanes_eq(word, 1, nb), c2 = lanes_eq(word, 2, nb), c3 = lanes_eq(word, 3, nb);
		cnt[0] += nb - c1 - c2 - c3; cnt[1] += c1; cnt[2] += c2; cnt[3] += c3;
		need -= nb;
	}
}

/* bwt_2occ4, bwt.c:189-220.  The reference shares one block read when both rows fall in the same 128-base
 * block and neither is -1 (bwt.c:194); the counts are the same either way, so only that line count is kept. */
int cso_2occ4(const cso_index_t *idx, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	uint64_t ak = k - (k >= idx->primary), al = l - (l >= idx->primary);
	cso_occ4(idx, k, ck);
	cso_occ4(idx, l, cl);
	if (k == NONE64 && l == NONE64) return 0;
	if (k == NONE64 || l == NONE64) return 1;
	return (ak >> 7) != (al >> 7) ? 2 : 1;
}

/* bwt_extend, bwt.c:262-275 */
int cso_extend(const cso_index_t *idx, const cso_intv_t *ik, cso_intv_t ok[4], int is_back)
{
	uint64_t tk[4], tl[4];
	const uint64_t *in = &ik->x0;
	int a = !is_back, b = is_back; /* a: the coordinate that is searched, b: the one that is shifted */
	int blocks = cso_2occ4(idx, in[a] - 1, in[a] - 1 + ik->x2, tk, tl);
	uint64_t *o[4] = {&ok[0].x0, &ok[1].x0, &ok[2].x0, &ok[3].x0};
	for (int c = 0; c < 4; ++c) {
		o[c][a] = idx->L2[c] + 1 + tk[c];
		o[c][2] = tl[c] - tk[c];
	}
	o[3][b] = in[b] + (in[a] <= idx->primary && in[a] + ik->x2 - 1 >= idx->primary);
	o[2][b] = o[3][b] + o[3][2];
	o[1][b] = o[2][b] + o[2][2];
	o[0][b] = o[1][b] + o[1][2];
	return blocks;
}

void cso_set_intv(const cso_index_t *idx, int c, cso_intv_t *ik) /* bwt_set_intv, bwt.h:82 */
{
	ik->x0 = idx->L2[c] + 1;
	ik->x2 = idx->L2[c + 1] - idx->L2[c];
	ik->x1 = idx->L2[3 - c] + 1;
	ik->info = 0;
}

uint64_t cso_occ(const cso_index_t *idx, uint64_t k, int c) /* bwt_occ, bwt.c:107-129 */
{
	uint64_t cnt[4];
	if (k == idx->seq_len) return idx->L2[c + 1] - idx->L2[c];
	cso_occ4(idx, k, cnt);
	return cnt[c];
}

uint64_t cso_inv_psi(const cso_index_t *idx, uint64_t k) /* bwt_invPsi, bwt.c:53-59 */
{
	if (k == idx->primary) return 0;
	uint64_t x = k - (k > idx->primary);
	uint32_t word = idx->bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)]; /* bwt_bwt / bwt_B0, bwt.h:74-80 */
	int c = (word >> ((~x & 15) << 1)) & 3;
	return idx->L2[c] + cso_occ(idx, k, c);
}

uint64_t cso_sa(const cso_index_t *idx, uint64_t k, uint64_t *steps) /* bwt_sa, bwt.c:86-96 */
{
	uint64_t s = 0, mask = idx->sa_intv - 1;
	while (k & mask) { ++s; k = cso_inv_psi(idx, k); }
	if (steps) *steps = s;
	return s + idx->sa[k / idx->sa_intv];
}

/* ------------------------------------------------------------------ uncached rounds (BWA-MEM flow) */

/* bwt_smem1a with max_intv = 0, bwt.c:289-351 (== smem1_profile, mapping/bwamem.c:102-168).
 * lep[] collects, in increasing end order, the forward intervals at which the occurrence count changes; the
 * backward sweep then walks the live part lep[lo..n) from the longest match down and compacts it in place. */
int cso_smem1(const cso_index_t *idx, int len, const uint8_t *q, int x, uint64_t min_intv,
              cso_intv_t *mem, int *n_mem, cso_intv_t *lep, cso_stats_t *st)
{
	cso_intv_t ik, ok[4];
	int i, n = 0, nm = 0;
	*n_mem = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	cso_set_intv(idx, q[x], &ik);
	ik.info = (uint64_t)x + 1;
	for (i = x + 1; i < len; ++i) {
		if (q[i] > 3) { lep[n++] = ik; break; }                   /* ambiguous base ends the match */
		int c = 3 - q[i];
		int nb = cso_extend(idx, &ik, ok, 0);
		if (st) { st->bwt_queries++; st->bwt_blocks_uncached += nb; st->q_fwd++; }
		if (ok[c].x2 != ik.x2) {
			lep[n++] = ik;
			if (ok[c].x2 < min_intv) break;
		}
		ik = ok[c]; ik.info = (uint64_t)i + 1;
	}
	if (i == len) lep[n++] = ik;
	int ret = (int)lep[n - 1].info; /* end of the longest forward match */
	int hcls = n <= 8 ? 0 : n <= 16 ? 1 : n <= 24 ? 2 : n <= 32 ? 3 : n <= 48 ? 4 : n <= 64 ? 5 : n <= 128 ? 6 : 7;
	if (st) { st->n_calls++; st->lep_sum += (uint64_t)n; if ((uint64_t)n > st->lep_max) st->lep_max = (uint64_t)n; st->lep_hist[hcls]++; }

	int lo = 0;
	for (i = x - 1; i >= -1; --i) {
		int c = (i < 0 || q[i] > 3) ? -1 : q[i];
		int w = n, kept = 0;
		uint64_t last_kept = 0;
		if (st) st->bwd_steps++;
		for (int j = n - 1; j >= lo; --j) {
			cso_intv_t p = lep[j];
			if (c >= 0) {
				int nb = cso_extend(idx, &p, ok, 1);
				if (st) { st->bwt_queries++; st->bwt_blocks_uncached += nb; st->q_bwd++; st->q_bwd_hist[hcls]++; }
			}
			if (c < 0 || ok[c].x2 < min_intv) {
				/* cannot grow: an SMEM iff no longer match survived this step and it is not contained */
				if (kept == 0 && (nm == 0 || (uint64_t)(i + 1) < (mem[nm - 1].info >> 32))) {
					mem[nm] = p;
					mem[nm++].info |= (uint64_t)(i + 1) << 32;
				}
			} else if (kept == 0 || ok[c].x2 != last_kept) {
				ok[c].info = p.info;
				last_kept = ok[c].x2;
				lep[--w] = ok[c];
				++kept;
			}
		}
		if (kept == 0) break;
		lo = w;
	}
	for (int a = 0, b = nm - 1; a < b; ++a, --b) { cso_intv_t t = mem[a]; mem[a] = mem[b]; mem[b] = t; } /* bwt.c:346 */
	*n_mem = nm;
	return ret;
}

/* bwt_seed_strategy1, bwt.c:358-379 (== seed_strategy_profile, bwamem.c:170-192) */
int cso_seed_strategy1(const cso_index_t *idx, int len, const uint8_t *q, int x, int min_len, uint64_t max_intv,
                       cso_intv_t *mem, cso_stats_t *st)
{
	cso_intv_t ik, ok[4];
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	cso_set_intv(idx, q[x], &ik);
	for (int i = x + 1; i < len; ++i) {
		if (q[i] > 3) return i + 1;
		int c = 3 - q[i];
		int nb = cso_extend(idx, &ik, ok, 0);
		if (st) { st->bwt_queries++; st->bwt_blocks_uncached += nb; st->q_r3++; }
		if (ok[c].x2 < max_intv && i - x >= min_len) {
			*mem = ok[c];
			mem->info = (uint64_t)x << 32 | (uint64_t)(i + 1);
			return i + 1;
		}
		ik = ok[c];
	}
	return len;
}

/* ------------------------------------------------------------------ SST emulation (mapping/SST.h, SST.cpp) */

typedef struct { uint64_t x0, x1, x2; int32_t ch[4]; } sst_node_t; /* SST_Node_t, SST.h:12-16 */
typedef struct {
	sst_node_t *a; size_t n, m;
	const cso_index_t *idx;
	uint64_t calls, blocks;
} sst_t;

static int sst_push(sst_t *t, uint64_t x0, uint64_t x1, uint64_t x2)
{
	if (t->n == t->m) { t->m = t->m ? t->m << 1 : 1024; t->a = (sst_node_t *)realloc(t->a, t->m * sizeof(sst_node_t)); }
	sst_node_t *nd = &t->a[t->n];
	nd->x0 = x0; nd->x1 = x1; nd->x2 = x2;
	nd->ch[0] = nd->ch[1] = nd->ch[2] = nd->ch[3] = -1;
	return (int)t->n++;
}

static void sst_init(sst_t *t, const cso_index_t *idx) /* SST::SST, SST.cpp:7-18 */
{
	memset(t, 0, sizeof(*t));
	t->idx = idx;
	sst_push(t, 0, 0, 0);
	for (int c = 0; c < 4; ++c) {
		cso_intv_t v; cso_set_intv(idx, c, &v);
		t->a[0].ch[c] = sst_push(t, v.x0, v.x1, v.x2);
	}
}

static void sst_clear(sst_t *t) /* SST::clear, SST.h:49-57 */
{
	t->n = 5;
	for (int i = 1; i <= 4; ++i) t->a[i].ch[0] = t->a[i].ch[1] = t->a[i].ch[2] = t->a[i].ch[3] = -1;
}

static void sst_real_extend(sst_t *t, int parent, cso_intv_t ok[4], int is_back)
{
	cso_intv_t ik = {t->a[parent].x0, t->a[parent].x1, t->a[parent].x2, 0};
	t->blocks += (uint64_t)cso_extend(t->idx, &ik, ok, is_back);
	t->calls++;
}

static int sst_child(sst_t *t, int parent, int base, int is_back) /* query_forward_child / query_backward_child, SST.h:60-92 */
{
	cso_intv_t ok[4];
	if (t->a[parent].ch[base] == -1) {
		sst_real_extend(t, parent, ok, is_back);
		int id = sst_push(t, ok[base].x0, ok[base].x1, ok[base].x2);
		t->a[parent].ch[base] = id;
	}
	int id = t->a[parent].ch[base];
	if (is_back && !(t->a[id].x0 | t->a[id].x1 | t->a[id].x2)) { /* an "empty" placeholder is filled on demand, SST.h:83-91 */
		sst_real_extend(t, parent, ok, 1);
		t->a[id].x0 = ok[base].x0; t->a[id].x1 = ok[base].x1; t->a[id].x2 = ok[base].x2;
	}
	return id;
}

static int sst_add_empty(sst_t *t, int parent, int base) /* SST.h:111-119 */
{
	if (t->a[parent].ch[base] == -1) { int id = sst_push(t, 0, 0, 0); t->a[parent].ch[base] = id; }
	return t->a[parent].ch[base];
}

static int sst_add_lep(sst_t *t, int parent, int base, const cso_intv_t *p) /* SST.h:94-109 */
{
	if (t->a[parent].ch[base] == -1) { int id = sst_push(t, p->x0, p->x1, p->x2); t->a[parent].ch[base] = id; }
	else { sst_node_t *c = &t->a[t->a[parent].ch[base]]; c->x0 = p->x0; c->x1 = p->x1; c->x2 = p->x2; }
	return t->a[parent].ch[base];
}

/* ------------------------------------------------------------------ cached rounds (CompSeed flow) */

typedef struct {
	sst_t fwd, bwd;
	cso_intv_t *lep, *mem; int cap; /* per-read scratch, len+1 entries each */
	cso_stats_t st;
} worker_t;

/* collect_mem_with_sst, mapping/comp_seed.cpp:67-139.  mems come out in DESCENDING start order (no reversal). */
static int collect_cached(worker_t *w, const uint8_t *seq, int len, int pivot, uint64_t min_hits, int *n_mem)
{
	cso_intv_t *lep = w->lep, *mem = w->mem;
	int n = 0, nm = 0, i, ret = len;
	*n_mem = 0;
	if (seq[pivot] > 3) return pivot + 1;
	int node = sst_child(&w->fwd, 0, seq[pivot], 0);
	cso_intv_t ik = {w->fwd.a[node].x0, w->fwd.a[node].x1, w->fwd.a[node].x2, (uint64_t)pivot + 1};
	for (i = pivot + 1; i < len; ++i) {
		if (seq[i] > 3) { lep[n++] = ik; ret = i + 1; break; }
		node = sst_child(&w->fwd, node, 3 - seq[i], 0);
		const sst_node_t *nd = &w->fwd.a[node];
		w->st.bwt_queries++;
		if (nd->x2 != ik.x2) {
			lep[n++] = ik;
			if (nd->x2 < min_hits) { ret = i; break; }
		}
		ik.x0 = nd->x0; ik.x1 = nd->x1; ik.x2 = nd->x2; ik.info = (uint64_t)i + 1;
	}
	if (ret == len) lep[n++] = ik;
	if (pivot == 0) { mem[0] = lep[n - 1]; *n_mem = 1; return ret; } /* comp_seed.cpp:98-101 */

	/* register every LEP in the backward trie, longest first (comp_seed.cpp:102-112); the node id rides in info>>32 */
	for (int j = n - 1; j >= 0; --j) {
		int id = 0;
		for (int t = (int)lep[j].info - 1; t >= pivot + 1; --t) id = sst_add_empty(&w->bwd, id, seq[t]);
		id = sst_add_lep(&w->bwd, id, seq[pivot], &lep[j]);
		lep[j].info |= (uint64_t)id << 32;
	}
	int lo = 0;
	for (i = pivot - 1; i >= -1; --i) {
		int c = (i == -1) ? 4 : seq[i];
		int wr = n, kept = 0;
		uint64_t last_kept = 0;
		for (int j = n - 1; j >= lo; --j) {
			cso_intv_t p = lep[j], nx = {0, 0, 0, 0};
			int id = (int)(p.info >> 32);
			if (c < 4) {
				id = sst_child(&w->bwd, id, c, 1);
				const sst_node_t *nd = &w->bwd.a[id];
				nx.x0 = nd->x0; nx.x1 = nd->x1; nx.x2 = nd->x2;
				w->st.bwt_queries++;
			}
			if (c > 3 || nx.x2 < min_hits) {
				if (nm == 0 || (uint64_t)(i + 1) < (mem[nm - 1].info >> 32)) { /* comp_seed.cpp:125-129 */
					mem[nm] = p;
					mem[nm++].info = (uint64_t)(i + 1) << 32 | (uint32_t)p.info;
				}
			} else if (kept == 0 || nx.x2 != last_kept) {
				nx.info = (uint64_t)id << 32 | (uint32_t)p.info;
				last_kept = nx.x2;
				lep[--wr] = nx;
				++kept;
			}
		}
		if (kept == 0) break;
		lo = wr;
	}
	*n_mem = nm;
	return ret;
}

/* tem_forward_sst, comp_seed.cpp:141-160 */
static int tem_forward_cached(worker_t *w, const cso_params_t *par, const uint8_t *seq, int len, int start, cso_intv_t *mem)
{
	memset(mem, 0, sizeof(*mem));
	if (seq[start] > 3) return start + 1;
	int node = sst_child(&w->fwd, 0, seq[start], 0);
	for (int i = start + 1; i < len; ++i) {
		if (seq[i] > 3) return i + 1;
		node = sst_child(&w->fwd, node, 3 - seq[i], 0);
		const sst_node_t *nd = &w->fwd.a[node];
		w->st.bwt_queries++;
		if (nd->x2 < par->max_mem_intv && i - start >= par->min_seed_len) {
			mem->x0 = nd->x0; mem->x1 = nd->x1; mem->x2 = nd->x2;
			mem->info = (uint64_t)start << 32 | (uint64_t)(i + 1);
			return i + 1;
		}
	}
	return len;
}

/* ------------------------------------------------------------------ per-read driver, batches, SAL */

typedef struct { cso_intv_t *a; size_t n, m; } ivec_t;
static void ivec_push(ivec_t *v, const cso_intv_t *x)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 64; v->a = (cso_intv_t *)realloc(v->a, v->m * sizeof(cso_intv_t)); }
	v->a[v->n++] = *x;
}
static int by_info(const void *a, const void *b)
{
	uint64_t x = ((const cso_intv_t *)a)->info, y = ((const cso_intv_t *)b)->info;
	return x < y ? -1 : x > y;
}
static inline int mem_span(const cso_intv_t *m) { return (int)(uint32_t)m->info - (int)(m->info >> 32); }

/* three rounds + sort for one read: comp_seed.cpp:2262-2301 (mode 1) or bwamem.c:218-272 (mode 0).
 * The re-seeding threshold is CompSeed's double-precision form (comp_seed.cpp:2279). */
static void seed_one_read(worker_t *w, const cso_index_t *idx, const cso_params_t *par, int mode,
                          const uint8_t *seq, int len, ivec_t *out)
{
	size_t first = out->n;
	int nm, split_len = (int)(1.0 * par->min_seed_len * par->split_factor + .499);
	for (int x = 0; x < len; ) { /* round 1 */
		if (mode == 0 && seq[x] > 3) { ++x; continue; }
		x = mode ? collect_cached(w, seq, len, x, 1, &nm) : cso_smem1(idx, len, seq, x, 1, w->mem, &nm, w->lep, &w->st);
		for (int i = 0; i < nm; ++i) if (mem_span(&w->mem[i]) >= par->min_seed_len) ivec_push(out, &w->mem[i]);
	}
	size_t old_n = out->n; /* round 2 */
	for (size_t k = first; k < old_n; ++k) {
		cso_intv_t p = out->a[k];
		int beg = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
		if (end - beg < split_len || p.x2 > (uint64_t)par->split_width) continue;
		if (mode) collect_cached(w, seq, len, (beg + end) / 2, p.x2 + 1, &nm);
		else cso_smem1(idx, len, seq, (beg + end) >> 1, p.x2 + 1, w->mem, &nm, w->lep, &w->st);
		for (int i = 0; i < nm; ++i) if (mem_span(&w->mem[i]) >= par->min_seed_len) ivec_push(out, &w->mem[i]);
	}
	if (par->max_mem_intv > 0) { /* round 3 */
		for (int x = 0; x < len; ) {
			if (seq[x] > 3) { ++x; continue; }
			cso_intv_t m;
			x = mode ? tem_forward_cached(w, par, seq, len, x, &m)
			         : cso_seed_strategy1(idx, len, seq, x, par->min_seed_len, par->max_mem_intv, &m, &w->st);
			if (m.x2 > 0) ivec_push(out, &m);
		}
	}
	qsort(out->a + first, out->n - first, sizeof(cso_intv_t), by_info); /* equal keys are equal elements */
}

typedef struct { uint64_t slot, coord, steps; } salreq_t;
static int by_slot(const void *a, const void *b)
{
	uint64_t x = ((const salreq_t *)a)->slot, y = ((const salreq_t *)b)->slot;
	return x < y ? -1 : x > y;
}

typedef struct {
	ivec_t mems; uint32_t *mem_cnt;
	cso_seed_t *seeds; size_t n_seeds; uint32_t *seed_cnt;
} batch_out_t;

typedef struct {
	const cso_index_t *idx; const cso_params_t *par;
	int64_t n_reads; const uint8_t *bases; const uint64_t *offsets;
	int mode, sst_batch, want_sal;
	int64_t n_batches; volatile int64_t next;
	batch_out_t *bout;
	pthread_mutex_t mu; cso_stats_t total;
} job_t;

/* SAL for one batch, comp_seed.cpp:2306-2347 */
static void sal_batch(worker_t *w, const job_t *J, batch_out_t *B, int64_t n)
{
	const cso_params_t *par = J->par;
	size_t cap = 0, ns = 0, pos = 0;
	for (int64_t r = 0; r < n; ++r)
		for (uint32_t i = 0; i < B->mem_cnt[r]; ++i, ++pos) {
			uint64_t x2 = B->mems.a[pos].x2;
			cap += x2 < (uint64_t)par->max_occ ? x2 : (uint64_t)par->max_occ;
		}
	B->seeds = (cso_seed_t *)malloc((cap + 1) * sizeof(cso_seed_t));
	B->seed_cnt = (uint32_t *)calloc((size_t)n + 1, sizeof(uint32_t));
	salreq_t *rq = (salreq_t *)malloc((cap + 1) * sizeof(salreq_t));
	pos = 0;
	for (int64_t r = 0; r < n; ++r)
		for (uint32_t i = 0; i < B->mem_cnt[r]; ++i, ++pos) {
			const cso_intv_t *m = &B->mems.a[pos];
			uint64_t step = m->x2 > (uint64_t)par->max_occ ? m->x2 / (uint64_t)par->max_occ : 1;
			for (uint64_t k = 0, cnt = 0; k < m->x2 && cnt < (uint64_t)par->max_occ; k += step, ++cnt) {
				cso_seed_t s = {(int64_t)(m->x0 + k), (int32_t)(m->info >> 32), mem_span(m)};
				rq[ns].slot = m->x0 + k; rq[ns].coord = NONE64; rq[ns].steps = 0;
				B->seeds[ns++] = s;
				B->seed_cnt[r]++;
			}
		}
	w->st.sal_queries += ns;
	qsort(rq, ns, sizeof(salreq_t), by_slot);
	size_t nu = 0;
	for (size_t i = 0; i < ns; ++i) if (i == 0 || rq[i].slot != rq[nu - 1].slot) rq[nu++] = rq[i];
	for (size_t i = 0; i < nu; ++i) {
		rq[i].coord = cso_sa(J->idx, rq[i].slot, &rq[i].steps);
		w->st.sal_calls++; w->st.sal_steps += rq[i].steps;
	}
	for (size_t i = 0; i < ns; ++i) {
		salreq_t key = {(uint64_t)B->seeds[i].rbeg, 0, 0};
		const salreq_t *hit = (const salreq_t *)bsearch(&key, rq, nu, sizeof(salreq_t), by_slot);
		B->seeds[i].rbeg = (int64_t)hit->coord;
		w->st.sal_steps_uncached += hit->steps;
	}
	B->n_seeds = ns;
	free(rq);
}

static void *worker_main(void *arg)
{
	job_t *J = (job_t *)arg;
	worker_t w;
	memset(&w, 0, sizeof w);
	sst_init(&w.fwd, J->idx); sst_init(&w.bwd, J->idx);
	uint8_t *seq = NULL; int seq_cap = 0;
	for (;;) {
		int64_t b = __sync_fetch_and_add(&J->next, 1);
		if (b >= J->n_batches) break;
		int64_t r0 = b * J->sst_batch, r1 = r0 + J->sst_batch < J->n_reads ? r0 + J->sst_batch : J->n_reads;
		batch_out_t *B = &J->bout[b];
		B->mem_cnt = (uint32_t *)calloc((size_t)(r1 - r0) + 1, sizeof(uint32_t));
		sst_clear(&w.fwd); sst_clear(&w.bwd); /* comp_seed.cpp:2254 */
		w.fwd.calls = w.bwd.calls = w.fwd.blocks = w.bwd.blocks = 0;
		for (int64_t r = r0; r < r1; ++r) {
			int len = (int)(J->offsets[r + 1] - J->offsets[r]);
			if (len + 2 > seq_cap) {
				seq_cap = len + 2;
				seq = (uint8_t *)realloc(seq, (size_t)seq_cap);
				w.lep = (cso_intv_t *)realloc(w.lep, (size_t)seq_cap * sizeof(cso_intv_t));
				w.mem = (cso_intv_t *)realloc(w.mem, (size_t)seq_cap * sizeof(cso_intv_t));
			}
			const uint8_t *src = J->bases + J->offsets[r];
			for (int j = 0; j < len; ++j) seq[j] = src[j] > 4 ? cso_nt4_table[src[j]] : src[j]; /* comp_seed.cpp:2258-2260 */
			size_t before = B->mems.n;
			seed_one_read(&w, J->idx, J->par, J->mode, seq, len, &B->mems);
			B->mem_cnt[r - r0] = (uint32_t)(B->mems.n - before);
		}
		if (J->mode) {
			w.st.bwt_calls += w.fwd.calls + w.bwd.calls;
			w.st.bwt_blocks += w.fwd.blocks + w.bwd.blocks;
		}
		if (J->want_sal) sal_batch(&w, J, B, r1 - r0);
	}
	if (!J->mode) { w.st.bwt_calls = w.st.bwt_queries; w.st.bwt_blocks = w.st.bwt_blocks_uncached; }
	pthread_mutex_lock(&J->mu);
	uint64_t *t = (uint64_t *)&J->total, *s = (uint64_t *)&w.st;
	uint64_t mx = J->total.lep_max > w.st.lep_max ? J->total.lep_max : w.st.lep_max;
	for (size_t i = 0; i < sizeof(cso_stats_t) / 8; ++i) t[i] += s[i];
	J->total.lep_max = mx;
	pthread_mutex_unlock(&J->mu);
	free(seq); free(w.lep); free(w.mem); free(w.fwd.a); free(w.bwd.a);
	return NULL;
}

int cso_seed_batch(const cso_index_t *idx, const cso_params_t *par, int64_t n_reads, const uint8_t *bases,
                   const uint64_t *offsets, int mode, int sst_batch, int want_sal, int n_threads,
                   uint64_t **mem_off, cso_intv_t **mems, uint64_t **seed_off, cso_seed_t **seeds, cso_stats_t *st)
{
	if (sst_batch < 1) sst_batch = 512;
	if (n_threads < 1) n_threads = 1;
	job_t J;
	memset(&J, 0, sizeof J);
	J.idx = idx; J.par = par; J.n_reads = n_reads; J.bases = bases; J.offsets = offsets;
	J.mode = mode; J.sst_batch = sst_batch; J.want_sal = want_sal;
	J.n_batches = (n_reads + sst_batch - 1) / sst_batch;
	J.bout = (batch_out_t *)calloc((size_t)J.n_batches + 1, sizeof(batch_out_t));
	pthread_mutex_init(&J.mu, NULL);
	if (n_threads > J.n_batches) n_threads = J.n_batches > 0 ? (int)J.n_batches : 1;
	pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
	for (int t = 1; t < n_threads; ++t) pthread_create(&th[t], NULL, worker_main, &J);
	worker_main(&J);
	for (int t = 1; t < n_threads; ++t) pthread_join(th[t], NULL);
	free(th);

	uint64_t nm = 0, ns = 0;
	for (int64_t b = 0; b < J.n_batches; ++b) { nm += J.bout[b].mems.n; ns += J.bout[b].n_seeds; }
	*mem_off = (uint64_t *)malloc(((size_t)n_reads + 1) * 8);
	*mems = (cso_intv_t *)malloc((nm + 1) * sizeof(cso_intv_t));
	if (seed_off) *seed_off = (uint64_t *)malloc(((size_t)n_reads + 1) * 8);
	if (seeds) *seeds = (cso_seed_t *)malloc((ns + 1) * sizeof(cso_seed_t));
	uint64_t pm = 0, ps = 0;
	(*mem_off)[0] = 0;
	if (seed_off) (*seed_off)[0] = 0;
	for (int64_t b = 0; b < J.n_batches; ++b) {
		batch_out_t *B = &J.bout[b];
		int64_t r0 = b * sst_batch, r1 = r0 + sst_batch < n_reads ? r0 + sst_batch : n_reads;
		memcpy(*mems + pm, B->mems.a, B->mems.n * sizeof(cso_intv_t));
		if (seeds && B->n_seeds) memcpy(*seeds + ps, B->seeds, B->n_seeds * sizeof(cso_seed_t));
		for (int64_t r = r0; r < r1; ++r) {
			pm += B->mem_cnt[r - r0];
			(*mem_off)[r + 1] = pm;
			if (seed_off) { ps += B->seed_cnt ? B->seed_cnt[r - r0] : 0; (*seed_off)[r + 1] = ps; }
		}
		free(B->mems.a); free(B->mem_cnt); free(B->seeds); free(B->seed_cnt);
	}
	free(J.bout);
	J.total.n_mems = nm; J.total.n_seeds = ns;
	if (st) *st = J.total;
	pthread_mutex_destroy(&J.mu);
	return 0;
}

void cso_free(void *p) { free(p); }
