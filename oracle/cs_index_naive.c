/* oracle/cs_index_naive.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * A plain, slow-but-obvious FM-index builder for the CPU-side tests: it lets a test that runs without a GPU (and
 * without the reference) obtain the index of a genome it generates itself, e.g. the E. coli-size plumbing set of
 * BASELINE.json configs[0].  It produces exactly the arrays `bwaidx` writes (FM_index/index_main.c:257-325):
 *   text T = forward ++ reverse complement                         bns_fasta2bntseq, FM_index/bntseq.c:306-312
 *   BWT of T$ without the $ row, primary, L2                       bwt_pac2bwt, index_main.c:66-127
 *   Occ counts interleaved every 128 rows + one trailing record    bwt_bwtupdate_core, index_main.c:152-174
 *   SA sampled every 32 rows, sa[0] = -1                           bwt_cal_sa, FM_index/bwt.c:62-84
 * The reference builds the BWT with SA-IS / BWT-SW; the BWT of T$ is unique, so a comparison sort of all suffixes gives
 * the same bytes.  Pinned by tests/test_oracle.py: the index built here from the fixture's .pac equals the fixture's
 * .bwt/.sa (written by the reference's own bwaidx) byte for byte.
 *
 * Method: suffixes are bucketed by their first 12 bases (counting sort), every bucket is sorted with a comparator that
 * compares 32 bases per step on a 2-bit packed copy of the text; buckets are independent, so threads share them.
 */
#include "cs_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const uint64_t *g_pk;   /* packed text, 32 bases per word, first base in the top bits */
static uint64_t g_n;

static inline uint64_t win32(uint64_t i) /* the 32 bases from position i on; positions >= n read as A (0) */
{
	uint64_t w = i >> 5; unsigned sh = (unsigned)(i & 31) << 1;
	uint64_t a = g_pk[w];
	return sh ? (a << sh) | (g_pk[w + 1] >> (64 - sh)) : a;
}

static int cmp_suffix(const void *pa, const void *pb)
{
	uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
	if (a == b) return 0;
	for (uint64_t off = 0;; off += 32) {
		uint64_t la = g_n - a - off, lb = g_n - b - off;         /* bases left in each suffix */
		uint64_t lim = la < lb ? la : lb;
		uint64_t wa = win32(a + off), wb = win32(b + off);
		if (wa != wb) {
			unsigned d = (unsigned)__builtin_clzll(wa ^ wb) >> 1; /* first differing base of the window */
			if (d < lim) return wa < wb ? -1 : 1;
			return la < lb ? -1 : 1;                              /* one suffix ends before they differ: the shorter is smaller */
		}
		if (lim <= 32) return la < lb ? -1 : 1;
	}
}

typedef struct { uint64_t *sfx; const uint64_t *start; uint64_t b0, b1; } job_t;
static void *sort_job(void *arg)
{
	job_t *j = (job_t *)arg;
	for (uint64_t b = j->b0; b < j->b1; ++b) {
		uint64_t cnt = j->start[b + 1] - j->start[b];
		if (cnt > 1) qsort(j->sfx + j->start[b], cnt, sizeof(uint64_t), cmp_suffix);
	}
	return NULL;
}

int cso_index_build(const uint8_t *fwd_nt4, uint64_t l_pac, int n_threads, cso_index_t *out)
{
	if (!fwd_nt4 || !out || l_pac == 0) return -1;
	const uint64_t n = 2 * l_pac, m = n + 1;
	const int KB = 12; const uint64_t NB = 1ull << (2 * KB);
	uint8_t *T = (uint8_t *)malloc(n + 64);
	uint64_t *pk = (uint64_t *)calloc((n >> 5) + 4, 8);
	uint64_t *sfx = (uint64_t *)malloc(m * 8), *start = (uint64_t *)calloc(NB + 2, 8);
	if (!T || !pk || !sfx || !start) { free(T); free(pk); free(sfx); free(start); return -3; }
	for (uint64_t i = 0; i < l_pac; ++i) {
		if (fwd_nt4[i] > 3) { free(T); free(pk); free(sfx); free(start); return -1; }
		T[i] = fwd_nt4[i]; T[n - 1 - i] = (uint8_t)(3 - fwd_nt4[i]);
	}
	for (uint64_t i = 0; i < n; ++i) pk[i >> 5] |= (uint64_t)T[i] << ((31 - (i & 31)) << 1);
	g_pk = pk; g_n = n;
	/* counting sort by the first KB bases; row 0 is the empty suffix (the $), kept apart */
	for (uint64_t i = 0; i < n; ++i) ++start[(win32(i) >> (64 - 2 * KB)) + 1];
	for (uint64_t b = 0; b < NB; ++b) start[b + 1] += start[b];
	{
		uint64_t *fill = (uint64_t *)malloc(NB * 8);
		if (!fill) { free(T); free(pk); free(sfx); free(start); return -3; }
		memcpy(fill, start, NB * 8);
		sfx[0] = n;
		for (uint64_t i = 0; i < n; ++i) sfx[1 + fill[win32(i) >> (64 - 2 * KB)]++] = i;
		free(fill);
	}
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 64) n_threads = 64;
	{
		pthread_t th[64]; job_t jobs[64];
		uint64_t per = (n + (uint64_t)n_threads - 1) / (uint64_t)n_threads, b = 0;
		int nj = 0;
		for (int t = 0; t < n_threads && b < NB; ++t) { /* bucket ranges holding about n / n_threads suffixes each */
			uint64_t b1 = b, goal = start[b] + per;
			while (b1 < NB && start[b1] < goal) ++b1;
			jobs[nj].sfx = sfx + 1; jobs[nj].start = start; jobs[nj].b0 = b; jobs[nj].b1 = (t == n_threads - 1) ? NB : b1;
			b = jobs[nj].b1; ++nj;
		}
		for (int t = 0; t < nj; ++t) pthread_create(&th[t], NULL, sort_job, &jobs[t]);
		for (int t = 0; t < nj; ++t) pthread_join(th[t], NULL);
	}
	/* BWT string without the $ row, Occ records, SA samples */
	const uint64_t n_words = (n + 15) >> 4, n_blocks = (n + 127) >> 7, bwt_size = n_words + (n_blocks + 1) * 8, n_sa = (n + 32) / 32;
	uint32_t *words = (uint32_t *)calloc(n_words + 8, 4), *bwt = (uint32_t *)calloc(bwt_size + 16, 4);
	uint64_t *sa = (uint64_t *)malloc(n_sa * 8);
	if (!words || !bwt || !sa) { free(T); free(pk); free(sfx); free(start); free(words); free(bwt); free(sa); return -3; }
	uint64_t primary = 0;
	for (uint64_t r = 0; r < m; ++r) if (sfx[r] == 0) { primary = r; break; }
	uint64_t tot[4] = {0, 0, 0, 0};
	for (uint64_t pos = 0; pos < n; ++pos) {
		uint64_t row = pos + (pos >= primary);
		uint32_t c = T[sfx[row] - 1];
		words[pos >> 4] |= c << ((15 - (pos & 15)) << 1);
	}
	for (uint64_t b = 0; b <= n_blocks; ++b) { /* record b: counts before row 128 b, then (except the trailing one) its 8 words */
		uint64_t at = b < n_blocks ? b * 16 : n_words + 8 * n_blocks; /* the trailing record follows the last, possibly partial, block */
		for (int c = 0; c < 4; ++c) { bwt[at + 2 * c] = (uint32_t)tot[c]; bwt[at + 2 * c + 1] = (uint32_t)(tot[c] >> 32); }
		if (b == n_blocks) break;
		for (int w = 0; w < 8; ++w) {
			uint64_t wi = b * 8 + (uint64_t)w;
			if (wi >= n_words) break;
			bwt[at + 8 + w] = words[wi];
			for (int t = 0; t < 16; ++t) { uint64_t pos = wi * 16 + (uint64_t)t; if (pos < n) ++tot[(words[wi] >> ((15 - t) << 1)) & 3u]; }
		}
	}
	for (uint64_t t = 0; t < n_sa; ++t) sa[t] = t == 0 ? ~0ull : sfx[t << 5];
	memset(out, 0, sizeof *out);
	out->primary = primary; out->L2[0] = 0;
	for (int c = 0; c < 4; ++c) out->L2[c + 1] = out->L2[c] + tot[c];
	out->seq_len = n; out->bwt_size = bwt_size; out->bwt = bwt; out->sa_intv = 32; out->n_sa = n_sa; out->sa = sa;
	out->owned_bwt = bwt; out->owned_sa = sa;
	free(T); free(pk); free(sfx); free(start); free(words);
	return out->L2[4] == n ? 0 : -2;
}
