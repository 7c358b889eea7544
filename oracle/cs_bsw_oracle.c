/* oracle/cs_bsw_oracle.c -- TEST INFRASTRUCTURE ONLY (the checker of the GPU extension kernel; never linked into the product).
 *
 * Plain-C restatement of the reference's banded Smith-Waterman seed extension: ksw_extend2 (bwalib/ksw.c:380-479), which is what
 * BandedPairWiseSW::scalarBandedSWA (mapping/bandedSWA.cpp:116-240) repeats and what the vectorised getScores8 / getScores16
 * (bandedSWA.cpp:412..., called from mem_chain2aln_across_reads_V2, mapping/comp_seed.cpp:1790,1859,2003,2074) must equal.
 * PINNED: tests/test_oracle_bsw.py replays every pair the real reference extended on the golden read sets (recorded by
 * oracle/ref_bsw_trace.cpp from the reference's own run; tests/golden/bsw1/) and requires all six outputs to be identical.
 *
 * Stated row by row over two arrays, the way the device kernel works (one column per lane):
 *   Hd[j] = H(i-1, j-1), the diagonal predecessor of column j (Hd[0] = the first-column score of the previous row)
 *   Ev[j] = E(i, j), the best score of a path that enters cell (i, j) by a deletion (a gap in the query)
 * A row [beg, end) needs three passes that each depend on the previous row only:
 *   M(j) = Hd[j] ? Hd[j] + S(t_i, q_j) : 0           a path may not restart from a zero cell (ksw.c:436)
 *   F(j) = max over beg <= k < j of max(M(k) - (o_ins + e_ins), 0) - (j - 1 - k) e_ins, F(beg) = 0    (a running maximum: ksw.c:446-449;
 *          insertions open from M only, never from E or F: "100M3I3D20M" is not allowed)
 *   H(j) = max(M(j), Ev[j], F(j));  Ev'[j] = max(Ev[j] - e_del, max(M(j) - (o_del + e_del), 0));  Hd'[j + 1] = H(j)
 * The band is adaptive: after each row, leading and trailing columns whose Hd' and Ev' are both 0 are dropped (ksw.c:470-473), and
 * columns that were dropped keep whatever the arrays last held -- so the arrays are kept exactly as the reference keeps `eh`. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cs_oracle.h"

static inline int imax(int a, int b) { return a > b ? a : b; }

/* Score of target code t against query code q.  Two rules exist in the reference and they differ for codes above 4 (a '-' in a read is
 * code 5, FM_index/bntseq.c:46-63): the scalar code indexes the 5 x 5 matrix, mat[5 t + q] (ksw.c:392-395), whereas the vectorised code
 * -- which is what extends every pair shorter than 32768 (comp_seed.cpp:1569-1577) -- compares the codes: ambiguous (== 4) on either
 * side scores w_ambig = -1, equal codes score the match, anything else the mismatch (mapping/bandedSWA.cpp:286-290, AMBIG :44). */
int cso_bsw_score(const cso_bsw_params_t *P, int vec_rule, int t, int q)
{
	if (!vec_rule) return P->mat[5 * t + q];
	if (t == 4 || q == 4) return -1;
	return t == q ? P->mat[0] : P->mat[1];
}
/* the reference's dispatch: pairs of the "8" and "16" classes go to the vectorised code, the rest to the scalar code (comp_seed.cpp:1569-1577) */
int cso_bsw_uses_vec_rule(const cso_bsw_params_t *P, int qlen, int tlen, int h0)
{
	const int mn = (qlen < tlen ? qlen : tlen) * P->mat[0] + h0;
	return qlen < 32768 && tlen < 32768 && mn < 32768;
}

int cso_extend_pair(const cso_bsw_params_t *P, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int h0, cso_bsw_result_t *out)
{
	return cso_extend_pair_rule(P, cso_bsw_uses_vec_rule(P, qlen, tlen, h0), qlen, query, tlen, target, w, h0, out);
}
int cso_extend_pair_rule(const cso_bsw_params_t *P, int vec_rule, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int h0, cso_bsw_result_t *out)
{
	const int gapo_del = P->o_del + P->e_del, gapo_ins = P->o_ins + P->e_ins;
	int *Hd = (int *)calloc((size_t)qlen + 2, sizeof(int)), *Ev = (int *)calloc((size_t)qlen + 2, sizeof(int));
	int *Mrow = (int *)malloc(((size_t)qlen + 2) * sizeof(int));
	if (!Hd || !Ev || !Mrow) { free(Hd); free(Ev); free(Mrow); return -1; }
	/* row "-1": the seed's score decays along the query by one insertion (ksw.c:398-400) */
	Hd[0] = h0;
	if (qlen >= 1) Hd[1] = h0 > gapo_ins ? h0 - gapo_ins : 0;
	for (int j = 2; j <= qlen && Hd[j - 1] > P->e_ins; ++j) Hd[j] = Hd[j - 1] - P->e_ins;
	/* the band cannot usefully be wider than the longest gap the best possible score pays for (ksw.c:402-410) */
	int best = 0;
	for (int k = 0; k < 25; ++k) best = imax(best, P->mat[k]);
	int lim = (int)((double)(qlen * best + P->end_bonus - P->o_ins) / P->e_ins + 1.);
	w = w < imax(lim, 1) ? w : imax(lim, 1);
	lim = (int)((double)(qlen * best + P->end_bonus - P->o_del) / P->e_del + 1.);
	w = w < imax(lim, 1) ? w : imax(lim, 1);

	int top = h0, top_i = -1, top_j = -1, g_best = -1, g_row = -1, off = 0;
	int beg = 0, end = qlen;
	for (int i = 0; i < tlen; ++i) {
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		const int ti = target[i];
		int left = 0; /* H(i, beg - 1): the first column while the band still starts there (ksw.c:419-423) */
		if (beg == 0) left = imax(h0 - (P->o_del + P->e_del * (i + 1)), 0);
		for (int j = beg; j < end; ++j) Mrow[j] = Hd[j] ? Hd[j] + cso_bsw_score(P, vec_rule, ti, query[j]) : 0;
		int f = 0, row_max = 0, row_arg = -1, h = left;
		for (int j = beg; j < end; ++j) {
			const int M = Mrow[j];
			Hd[j] = h;                               /* H(i, j-1) becomes the diagonal of column j in the next row */
			h = imax(imax(M, Ev[j]), f);
			if (h >= row_max) { row_max = h; row_arg = j; } /* the LAST column that reaches the row maximum (ksw.c:440-441) */
			Ev[j] = imax(Ev[j] - P->e_del, imax(M - gapo_del, 0));
			f = imax(f - P->e_ins, imax(M - gapo_ins, 0));
		}
		Hd[end] = h; Ev[end] = 0;
		if ((beg < end ? end : beg) == qlen) { /* the row reached the end of the query: a candidate for the end-to-end score (ksw.c:452-455) */
			if (!(g_best > h)) g_row = i;
			g_best = imax(g_best, h);
		}
		if (row_max == 0) break;
		if (row_max > top) {
			top = row_max; top_i = i; top_j = row_arg;
			off = imax(off, abs(row_arg - i));
		} else if (P->zdrop > 0) { /* Z-drop with the diagonal shift priced as a gap extension (ksw.c:461-467) */
			const int di = i - top_i, dj = row_arg - top_j;
			if (di > dj) { if (top - row_max - (di - dj) * P->e_del > P->zdrop) break; }
			else if (top - row_max - (dj - di) * P->e_ins > P->zdrop) break;
		}
		int j = beg;
		while (j < end && Hd[j] == 0 && Ev[j] == 0) ++j;
		beg = j;
		j = end;
		while (j >= beg && Hd[j] == 0 && Ev[j] == 0) --j;
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(Hd); free(Ev); free(Mrow);
	out->score = top; out->qle = top_j + 1; out->tle = top_i + 1; out->gtle = g_row + 1; out->gscore = g_best; out->max_off = off;
	return 0;
}

void cso_bsw_params_default(cso_bsw_params_t *P)
{
	/* mem_opt_init (mapping/comp_seed.cpp:26-58): a = 1, b = 4, o = 6, e = 1, zdrop = 100, pen_clip = 5; bwa_fill_scmat (bwalib/bwa.c:17-29) */
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 5; ++j) P->mat[5 * i + j] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? 1 : -4);
	P->o_del = P->o_ins = 6; P->e_del = P->e_ins = 1; P->zdrop = 100; P->end_bonus = 5;
}

/* a batch, one pair per (q_off, t_off, qlen, tlen, h0): int32 x 5 per pair plus 64-bit offsets; n_threads over contiguous ranges */
#include <pthread.h>
typedef struct { const cso_bsw_params_t *P; const cso_bsw_pair_t *pairs; const uint8_t *q, *t; int w; cso_bsw_result_t *out; int64_t a, b; int rc; } bsw_job_t;
static void *bsw_worker(void *arg)
{
	bsw_job_t *J = (bsw_job_t *)arg;
	for (int64_t i = J->a; i < J->b; ++i) {
		const cso_bsw_pair_t *p = J->pairs + i;
		if (cso_extend_pair(J->P, p->qlen, J->q + p->q_off, p->tlen, J->t + p->t_off, J->w, p->h0, J->out + i)) J->rc = -1;
	}
	return NULL;
}
int cso_extend_batch(const cso_bsw_params_t *P, int64_t n, const cso_bsw_pair_t *pairs, const uint8_t *qbuf, const uint8_t *tbuf, int w,
                     int n_threads, cso_bsw_result_t *out)
{
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 64) n_threads = 64;
	pthread_t th[64]; bsw_job_t job[64];
	for (int t = 0; t < n_threads; ++t) {
		bsw_job_t j = {P, pairs, qbuf, tbuf, w, out, n * t / n_threads, n * (t + 1) / n_threads, 0};
		job[t] = j;
		if (n_threads == 1) bsw_worker(&job[t]); else pthread_create(&th[t], NULL, bsw_worker, &job[t]);
	}
	int rc = 0;
	for (int t = 0; t < n_threads; ++t) { if (n_threads > 1) pthread_join(th[t], NULL); if (job[t].rc) rc = -1; }
	return rc;
}
