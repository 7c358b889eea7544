// dedup.cpp -- what the reference does with a read's alignment regions after the extension stage (mapping/comp_seed.cpp:2385-2395): the
// regions the purge marked (qb = qe = -1) are dropped and mem_sort_dedup_patch (comp_seed.cpp:629-687) runs over the rest -- host code:
//   * sorted by END position on the reference (klib's introsort: the order among equal ends is reproduced, klib_sort.hpp),
//   * going up the list, a region is compared with the earlier ones that end within max_chain_gap of its start on the same sequence:
//     if the two overlap by more than mask_level_redun of the shorter one on both read and reference, the lower-scoring one goes; else,
//     if they are colinear and close to one diagonal, mem_patch_reg (comp_seed.cpp:599-627) aligns the read from the start of the first
//     to the end of the second globally against the reference span (bwa_gen_cigar2, bwalib/bwa.c:147-194; ksw_global2, bwalib/ksw.c:504)
//     and, if that scores at least 0.9 of what the two regions promise, merges them into one,
//   * sorted by score (then start, then read start), identical neighbours dropped.
// The global alignment is score-only here: rows = reference, a band of +-w columns around the main diagonal, gaps opening from the
// diagonal move only (ksw.c:536-545), both sequences reversed first for a span on the reverse strand, as the reference does to put
// indels leftmost -- the band is tied to the top-left corner, so that matters for the score, too.
#include "cs_internal.hpp"
#include "klib_sort.hpp"

#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {
constexpr int32_t NEG = -0x40000000; // MINUS_INF (ksw.c:490)

struct Reg { cs_alnreg_t r; int32_t n_comp; };

// ksw_global2's score (ksw.c:504-587, the branch without backtracking)
int banded_global_score(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w, std::vector<int32_t> &buf)
{
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	buf.assign(((size_t)qlen + 1) * 2, 0);
	int32_t *H = buf.data(), *E = H + qlen + 1;
	H[0] = 0; E[0] = NEG;
	int j = 1;
	for (; j <= qlen && j <= w; ++j) { H[j] = -(o_ins + e_ins * j); E[j] = NEG; }
	for (; j <= qlen; ++j) H[j] = E[j] = NEG;
	for (int i = 0; i < tlen; ++i) {
		int32_t f = NEG;
		const int8_t *row = mat + (size_t)t[i] * 5;
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int32_t h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG;
		for (j = beg; j < end; ++j) {
			int32_t m = H[j], e = E[j];
			H[j] = h1;
			m += row[q[j]];
			int32_t h = m >= e ? m : e;
			h = h >= f ? h : f;
			h1 = h;
			int32_t x = m - oe_del;
			e -= e_del; e = e > x ? e : x; E[j] = e;
			x = m - oe_ins;
			f -= e_ins; f = f > x ? f : x;
		}
		H[end] = h1; E[end] = NEG;
	}
	return H[qlen];
}

struct Ctx {
	const cs_refseq_view *ref; const std::vector<uint8_t> *pac; cs_aln_params_t o; cs_dedup_params_t d; int8_t mat[25];
	std::vector<uint8_t> qbuf, tbuf; std::vector<int32_t> dp;
};

// bwa_gen_cigar2 without the CIGAR: the score of the global alignment of query[0, l_query) against the reference span [rb, re); false = no alignment
bool global_span_score(Ctx &C, int w_, int l_query, const uint8_t *query, int64_t rb, int64_t re, int *score)
{
	const int64_t l_pac = C.ref->l_pac;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
	int64_t b = rb, e = re;                                     // bns_get_seq (bntseq.c:403-424)
	if (e > (l_pac << 1)) e = l_pac << 1;
	if (b < 0) b = 0;
	if (!(b >= l_pac || e <= l_pac)) return false;
	const int64_t rlen = e - b;
	if (re - rb != rlen) return false;
	C.tbuf.resize((size_t)rlen);
	for (int64_t p = b; p < e; ++p) C.tbuf[(size_t)(p - b)] = p >= l_pac ? (uint8_t)(3 - cs_pac_base_(*C.pac, (l_pac << 1) - 1 - p)) : cs_pac_base_(*C.pac, p);
	C.qbuf.assign(query, query + l_query);
	if (rb >= l_pac) { std::reverse(C.qbuf.begin(), C.qbuf.end()); std::reverse(C.tbuf.begin(), C.tbuf.end()); }
	if (l_query == re - rb && w_ == 0) {
		int sc = 0;
		for (int i = 0; i < l_query; ++i) sc += C.mat[C.tbuf[(size_t)i] * 5 + C.qbuf[(size_t)i]];
		*score = sc;
		return true;
	}
	const int max_ins = (int)((double)(((l_query + 1) >> 1) * C.mat[0] - C.o.o_ins) / C.o.e_ins + 1.);
	const int max_del = (int)((double)(((l_query + 1) >> 1) * C.mat[0] - C.o.o_del) / C.o.e_del + 1.);
	int max_gap = std::max(max_ins, max_del);
	max_gap = std::max(max_gap, 1);
	int w = (max_gap + std::abs((int)rlen - l_query) + 1) >> 1;
	w = std::min(w, w_);
	const int min_w = std::abs((int)rlen - l_query) + 3;
	w = std::max(w, min_w);
	*score = banded_global_score(l_query, C.qbuf.data(), (int)rlen, C.tbuf.data(), C.mat, C.o.o_del, C.o.e_del, C.o.o_ins, C.o.e_ins, w, C.dp);
	return true;
}

// mem_patch_reg (comp_seed.cpp:599-627): can a (earlier start) and b be one alignment?  returns its score (0: no) and the band it needs
int patch_score(Ctx &C, const uint8_t *query, const cs_alnreg_t &a, const cs_alnreg_t &b, int *w_out)
{
	const int64_t l_pac = C.ref->l_pac;
	if (a.rb < l_pac && b.rb >= l_pac) return 0;               // on different strands
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0; // not colinear
	int w = (int)((a.re - b.rb) - (a.qe - b.qb));              // the diagonal shift between the two
	w = w > 0 ? w : -w;
	double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > C.o.w << 1 || r >= 0.05f) return 0; }   // PATCH_MAX_R_BW
	else if (w > C.o.w << 2 || r >= 0.05f * 2) return 0;
	w += a.w + b.w;
	w = std::min(w, C.o.w << 2);
	int score = 0;
	global_span_score(C, w, b.qe - a.qb, query + a.qb, a.rb, b.re, &score);   // (a failed fetch leaves the score at 0, as the reference's uninitialised int usually is not -- see below)
	const int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	const int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;                   // PATCH_MIN_SC_RATIO
	*w_out = w;
	return score;
}

void dedup_range(const cs_aligner_core &A, const cs_dedup_params_t &d, const cs_aln_result_t &in, const uint8_t *bases, const uint64_t *read_off, int64_t r0, int64_t r1,
                 std::vector<Reg> &out, std::vector<uint32_t> &per_read)
{
	Ctx C; C.ref = A.ref; C.pac = A.pac; C.o = *A.par; C.d = d;
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) C.mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? C.o.a : -C.o.b); // bwa_fill_scmat (bwalib/bwa.c:17-29)
	std::vector<Reg> a; std::vector<uint8_t> query;
	for (int64_t rd = r0; rd < r1; ++rd) {
		a.clear();
		for (uint64_t k = in.reg_off[rd]; k < in.reg_off[rd + 1]; ++k) if (in.regs[k].qe > in.regs[k].qb) a.push_back({in.regs[k], 0}); // (comp_seed.cpp:2387-2393; n_comp as the extension stage's calloc left it)
		int n = (int)a.size();
		if (n > 1) {
			const int l_query = (int)(read_off[rd + 1] - read_off[rd]);
			query.resize((size_t)l_query);
			for (int j = 0; j < l_query; ++j) query[(size_t)j] = cs_base_code_(bases[read_off[rd] + (uint64_t)j]);
			cs_klib_introsort((size_t)n, a.data(), [](const Reg &x, const Reg &y) { return x.r.re < y.r.re; });   // by the END position (alnreg_slt2)
			for (auto &g : a) g.n_comp = 1;                                                                          // (a read's only region keeps 0: the function returns before this line)
			for (int i = 1; i < n; ++i) {
				cs_alnreg_t &p = a[(size_t)i].r;
				if (p.rid != a[(size_t)i - 1].r.rid || p.rb >= a[(size_t)i - 1].r.re + d.max_chain_gap) continue;
				for (int j = i - 1; j >= 0 && p.rid == a[(size_t)j].r.rid && p.rb < a[(size_t)j].r.re + d.max_chain_gap; --j) {
					cs_alnreg_t &q = a[(size_t)j].r;
					if (q.qe == q.qb) continue;                                          // excluded before
					const int64_t o_r = q.re - p.rb;
					const int64_t o_q = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
					const int64_t m_r = std::min(q.re - q.rb, p.re - p.rb);
					const int64_t m_q = std::min(q.qe - q.qb, p.qe - p.qb);
					int score, w;
					if ((float)o_r > d.mask_level_redun * (float)m_r && (float)o_q > d.mask_level_redun * (float)m_q) { // one of the two is redundant
						if (p.score < q.score) { p.qe = p.qb; break; }
						q.qe = q.qb;
					} else if (q.rb < p.rb && (score = patch_score(C, query.data(), q, p, &w)) > 0) {                   // merge q into p
						a[(size_t)i].n_comp += a[(size_t)j].n_comp + 1;
						p.seedcov = std::max(p.seedcov, q.seedcov);
						p.qb = q.qb; p.rb = q.rb;
						p.truesc = p.score = score;
						p.w = w;
						q.qb = q.qe;
					}
				}
			}
			int m = 0;
			for (int i = 0; i < n; ++i) if (a[(size_t)i].r.qe > a[(size_t)i].r.qb) { if (m != i) a[(size_t)m] = a[(size_t)i]; ++m; }
			n = m;
			cs_klib_introsort((size_t)n, a.data(), [](const Reg &x, const Reg &y) {                                    // alnreg_slt
				return x.r.score > y.r.score || (x.r.score == y.r.score && (x.r.rb < y.r.rb || (x.r.rb == y.r.rb && x.r.qb < y.r.qb))); });
			for (int i = 1; i < n; ++i)
				if (a[(size_t)i].r.score == a[(size_t)i - 1].r.score && a[(size_t)i].r.rb == a[(size_t)i - 1].r.rb && a[(size_t)i].r.qb == a[(size_t)i - 1].r.qb) a[(size_t)i].r.qe = a[(size_t)i].r.qb;
			m = n > 0 ? 1 : 0;
			for (int i = 1; i < n; ++i) if (a[(size_t)i].r.qe > a[(size_t)i].r.qb) { if (m != i) a[(size_t)m] = a[(size_t)i]; ++m; }
			n = m;
		}
		out.insert(out.end(), a.begin(), a.begin() + n);
		per_read.push_back((uint32_t)n);
	}
}
} // namespace

extern "C" void cs_dedup_params_default(cs_dedup_params_t *p)
{
	if (!p) return;
	p->max_chain_gap = 10000; p->mask_level_redun = 0.95f; // mem_opt_init (comp_seed.cpp:26-58)
}

int cs_dedup_regions_(const cs_aligner_core &A, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets,
                      std::vector<uint64_t> &out_off, std::vector<cs_alnreg_t> &out_regs, std::vector<int32_t> &out_ncomp)
{
	const int64_t n = regs->n_reads;
	int T = std::max(1, std::min(A.par->threads, 256));
	if (n < 1024) T = 1;
	// chunks of reads, handed out by a counter (cs_for_chunks_: a read with 2,000 regions is two million pair tests)
	const int64_t CH = cs_chunk_reads_(n, T), K = (n + CH - 1) / CH;
	std::vector<std::vector<Reg>> part((size_t)K); std::vector<std::vector<uint32_t>> cnt((size_t)K);
	cs_for_chunks_(T, K, [&](int64_t k) { dedup_range(A, *par, *regs, bases, read_offsets, k * CH, std::min(n, (k + 1) * CH), part[(size_t)k], cnt[(size_t)k]); });
	std::vector<size_t> rb((size_t)K + 1, 0);
	for (int64_t k = 0; k < K; ++k) rb[(size_t)k + 1] = rb[(size_t)k] + part[(size_t)k].size();
	out_off.resize((size_t)n + 1); out_off[0] = 0; out_regs.resize(rb[(size_t)K]); out_ncomp.resize(rb[(size_t)K]);
	cs_for_chunks_(T, K, [&](int64_t k) { // every chunk's share to where the prefix sums say
		uint64_t o = rb[(size_t)k]; int64_t r = k * CH;
		for (uint32_t q : cnt[(size_t)k]) { o += q; out_off[(size_t)++r] = o; }
		size_t i = rb[(size_t)k];
		for (const Reg &g : part[(size_t)k]) { out_regs[i] = g.r; out_ncomp[i] = g.n_comp; ++i; }
	});
	return CS_OK;
}
