// chain_filter.cpp -- the two filters between chaining and extension: mem_chain_flt (mapping/comp_seed.cpp:297-360) and
// mem_flt_chained_seeds (comp_seed.cpp:393-412, with mem_seed_sw :367-391), per read, host code; together with cs_chain_batch in front
// and cs_extend_chains behind, the library's side of comp_seed.cpp:2361-2374.
//
// mem_chain_flt: every chain gets a weight (bases of the read, or of the reference if fewer, covered by its seeds), the chains are
// sorted by weight, and going down the list a chain is dropped when it overlaps a kept, much heavier one on the read; the first chain
// shadowed by each kept one survives too (for the mapping quality), at most max_chain_extend such extras.  What has to be reproduced
// beyond the rule is the ORDER among chains of equal weight: the reference sorts with klib's introsort, which is not stable, and the
// order decides which chain is "kept" and which is "shadowed".  So that sort is restated here with klib's steps (median of first /
// middle+1 / last, pivot parked at the end, partitions of 16 or fewer left to one final insertion sort, comb sort when 2 log2(n) levels
// are used up; cstl/ksort.h:146-226) over (weight, chain index) records.
//
// mem_flt_chained_seeds only acts on long reads (5.5 ln(l) <= 0.05 l, i.e. from ~700 bases): a seed shorter than 200 whose
// neighbourhood (50 bases either side) does not reach a local alignment score of ~5.5 ln(l) is dropped, the others get that score.  The
// score is ksw_align2's (bwalib/ksw.c:343, 16-bit striped kernel :232-331), an affine-gap Smith-Waterman whose deletion state is
// updated from H BEFORE the lazy-F correction across its eight query segments: restated as a scalar recurrence with exactly that rule.
#include "cs_internal.hpp"
#include "klib_sort.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct WRec { int32_t w, idx; };
inline bool heavier(const WRec &a, const WRec &b) { return a.w > b.w; } // flt_lt (comp_seed.cpp:294): descending weight

// bases covered by a chain's seeds, on the read and on the reference, whichever is less (mem_chain_weight, comp_seed.cpp:205-224)
int chain_weight(const cs_seed_t *sd, int n)
{
	int64_t end = 0; int wq = 0, wr = 0;
	for (int j = 0; j < n; ++j) {
		const int64_t b = sd[j].qbeg, e = b + sd[j].len;
		if (b >= end) wq += sd[j].len; else if (e > end) wq += (int)(e - end);
		end = std::max(end, e);
	}
	end = 0;
	for (int j = 0; j < n; ++j) {
		const int64_t b = sd[j].rbeg, e = b + sd[j].len;
		if (b >= end) wr += sd[j].len; else if (e > end) wr += (int)(e - end);
		end = std::max(end, e);
	}
	const int w = std::min(wq, wr);
	return w < (1 << 30) ? w : (1 << 30) - 1;
}

// ksw_align2's score for a query against a target (codes; the target holds 0..3): see the header comment.  Hprev / E hold one value
// per query position; segments of slen = ceil(qlen / 8) positions are the lanes' shares of the striped layout.
int striped_sw_score(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, std::vector<int32_t> &buf)
{
	if (qlen <= 0 || tlen <= 0) return 0;
	const int slen = (qlen + 7) / 8, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	buf.assign((size_t)qlen * 3, 0);
	int32_t *H = buf.data(), *E = H + qlen, *Hn = E + qlen;
	int best = 0;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *row = mat + (size_t)t[i] * 5;
		int f = 0, rowmax = 0;
		for (int j = 0; j < qlen; ++j) {            // the main loop of the striped kernel: F only from inside the position's own segment
			if (j % slen == 0) f = 0;
			int h = (j ? H[j - 1] : 0) + row[q[j]];
			h = std::max(h, E[j]); h = std::max(h, f);
			Hn[j] = h; rowmax = std::max(rowmax, h);
			E[j] = std::max(std::max(E[j] - e_del, 0), std::max(h - oe_del, 0));   // (unsigned saturating subtractions: never below 0)
			f = std::max(std::max(f - e_ins, 0), std::max(h - oe_ins, 0));
		}
		f = 0;
		for (int j = 0; j < qlen; ++j) {            // the lazy-F loop: insertions that cross segment boundaries reach H, not E
			const int h = std::max(Hn[j], f);
			Hn[j] = h;
			f = std::max(std::max(f - e_ins, 0), std::max(h - oe_ins, 0));
		}
		best = std::max(best, rowmax);
		std::swap(H, Hn);
	}
	return best;
}

struct ReadOut { std::vector<cs_chain_t> chains; std::vector<cs_seed_t> seeds; std::vector<int32_t> score; std::vector<uint32_t> per_read; };

// ---- the overlap scan of mem_chain_flt (comp_seed.cpp:311-336).  Chains in descending weight; chain i is compared with the chains kept so far,
// in the order they were kept: a kept chain j that overlaps i significantly on the read gets i as its `first` shadow (if it has none), and if i is
// much lighter than j the scan stops there and i is not kept; a chain that reaches the end of the list is kept (2: overlapping, 3: on its own).
// The loop as the reference writes it -- quadratic where nothing is "much lighter", e.g. the thousands of equal chains of a read inside a
// tandem array: 6.6 G steps for 400,000 reads of the golden reference, all of cs_chain_filter's time there.
inline bool sig_overlap(const cs_flt_params_t &o, int bj, int ej, bool aj, int bi, int ei, bool ai)
{
	const int b_max = std::max(bj, bi), e_min = std::min(ej, ei);
	if (!(e_min > b_max && (!aj || ai))) return false;   // no overlap on the read (not counted when the kept chain is ALT and this one is not)
	const int li = ei - bi, lj = ej - bj, min_l = std::min(li, lj);
	return (float)(e_min - b_max) >= (float)min_l * o.mask_level && min_l < o.max_chain_gap;
}
inline bool much_lighter(const cs_flt_params_t &o, int wi, int wj) { return (float)wi < (float)wj * o.drop_ratio && wj - wi >= (o.min_seed_len << 1); }
[[maybe_unused]] void overlap_scan_plain(const cs_flt_params_t &o, int n, const WRec *srt, const int *cb, const int *ce, const uint8_t *calt, std::vector<int> &keptv, std::vector<int> &first, std::vector<int> &nonov)
{
	keptv[0] = 3; nonov.push_back(0);
	for (int i = 1; i < n; ++i) {
		bool large = false; size_t k = 0;
		for (; k < nonov.size(); ++k) {
			const int j = nonov[k];
			if (sig_overlap(o, cb[j], ce[j], calt[j] != 0, cb[i], ce[i], calt[i] != 0)) {
				large = true;
				if (first[(size_t)j] < 0) first[(size_t)j] = i;
				if (much_lighter(o, srt[i].w, srt[j].w)) break;
			}
		}
		if (k == nonov.size()) { nonov.push_back(i); keptv[(size_t)i] = large ? 2 : 3; }
	}
}
// The same result without the quadratic part.  The kept chains are in descending weight, so "i is much lighter than kept chain k" holds for a
// PREFIX of the list (both halves of the test are monotone in the kept chain's weight; drop_ratio >= 0).  Inside that prefix the scan stops
// at the first significant overlap; behind it nothing can stop it, and all that is left to find is (a) whether any kept chain overlaps i --
// the first hit answers -- and (b) the kept chains without a shadow yet that overlap i: those are kept in a list of their own, which a
// chain leaves for good when it gets its shadow.  Among equal chains every kept chain is shadowed by the next one: the list holds one entry.
struct ScanBuf { std::vector<int> kb, ke, kw, unset; std::vector<uint8_t> ka; };
void overlap_scan(const cs_flt_params_t &o, int n, const WRec *srt, const int *cb, const int *ce, const uint8_t *calt, std::vector<int> &keptv, std::vector<int> &first, std::vector<int> &nonov, ScanBuf &B)
{
	if (!(o.drop_ratio >= 0.f)) { overlap_scan_plain(o, n, srt, cb, ce, calt, keptv, first, nonov); return; }
	B.kb.clear(); B.ke.clear(); B.kw.clear(); B.ka.clear(); B.unset.clear();
	auto keep = [&](int i, int mark) { B.unset.push_back((int)nonov.size()); nonov.push_back(i); B.kb.push_back(cb[i]); B.ke.push_back(ce[i]); B.kw.push_back(srt[i].w); B.ka.push_back(calt[i]); keptv[(size_t)i] = mark; };
	keep(0, 3);
	for (int i = 1; i < n; ++i) {
		const int bi = cb[i], ei = ce[i], wi = srt[i].w; const bool ai = calt[i] != 0;
		const int nk = (int)nonov.size();
		int k = 0; bool stopped = false;
		for (; k < nk && much_lighter(o, wi, B.kw[(size_t)k]); ++k)            // the prefix of much heavier kept chains
			if (sig_overlap(o, B.kb[(size_t)k], B.ke[(size_t)k], B.ka[(size_t)k] != 0, bi, ei, ai)) { stopped = true; break; }
		if (stopped) { const int j = nonov[(size_t)k]; if (first[(size_t)j] < 0) first[(size_t)j] = i; continue; }   // (its entry leaves `unset` when that list is next walked)
		const int nw = k;                                                      // kept chains [nw, nk): overlaps are recorded, nothing stops the scan
		bool large = false; size_t keep_u = 0;
		for (size_t u = 0; u < B.unset.size(); ++u) {
			const int q = B.unset[u], j = nonov[(size_t)q];
			if (first[(size_t)j] >= 0) continue;                               // shadowed meanwhile
			if (q >= nw && sig_overlap(o, B.kb[(size_t)q], B.ke[(size_t)q], B.ka[(size_t)q] != 0, bi, ei, ai)) { first[(size_t)j] = i; large = true; continue; }
			B.unset[keep_u++] = q;
		}
		B.unset.resize(keep_u);
		for (int q = nw; q < nk && !large; ++q) large = sig_overlap(o, B.kb[(size_t)q], B.ke[(size_t)q], B.ka[(size_t)q] != 0, bi, ei, ai);
		keep(i, large ? 2 : 3);
	}
}

void filter_range(const cs_chainer &C, const cs_flt_params_t &o, const cs_chain_result_t &in, const uint8_t *bases, const uint64_t *read_off, int64_t r0, int64_t r1, ReadOut &out)
{
	int8_t mat[25];
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? o.a : -o.b); // bwa_fill_scmat (bwalib/bwa.c:17-29)
	const int64_t l_pac = C.ref.l_pac;
	std::vector<WRec> srt; std::vector<int> keptv, first, order, nonov, cb, ce; std::vector<uint8_t> query, tseq, calt; std::vector<int32_t> swbuf; ScanBuf sc;
	for (int64_t r = r0; r < r1; ++r) {
		const uint64_t c0 = in.chain_off[r], c1 = in.chain_off[r + 1];
		const int l_query = (int)(read_off[r + 1] - read_off[r]);
		auto seeds_of = [&](int idx) { return in.cseeds + in.cseed_off[c0 + (uint64_t)idx]; };
		auto nseeds_of = [&](int idx) { return (int)(in.cseed_off[c0 + (uint64_t)idx + 1] - in.cseed_off[c0 + (uint64_t)idx]); };
		// ---- mem_chain_flt
		srt.clear();
		for (uint64_t c = c0; c < c1; ++c) {
			const int w = chain_weight(in.cseeds + in.cseed_off[c], (int)(in.cseed_off[c + 1] - in.cseed_off[c]));
			if (w >= o.min_chain_weight) srt.push_back({w, (int32_t)(c - c0)});
		}
		const int n = (int)srt.size();
		order.clear();
		if (n > 0) {
			cs_klib_introsort((size_t)n, srt.data(), heavier);
			// per chain, in the sorted order: its span on the read and whether it lies on an ALT contig
			cb.resize((size_t)n); ce.resize((size_t)n); calt.resize((size_t)n);
			for (int i = 0; i < n; ++i) {
				const int idx = srt[(size_t)i].idx; const cs_seed_t *sd = seeds_of(idx); const cs_seed_t &last = sd[nseeds_of(idx) - 1];
				cb[(size_t)i] = sd[0].qbeg; ce[(size_t)i] = last.qbeg + last.len; calt[(size_t)i] = in.chains[c0 + (uint64_t)idx].is_alt != 0;
			}
			keptv.assign((size_t)n, 0); first.assign((size_t)n, -1);
			nonov.clear();                                                       // the chains kept so far, which new ones are compared with
			overlap_scan(o, n, srt.data(), cb.data(), ce.data(), calt.data(), keptv, first, nonov, sc);
#ifdef CS_FLT_SELFCHECK // (tools/sanitize_host.sh builds with it: the scan below against the loop as the reference writes it, on every read)
			{
				std::vector<int> k2((size_t)n, 0), f2((size_t)n, -1), n2;
				overlap_scan_plain(o, n, srt.data(), cb.data(), ce.data(), calt.data(), k2, f2, n2);
				if (k2 != keptv || f2 != first || n2 != nonov) { fprintf(stderr, "cs_chain_filter: overlap_scan differs from the plain loop (read %lld, %d chains)\n", (long long)r, n); abort(); }
			}
#endif
			for (int j : nonov) if (first[(size_t)j] >= 0) keptv[(size_t)first[(size_t)j]] = 1;
			int i = 0, extras = 0;
			for (; i < n; ++i) { if (keptv[(size_t)i] == 0 || keptv[(size_t)i] == 3) continue; if (++extras >= o.max_chain_extend) break; }
			for (; i < n; ++i) if (keptv[(size_t)i] < 3) keptv[(size_t)i] = 0;
			order.clear();
			for (int k = 0; k < n; ++k) if (keptv[(size_t)k]) order.push_back(srt[(size_t)k].idx);
		}
		// ---- mem_flt_chained_seeds
		const double min_l = o.min_chain_weight ? 1.1f * (float)o.min_chain_weight : 5.5f * std::log((double)l_query);   // MEM_HSP_COEF, MEM_MINSC_COEF
		const int min_hsp = (l_query > 0 || o.min_chain_weight) ? (int)(o.a * min_l + .499) : 0;                         // (an empty read has no chains; log(0) must not reach the conversion)
		const bool seed_sw = !(min_l > 0.05f * (float)l_query) && l_query > 0;                                          // MEM_SEEDSW_COEF: not for short reads
		if (seed_sw) { query.resize((size_t)l_query); for (int j = 0; j < l_query; ++j) query[(size_t)j] = cs_base_code_(bases[read_off[r] + (uint64_t)j]); }
		for (int idx : order) {
			cs_chain_t ch = in.chains[c0 + (uint64_t)idx];
			const cs_seed_t *sd = seeds_of(idx); const int ns = nseeds_of(idx);
			int kept_seeds = 0;
			for (int j = 0; j < ns; ++j) {
				const cs_seed_t &s = sd[j];
				int score = s.len;                                 // what mem_chain leaves (comp_seed.cpp:262) when the filter does not run
				if (seed_sw) {
					// mem_seed_sw: the seed and 50 bases either side, on the read and on the reference, clipped to the strand and the contig
					int sw = -1;
					if (s.len < 200) {                              // MEM_SHORT_LEN
						int qb = std::max(s.qbeg - 50, 0), qe = std::min(s.qbeg + s.len + 50, l_query); // MEM_SHORT_EXT
						int64_t rb = std::max<int64_t>(s.rbeg - 50, 0), re = std::min<int64_t>(s.rbeg + s.len + 50, l_pac << 1);
						const int64_t mid = (s.rbeg + s.rbeg + s.len) >> 1;
						if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
						if (!(qe - qb >= 200 || re - rb >= 200)) {
							// bns_fetch_seq (bntseq.c:426-450): clip to the contig that holds `mid`, on the strand of `mid`
							const bool rev = mid >= l_pac;
							const int64_t mid_f = rev ? (l_pac << 1) - 1 - mid : mid;
							const int rid = (int)(std::upper_bound(C.ref.offset.begin(), C.ref.offset.end(), mid_f) - C.ref.offset.begin()) - 1;
							int64_t far_b = C.ref.offset[(size_t)rid], far_e = far_b + C.ref.len[(size_t)rid];
							if (rev) { const int64_t tmp = far_b; far_b = (l_pac << 1) - far_e; far_e = (l_pac << 1) - tmp; }
							rb = std::max(rb, far_b); re = std::min(re, far_e);
							tseq.resize((size_t)std::max<int64_t>(re - rb, 0));
							for (int64_t p = rb; p < re; ++p) tseq[(size_t)(p - rb)] = p >= l_pac ? (uint8_t)(3 - cs_pac_base_(C.pac, (l_pac << 1) - 1 - p)) : cs_pac_base_(C.pac, p);
							sw = striped_sw_score(qe - qb, query.data() + qb, (int)(re - rb), tseq.data(), mat, o.o_del, o.e_del, o.o_ins, o.e_ins, swbuf);
						}
					}
					if (!(sw < 0 || sw >= min_hsp)) continue;      // a short seed in a poor neighbourhood: dropped
					score = sw < 0 ? s.len * o.a : sw;
				}
				out.seeds.push_back(s); out.score.push_back(score); ++kept_seeds;
			}
			ch.n_seeds = kept_seeds;
			out.chains.push_back(ch);
		}
		out.per_read.push_back((uint32_t)order.size());
	}
}

} // namespace

int cs_load_pac_(const char *prefix, int64_t l_pac, std::vector<uint8_t> &pac)
{
	// <prefix>.pac: l_pac / 4 bytes (+1 if l_pac % 4), a zero byte if l_pac % 4 == 0, then l_pac % 4 (bntseq.c:316-324)
	FILE *fp = fopen((std::string(prefix) + ".pac").c_str(), "rb");
	if (!fp) return cs_fail_(CS_EIO, std::string("cannot read ") + prefix + ".pac");
	const size_t need = (size_t)(l_pac >> 2) + 1;
	pac.assign(need + 8, 0);
	const size_t got = fread(pac.data(), 1, need, fp);
	fclose(fp);
	if (got < (size_t)((l_pac + 3) >> 2)) return cs_fail_(CS_EIO, std::string(prefix) + ".pac is truncated");
	return CS_OK;
}

extern "C" void cs_flt_params_default(cs_flt_params_t *p)
{
	if (!p) return;
	// mem_opt_init (comp_seed.cpp:26-58)
	p->min_chain_weight = 0; p->max_chain_extend = 1 << 30; p->max_chain_gap = 10000; p->min_seed_len = 19; p->mask_level = 0.50f; p->drop_ratio = 0.50f;
	p->a = 1; p->b = 4; p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1;
}

extern "C" int cs_chain_filter(cs_chainer_t *c, const cs_flt_params_t *par, const cs_chain_result_t *in, const uint8_t *bases, const uint64_t *read_offsets,
                               int n_threads, cs_chain_result_t *out, const int32_t **cseed_score)
{
	if (!c || !par || !in || !out || (in->n_reads > 0 && (!in->chain_off || !read_offsets)) || (in->n_chains > 0 && (!in->chains || !in->cseed_off)) || (in->n_seeds > 0 && !in->cseeds))
		return cs_fail_(CS_EINVAL, "cs_chain_filter: bad argument");
	if (par->a < 1 || par->b < 0 || par->e_del < 1 || par->e_ins < 1 || par->o_del < 0 || par->o_ins < 0 || par->max_chain_extend < 1) return cs_fail_(CS_EINVAL, "cs_chain_filter: bad parameters");
	const int64_t n = in->n_reads;
	if (in->chains == c->f_chains.data() && in->n_chains) return cs_fail_(CS_EINVAL, "cs_chain_filter: the input is this function's own previous output");
	for (uint64_t k = 0; k < in->n_chains; ++k) if (in->cseed_off[k + 1] <= in->cseed_off[k]) return cs_fail_(CS_EINVAL, "cs_chain_filter: a chain without seeds");
	// the seed test reads the reference: needed only when some read is long enough for it (and then its bases are, too)
	bool need_pac = false;
	double l_seen = -1.;                                            // (a batch has few distinct lengths: one logarithm each, not one per read)
	for (int64_t r = 0; r < n && !need_pac; ++r) {
		const double l = (double)(read_offsets[r + 1] - read_offsets[r]);
		if (l == l_seen) continue;
		if (l > 0 && in->chain_off[r + 1] > in->chain_off[r]) {
			l_seen = l; const double min_l = par->min_chain_weight ? 1.1f * (float)par->min_chain_weight : 5.5f * std::log(l); need_pac = !(min_l > 0.05f * (float)l); }
	}
	if (need_pac) {
		if (!bases) return cs_fail_(CS_EINVAL, "cs_chain_filter: the reads are needed for the seed test of long reads");
		if (c->pac.empty()) { const int rc = cs_load_pac_(c->prefix.c_str(), c->ref.l_pac, c->pac); if (rc != CS_OK) { c->pac.clear(); return rc; } }
	}
	int T = std::max(1, std::min(n_threads, 256));
	if (n < 1024) T = 1;
	// chunks of reads, handed out by a counter (cs_for_chunks_); every chunk's share goes to where the prefix sums over the chunks say
	const int64_t CH = cs_chunk_reads_(n, T), K = (n + CH - 1) / CH;
	std::vector<ReadOut> part((size_t)K);
	cs_for_chunks_(T, K, [&](int64_t k) { filter_range(*c, *par, *in, bases, read_offsets, k * CH, std::min(n, (k + 1) * CH), part[(size_t)k]); });
	std::vector<size_t> cb((size_t)K + 1, 0), sb((size_t)K + 1, 0);
	for (int64_t k = 0; k < K; ++k) { cb[(size_t)k + 1] = cb[(size_t)k] + part[(size_t)k].chains.size(); sb[(size_t)k + 1] = sb[(size_t)k] + part[(size_t)k].seeds.size(); }
	c->f_chains.resize(cb[(size_t)K]); c->f_cseeds.resize(sb[(size_t)K]); c->f_score.resize(sb[(size_t)K]); c->f_chain_off.resize((size_t)n + 1); c->f_cseed_off.resize(cb[(size_t)K] + 1);
	c->f_chain_off[0] = 0; c->f_cseed_off[0] = 0;
	cs_for_chunks_(T, K, [&](int64_t k) {
		const ReadOut &p = part[(size_t)k];
		uint64_t co = cb[(size_t)k], so = sb[(size_t)k];
		int64_t r = k * CH;
		for (uint32_t q : p.per_read) { co += q; c->f_chain_off[(size_t)++r] = co; }
		for (size_t i = 0; i < p.chains.size(); ++i) { so += (uint64_t)p.chains[i].n_seeds; c->f_cseed_off[cb[(size_t)k] + i + 1] = so; }
		if (!p.chains.empty()) memcpy(c->f_chains.data() + cb[(size_t)k], p.chains.data(), p.chains.size() * sizeof(cs_chain_t));
		if (!p.seeds.empty()) { memcpy(c->f_cseeds.data() + sb[(size_t)k], p.seeds.data(), p.seeds.size() * sizeof(cs_seed_t)); memcpy(c->f_score.data() + sb[(size_t)k], p.score.data(), p.score.size() * sizeof(int32_t)); }
	});
	out->n_reads = n; out->n_chains = c->f_chains.size(); out->n_seeds = c->f_cseeds.size();
	out->chain_off = c->f_chain_off.data(); out->chains = c->f_chains.data(); out->cseed_off = c->f_cseed_off.data(); out->cseeds = c->f_cseeds.data();
	if (cseed_score) *cseed_score = c->f_score.data();
	return CS_OK;
}
