// align.cpp -- the extension stage behind chaining (SURVEY 8f row 4 as a whole): chains in, alignment regions out.
//
// Replaces mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319-2237): for every seed of every chain of every read of a batch an
// alignment region is opened, extended to the left and then to the right by banded Smith-Waterman (twice if the first band proved narrow:
// MAX_BAND_TRY 2, comp_seed.cpp:423,1717-1776), and regions whose seed lies inside an earlier region of the read on nearly the same
// diagonal, with no competing seed nearby, are purged afterwards (comp_seed.cpp:2141-2232).  The dynamic programming runs on the GPU
// (extend.hip: the reference's getScores8 / getScores16 / scalarBandedSWAWrapper); everything around it is this host code, stated in its
// own order: windows and pairs are laid out so that NO per-pair sequence copy is made -- the query buffer holds every read once forward
// and once reversed, the target buffer every chain's reference window once forward and once reversed, and a pair is four numbers.
// Results are the reference's regions field by field (tests/golden/aln1/: its own output on five read sets; tests/test_gpu_align.py).
#include "cs_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct cs_aligner {
	cs_refseq_view ref;
	std::vector<uint8_t> pac;             // forward strand, four bases per byte, first base in the top bits (bntseq.c:236-237)
	cs_aln_params_t par{};
	cs_extender_t *ext = nullptr;
	std::vector<uint64_t> reg_off; std::vector<cs_alnreg_t> regs;
	std::vector<uint64_t> dd_off; std::vector<cs_alnreg_t> dd_regs; std::vector<int32_t> dd_ncomp;   // cs_dedup_regions' result
	cs_aln_stats_t st{};
};

namespace {
constexpr int32_t UNSET = -99;            // the reference's H0_ (mapping/macro.h:44): a coordinate that has not been set yet

// longest gap an extension over qlen query bases can afford, capped at twice the band (cal_max_gap, comp_seed.cpp:415-421)
inline int affordable_gap(const cs_aln_params_t &o, int qlen)
{
	const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.), l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = std::max(l_del, l_ins);
	l = std::max(l, 1);
	return std::min(l, o.w << 1);
}
inline int contig_at(const cs_refseq_view &R, int64_t fwd_pos) // (bns_pos2rid, bntseq.c:346-362)
{
	if (fwd_pos >= R.l_pac) return -1;
	return (int)(std::upper_bound(R.offset.begin(), R.offset.end(), fwd_pos) - R.offset.begin()) - 1;
}
inline uint8_t pac_base(const std::vector<uint8_t> &pac, int64_t p) { return cs_pac_base_(pac, p); }
inline uint8_t base_code(uint8_t c) { return cs_base_code_(c); }

struct Job { uint32_t reg; int64_t q_off, t_off; int32_t qlen, tlen; }; // an extension still to run: region (global index) + its pair
} // namespace

extern "C" void cs_aln_params_default(cs_aln_params_t *p)
{
	if (!p) return;
	// mem_opt_init (comp_seed.cpp:26-58)
	p->a = 1; p->b = 4; p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1; p->pen_clip5 = p->pen_clip3 = 5; p->w = 100; p->zdrop = 100;
	p->threads = 16;
}

extern "C" int cs_aligner_create(const char *prefix, int device, const cs_aln_params_t *par, cs_aligner_t **out)
{
	if (!prefix || !out) return cs_fail_(CS_EINVAL, "cs_aligner_create: null argument");
	*out = nullptr;
	cs_aligner *A = new cs_aligner();
	if (par) A->par = *par; else cs_aln_params_default(&A->par);
	const cs_aln_params_t &o = A->par;
	if (o.a < 1 || o.b < 0 || o.e_del < 1 || o.e_ins < 1 || o.o_del < 0 || o.o_ins < 0 || o.w < 1 || o.zdrop < 0) { delete A; return cs_fail_(CS_EINVAL, "cs_aligner_create: bad scoring parameters"); }
	int rc = cs_load_contigs_(prefix, A->ref);
	if (rc != CS_OK) { delete A; return rc; }
	rc = cs_load_pac_(prefix, A->ref.l_pac, A->pac);
	if (rc != CS_OK) { delete A; return rc; }
	cs_ext_params_t xp;
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) xp.mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? o.a : -o.b); // bwa_fill_scmat (bwalib/bwa.c:17-29)
	xp.o_del = o.o_del; xp.e_del = o.e_del; xp.o_ins = o.o_ins; xp.e_ins = o.e_ins; xp.zdrop = o.zdrop; xp.end_bonus = o.pen_clip5; xp.flags = 0;
	// (the end bonus only enters the band limit, ksw.c:402-410; the reference builds one object per side, with pen_clip5 and pen_clip3,
	// comp_seed.cpp:1702-1708: two extenders are kept when the two differ)
	if (device >= 0) { // device -1: an aligner for the host-side passes only (cs_dedup_regions); cs_extend_chains then fails with CS_EDEVICE
		rc = cs_extender_create(device, &xp, &A->ext);
		if (rc != CS_OK) { delete A; return rc; }
	}
	*out = A;
	return CS_OK;
}
extern "C" void cs_aligner_destroy(cs_aligner_t *A)
{
	if (!A) return;
	if (A->ext) cs_extender_destroy(A->ext);
	delete A;
}

extern "C" int cs_extend_chains(cs_aligner_t *A, const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases,
                                const uint64_t *read_offsets, cs_aln_result_t *out)
{
	if (A && !A->ext) return cs_fail_(CS_EDEVICE, "cs_extend_chains: this aligner was created without a device (device -1): the extension runs on the GPU only");
	if (!A || !chains || !out || (chains->n_reads > 0 && (!read_offsets || !chains->chain_off)) || (chains->n_chains > 0 && (!chains->chains || !chains->cseed_off)) ||
	    (chains->n_seeds > 0 && (!chains->cseeds || !bases)))
		return cs_fail_(CS_EINVAL, "cs_extend_chains: bad argument");
	if (A->par.pen_clip5 != A->par.pen_clip3) return cs_fail_(CS_EINVAL, "cs_extend_chains: pen_clip5 != pen_clip3 is not supported yet (one extender, one end bonus)");
	const cs_aln_params_t &o = A->par;
	const cs_refseq_view &R = A->ref;
	const int64_t n = chains->n_reads, l_pac = R.l_pac;
#ifdef CS_ALIGN_TIMING
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[cs_extend_chains] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
#else
	auto lap = [](const char *) {};
#endif
	A->reg_off.assign((size_t)n + 1, 0); A->regs.clear();

	// Host work is per read and independent: T threads take contiguous read ranges; each builds its regions, jobs and target windows into
	// buffers of its own, which are then joined (region and window offsets shifted by what the ranges before it produced).
	const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(o.threads > 0 ? o.threads : 1, 64), n / 256 + 1));
	auto for_ranges = [&](auto &&fn) {
		std::vector<std::thread> th;
		for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t, n * t / T, n * (t + 1) / T); });
		fn(0, (int64_t)0, n / T);
		for (auto &x : th) x.join();
	};
	// ---- the two sequence buffers.  Queries: every read once as codes and once reversed.  A left extension reads the reversed prefix
	// [0, qbeg) = a suffix of the reversed read, a right extension the suffix [qe, len) of the forward read (comp_seed.cpp:1525,1665).
	const uint64_t n_bases = n ? read_offsets[n] : 0;
	std::vector<uint8_t> qbuf((size_t)n_bases * 2 + 8);
	for_ranges([&](int, int64_t r0, int64_t r1) {
		for (int64_t r = r0; r < r1; ++r) {
			const uint64_t b0 = read_offsets[r], len = read_offsets[r + 1] - b0;
			for (uint64_t j = 0; j < len; ++j) { const uint8_t c = base_code(bases[b0 + j]); qbuf[b0 + j] = c; qbuf[n_bases + b0 + (len - 1 - j)] = c; }
		}
	});
	lap("queries (codes, reversed)");
	// Targets: per chain the reference window [w0, w1) the chain's seeds can reach (comp_seed.cpp:1395-1428), forward and reversed.
	struct Meta { int64_t read; const cs_seed_t *seeds; int32_t n_seeds; };   // per region: what the later passes need
	struct Part { std::vector<cs_alnreg_t> regs; std::vector<Meta> meta; std::vector<Job> left, right; std::vector<int32_t> h0_left; std::vector<uint8_t> tbuf; bool bad = false; };
	std::vector<Part> part((size_t)T);
	std::vector<uint32_t> order_flat((size_t)chains->n_seeds);                      // per chain (at its cseed_off): its seeds in the order they were extended (for the purge pass)
	for_ranges([&](int t, int64_t r0, int64_t r1) {
		Part &P = part[(size_t)t];
		for (int64_t r = r0; r < r1; ++r) {
			const int l_query = (int)(read_offsets[r + 1] - read_offsets[r]);
			for (uint64_t ci = chains->chain_off[r]; ci < chains->chain_off[r + 1]; ++ci) {
				const cs_chain_t &c = chains->chains[ci];
				const cs_seed_t *sd = chains->cseeds + chains->cseed_off[ci];
				const int32_t *sc = cseed_score ? cseed_score + chains->cseed_off[ci] : nullptr;
				const int ns = c.n_seeds;
				if (ns <= 0) continue;
				int64_t w0 = l_pac << 1, w1 = 0;
				for (int i = 0; i < ns; ++i) {
					const cs_seed_t &sdi = sd[i];
					const int64_t b = sdi.rbeg - (sdi.qbeg + affordable_gap(o, sdi.qbeg));
					const int tail = l_query - sdi.qbeg - sdi.len;
					const int64_t e = sdi.rbeg + sdi.len + (tail + affordable_gap(o, tail));
					w0 = std::min(w0, b); w1 = std::max(w1, e);
				}
				w0 = std::max<int64_t>(w0, 0); w1 = std::min<int64_t>(w1, l_pac << 1);
				if (w0 < l_pac && l_pac < w1) { if (sd[0].rbeg < l_pac) w1 = l_pac; else w0 = l_pac; } // never across the strands
				{ // clip to the contig of the first seed (bns_fetch_seq, bntseq.c:426-451)
					const int64_t mid = sd[0].rbeg;
					const bool rev = mid >= l_pac;
					const int rid = contig_at(R, rev ? (l_pac << 1) - 1 - mid : mid);
					if (rid < 0 || !(w0 <= mid && mid < w1)) { P.bad = true; return; }
					int64_t far_b = R.offset[(size_t)rid], far_e = far_b + R.len[(size_t)rid];
					if (rev) { const int64_t x = far_b; far_b = (l_pac << 1) - far_e; far_e = (l_pac << 1) - x; }
					w0 = std::max(w0, far_b); w1 = std::min(w1, far_e);
				}
				const int64_t L = w1 - w0, tb0 = (int64_t)P.tbuf.size();
				P.tbuf.resize(P.tbuf.size() + (size_t)L * 2);
				for (int64_t k = 0; k < L; ++k) { // the window's bases: forward strand as stored, reverse strand complemented from the mirror position
					const int64_t p = w0 + k;
					const uint8_t b = p < l_pac ? pac_base(A->pac, p) : (uint8_t)(3 - pac_base(A->pac, (l_pac << 1) - 1 - p));
					P.tbuf[(size_t)(tb0 + k)] = b; P.tbuf[(size_t)(tb0 + L + (L - 1 - k))] = b;
				}
				// seeds by score, highest first, later ones first among equals (ks_introsort over score << 32 | index, walked from the top: comp_seed.cpp:1440-1458)
				uint32_t *ord = order_flat.data() + chains->cseed_off[ci];
				for (int i = 0; i < ns; ++i) ord[(size_t)i] = (uint32_t)i;
				std::sort(ord, ord + ns, [&](uint32_t x, uint32_t y) {
					const int sx = sc ? sc[x] : sd[x].len, sy = sc ? sc[y] : sd[y].len;
					return sx != sy ? sx > sy : x > y;
				});
				const uint64_t rb0 = read_offsets[r];
				for (int k = 0; k < ns; ++k) {
					const cs_seed_t &s = sd[ord[(size_t)k]];
					cs_alnreg_t a; memset(&a, 0, sizeof a);
					a.w = o.w; a.score = a.truesc = -1; a.rid = c.rid; a.frac_rep = c.frac_rep; a.seedlen0 = s.len; a.chain = (int32_t)(ci - chains->chain_off[r]);
					a.rb = a.re = UNSET; a.qb = a.qe = UNSET;
					const uint32_t reg = (uint32_t)P.regs.size();                   // (local: shifted when the ranges are joined)
					if (s.qbeg) { // left: reversed read prefix against the reversed window in front of the seed
						const int64_t tl = s.rbeg - w0;
						Job j = {reg, (int64_t)(n_bases + rb0 + (uint64_t)(l_query - s.qbeg)), tb0 + L + (L - tl), s.qbeg, (int32_t)tl};
						P.left.push_back(j); P.h0_left.push_back(s.len * o.a);
						a.qb = s.qbeg; a.rb = s.rbeg;
					} else { a.score = a.truesc = s.len * o.a; a.qb = 0; a.rb = s.rbeg; }
					if (s.qbeg + s.len != l_query) { // right: the rest of the read against the window behind the seed
						const int qe = s.qbeg + s.len; const int64_t re = s.rbeg + s.len - w0;
						Job j = {reg, (int64_t)(rb0 + (uint64_t)qe), tb0 + re, l_query - qe, (int32_t)(L - re)};
						P.right.push_back(j);
						a.qe = qe; a.re = w0 + re;
					} else { a.qe = l_query; a.re = s.rbeg + s.len; }
					P.regs.push_back(a);
					Meta m = {r, sd, ns}; P.meta.push_back(m);
				}
			}
			A->reg_off[(size_t)r + 1] = P.regs.size();                                  // (local count so far; made global below)
		}
	});
	lap("windows, regions, jobs");
	for (const Part &P : part) if (P.bad) return cs_fail_(CS_EINVAL, "cs_extend_chains: a chain's first seed lies outside the reference");
	// join the ranges
	std::unique_ptr<uint8_t[]> tbuf; size_t tbuf_bytes = 0;                           // (not a vector: nobody needs 10 GB of zeros written first)
	std::vector<Job> left, right; std::vector<Meta> meta; std::vector<int32_t> h0_left;
	{
		size_t nr = 0, nt = 0, nl = 0, nrt = 0;
		for (const Part &P : part) { nr += P.regs.size(); nt += P.tbuf.size(); nl += P.left.size(); nrt += P.right.size(); }
		if (nr >= 0xffffffffull) return cs_fail_(CS_ERANGE, "cs_extend_chains: more than 2^32 regions in one call");
		A->regs.resize(nr); meta.resize(nr); tbuf.reset(new uint8_t[nt + 8]); tbuf_bytes = nt; left.resize(nl); right.resize(nrt); h0_left.resize(nl);
		std::vector<size_t> reg_base((size_t)T + 1, 0), t_base((size_t)T + 1, 0), l_base((size_t)T + 1, 0), r_base((size_t)T + 1, 0);
		for (int t = 0; t < T; ++t) {
			reg_base[(size_t)t + 1] = reg_base[(size_t)t] + part[(size_t)t].regs.size(); t_base[(size_t)t + 1] = t_base[(size_t)t] + part[(size_t)t].tbuf.size();
			l_base[(size_t)t + 1] = l_base[(size_t)t] + part[(size_t)t].left.size(); r_base[(size_t)t + 1] = r_base[(size_t)t] + part[(size_t)t].right.size();
		}
		for_ranges([&](int t, int64_t r0, int64_t r1) { // every range copies its own share to where the prefix sums say
			Part &P = part[(size_t)t];
			const size_t rb = reg_base[(size_t)t], tb = t_base[(size_t)t];
			for (int64_t r = r0; r < r1; ++r) A->reg_off[(size_t)r + 1] += rb;
			if (!P.tbuf.empty()) memcpy(tbuf.get() + tb, P.tbuf.data(), P.tbuf.size());
			for (size_t i = 0; i < P.left.size(); ++i) { Job j = P.left[i]; j.reg += (uint32_t)rb; j.t_off += (int64_t)tb; left[l_base[(size_t)t] + i] = j; h0_left[l_base[(size_t)t] + i] = P.h0_left[i]; }
			for (size_t i = 0; i < P.right.size(); ++i) { Job j = P.right[i]; j.reg += (uint32_t)rb; j.t_off += (int64_t)tb; right[r_base[(size_t)t] + i] = j; }
			if (!P.regs.empty()) { memcpy(A->regs.data() + rb, P.regs.data(), P.regs.size() * sizeof(cs_alnreg_t)); memcpy(meta.data() + rb, P.meta.data(), P.meta.size() * sizeof(Meta)); }
			P = Part();
		});
	}
	lap("join");
	// ---- the dynamic programming, on the GPU: sequences go up once, each band try moves its pairs and results only
	int rc = cs_extender_upload(A->ext, qbuf.data(), qbuf.size(), tbuf_bytes ? tbuf.get() : nullptr, tbuf_bytes);
	if (rc != CS_OK) return rc;
	lap("upload sequences");
	std::vector<cs_ext_pair_t> pairs; std::vector<cs_ext_result_t> res;
	auto run_side = [&](std::vector<Job> &jobs, std::vector<int32_t> &h0, bool is_left, int pen_clip) -> int {
		for (int attempt = 0; attempt < 2 && !jobs.empty(); ++attempt) { // MAX_BAND_TRY (comp_seed.cpp:423)
			const int w = o.w << attempt;
			pairs.resize(jobs.size()); res.resize(jobs.size());
			for (size_t i = 0; i < jobs.size(); ++i) { cs_ext_pair_t p = {(uint64_t)jobs[i].q_off, (uint64_t)jobs[i].t_off, jobs[i].qlen, jobs[i].tlen, h0[i], 0}; pairs[i] = p; }
			const int e = cs_extend_batch_resident(A->ext, (int64_t)jobs.size(), pairs.data(), w, res.data());
			if (e != CS_OK) return e;
			A->st.pairs += jobs.size(); A->st.launches++;
			size_t keep = 0;
			for (size_t i = 0; i < jobs.size(); ++i) {
				cs_alnreg_t &a = A->regs[jobs[i].reg];
				const cs_ext_result_t &x = res[i];
				const int prev = a.score;
				a.score = x.score;
				// settled unless the band may have cut the alignment: same score as before, the path stayed within 3/4 of the band, or no try left
				if (a.score == prev || x.max_off < (w >> 1) + (w >> 2) || attempt == 1) {
					const bool local = x.gscore <= 0 || x.gscore <= a.score - pen_clip;   // clipping beats reaching the end of the read
					if (is_left) {
						if (local) { a.qb -= x.qle; a.rb -= x.tle; a.truesc = a.score; }
						else { a.qb = 0; a.rb -= x.gtle; a.truesc = x.gscore; }
					} else {
						if (local) { a.qe += x.qle; a.re += x.tle; a.truesc += a.score - h0[i]; }
						else { a.qe = (int32_t)(read_offsets[meta[jobs[i].reg].read + 1] - read_offsets[meta[jobs[i].reg].read]); a.re += x.gtle; a.truesc += x.gscore - h0[i]; }
					}
					a.w = std::max(a.w, w);
				} else { jobs[keep] = jobs[i]; h0[keep] = h0[i]; ++keep; A->st.retries++; }
			}
			jobs.resize(keep); h0.resize(keep);
		}
		return CS_OK;
	};
	rc = run_side(left, h0_left, true, o.pen_clip5);
	if (rc != CS_OK) return rc;
	lap("left side (2 tries)");
	std::vector<int32_t> h0_right(right.size());
	for (size_t i = 0; i < right.size(); ++i) h0_right[i] = A->regs[right[i].reg].score;   // the right side starts from what the left side reached (comp_seed.cpp:1917-1922)
	rc = run_side(right, h0_right, false, o.pen_clip3);
	if (rc != CS_OK) return rc;
	lap("right side (2 tries)");

	// ---- seed coverage of the final region: the chain's seeds that lie inside it on both axes (comp_seed.cpp:1758-1766)
	for_ranges([&](int, int64_t r0, int64_t r1) {
		for (size_t g = (size_t)A->reg_off[(size_t)r0]; g < (size_t)A->reg_off[(size_t)r1]; ++g) {
			cs_alnreg_t &a = A->regs[g];
			int cov = 0;
			for (int i = 0; i < meta[g].n_seeds; ++i) {
				const cs_seed_t &t = meta[g].seeds[i];
				if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
			}
			a.seedcov = cov;
		}
	});

	lap("seed coverage");
	// ---- purge (comp_seed.cpp:2141-2232): walking the seeds in the order they were extended, a seed that lies inside an earlier, surviving
	// region of its read, is not much longer than that region's seed, and sits within the band of its diagonal at either end is redundant
	// -- unless a later-ranked seed of its chain overlaps it on another diagonal.  Its region is marked qb = qe = -1.
	std::atomic<uint64_t> n_purged(0);
	for_ranges([&](int, int64_t r0, int64_t r1) {
	uint64_t my_purged = 0;
	for (int64_t r = r0; r < r1; ++r) {
		const int l_query = (int)(read_offsets[r + 1] - read_offsets[r]);
		const size_t g0 = (size_t)A->reg_off[(size_t)r], g1 = (size_t)A->reg_off[(size_t)r + 1];
		size_t g = g0; int kept = 0;
		for (uint64_t ci = chains->chain_off[r]; ci < chains->chain_off[r + 1]; ++ci) {
			const cs_seed_t *sd = chains->cseeds + chains->cseed_off[ci];
			uint32_t *ord = order_flat.data() + chains->cseed_off[ci];
			const int ns = (int)(chains->cseed_off[ci + 1] - chains->cseed_off[ci]);
			// `ord` is descending; the reference indexes the ascending array from the top, k = ns - 1 .. 0, i.e. position ns - 1 - k here
			for (int k = 0; k < ns; ++k, ++g) {
				const cs_seed_t &s = sd[ord[(size_t)k]];
				int seen = 0; bool around = false;
				for (size_t i = g0; i < g1 && seen < kept; ++i) {
					const cs_alnreg_t &p = A->regs[i];
					if (p.qb == -1 && p.qe == -1) continue;
					if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) { ++seen; continue; }
					if (s.len - p.seedlen0 > .1 * l_query) { ++seen; continue; }
					int qd = s.qbeg - p.qb; int64_t rd = s.rbeg - p.rb;
					int gap = affordable_gap(o, (int)(qd < rd ? qd : rd)), w = std::min(gap, p.w);
					if (qd - rd < w && rd - qd < w) { around = true; break; }
					qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
					gap = affordable_gap(o, (int)(qd < rd ? qd : rd)); w = std::min(gap, p.w);
					if (qd - rd < w && rd - qd < w) { around = true; break; }
					++seen;
				}
				if (around) {
					bool rival = false;
					for (int v = k - 1; v >= 0 && !rival; --v) { // seeds ranked above this one that are still in play
						if (ord[(size_t)v] == 0xffffffffu) continue;
						const cs_seed_t &t = sd[ord[(size_t)v]];
						if (t.len < s.len * .95) continue;
						if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) rival = true;
						else if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) rival = true;
					}
					if (!rival) { A->regs[g].qb = A->regs[g].qe = -1; ord[(size_t)k] = 0xffffffffu; ++my_purged; continue; }
				}
				++kept;
			}
		}
	}
	n_purged += my_purged;
	});
	lap("purge");
	A->st.purged += n_purged.load();
	A->st.reads += (uint64_t)n; A->st.regions += A->regs.size();
	out->n_reads = n; out->n_regs = A->regs.size(); out->reg_off = A->reg_off.data(); out->regs = A->regs.data();
	return CS_OK;
}

extern "C" int cs_dedup_regions(cs_aligner_t *A, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets,
                                cs_aln_result_t *out, const int32_t **n_comp)
{
	if (!A || !par || !regs || !out || (regs->n_reads > 0 && (!regs->reg_off || !read_offsets)) || (regs->n_regs > 0 && (!regs->regs || !bases))) return cs_fail_(CS_EINVAL, "cs_dedup_regions: bad argument");
	if (regs->regs == A->dd_regs.data() && regs->n_regs) return cs_fail_(CS_EINVAL, "cs_dedup_regions: the input is this function's own previous output");
	if (par->max_chain_gap < 0 || !(par->mask_level_redun > 0.f)) return cs_fail_(CS_EINVAL, "cs_dedup_regions: bad parameters");
	const cs_aligner_core core = {&A->ref, &A->pac, &A->par};
	const int rc = cs_dedup_regions_(core, par, regs, bases, read_offsets, A->dd_off, A->dd_regs, A->dd_ncomp);
	if (rc != CS_OK) return rc;
	out->n_reads = regs->n_reads; out->n_regs = A->dd_regs.size(); out->reg_off = A->dd_off.data(); out->regs = A->dd_regs.data();
	if (n_comp) *n_comp = A->dd_ncomp.data();
	return CS_OK;
}

extern "C" int cs_aligner_stats(const cs_aligner_t *A, cs_aln_stats_t *st)
{
	if (!A || !st) return cs_fail_(CS_EINVAL, "null argument");
	*st = A->st;
	return CS_OK;
}
