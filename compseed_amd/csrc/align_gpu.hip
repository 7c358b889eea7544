// align_gpu.hip -- cs_extend_chains = mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319-2237) with everything per chain, per seed
// and per read on the GPU.  The first version of this driver built the reference windows, the regions and the job lists on host threads
// around the extension kernels (30 M extensions: 0.8 s of kernels inside 6 s of host work); here the chains and the
// reads go up once, and what comes back are the regions:
//   chain_read_kernel   which read a chain belongs to
//   query_kernel        every read as codes, forward and reversed (a left extension reads the reversed prefix, comp_seed.cpp:1525)
//   window_kernel       per chain the reference window its seeds can reach with the gap they can afford (cal_max_gap), clipped to the
//                       strand and to the contig of the first seed (bns_fetch_seq) -> w0, length; a scan gives every window its place
//   fill_kernel         the window's bases from the 2-bit packed reference, forward and reversed (one wave per chain)
//   region_kernel       per chain: its seeds ranked by score (highest first, later ones first among equals: comp_seed.cpp:1440-1458), one
//                       region per seed, its left / right extension appended to the pair lists (a pair's `reserved` word = its region)
//   cs_extend_batch_device  the dynamic programming (extend.hip), band w, then 2w for the pairs apply_kernel sends back (MAX_BAND_TRY 2)
//   apply_kernel        a side's results into the regions: clipped or to the end of the read, truesc, the band used; retry list
//   seedcov_kernel, purge_kernel   comp_seed.cpp:1758-1766 and :2141-2232 (one thread per read walks its regions in extension order)
// Same results as the host driver it replaces, field by field (tests/test_gpu_align.py against the reference's own regions).
#include "cs_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#define HIP_TRYA(expr)                                                                              \
	do {                                                                                            \
		hipError_t e__ = (expr);                                                                    \
		if (e__ != hipSuccess) {                                                                    \
			(void)hipGetLastError();                                                                \
			return cs_fail_(e__ == hipErrorOutOfMemory ? CS_ENOMEM : CS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
		}                                                                                           \
	} while (0)

namespace csa {
constexpr int32_t UNSET = -99;            // the reference's H0_ (mapping/macro.h:44): a coordinate that has not been set yet

struct Args {
	const uint64_t *chain_off, *cseed_off, *read_off; const cs_chain_t *chains; const cs_seed_t *cseeds; const int32_t *score; const uint8_t *bases;
	int64_t n_reads, n_chains, n_seeds, l_pac; uint64_t n_bases;
	const uint8_t *pac; const int64_t *ctg_off; const int32_t *ctg_len; int32_t n_ctg;
	cs_aln_params_t o;
	uint32_t *chain_read; int64_t *w0; uint64_t *wlen2, *tb0;   // per chain: window start, 2 x length (scanned into tb0)
	uint8_t *qbuf, *tbuf; uint32_t *ord; cs_alnreg_t *regs; uint32_t *reg_ci;
	cs_ext_pair_t *lp, *rp;
	int32_t small_chain, light_max, purge_cap;   // SMALL_CHAIN / 64 / PURGE_CAP unless cs_aln_params_t.flags says otherwise (A/B tests)
	uint32_t *big, *big_reads;             // chains of more than SMALL_CHAIN seeds / reads of more than 64 regions: lists for the *_big kernels
	unsigned long long *ctr;               // [0] left pairs [1] right pairs [2] retries [3] bad chains [4] purged [5] long chains [6] long reads
};

__device__ __forceinline__ int affordable_gap(const cs_aln_params_t &o, int qlen) // cal_max_gap (comp_seed.cpp:415-421)
{
	const int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.), l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < (o.w << 1) ? l : (o.w << 1);
}
__device__ __forceinline__ uint8_t pac_base(const uint8_t *pac, int64_t p) { return (uint8_t)((pac[p >> 2] >> ((~p & 3) << 1)) & 3); }
__device__ __forceinline__ uint8_t base_code(uint8_t c) // nst_nt4_table (bntseq.c:46-63); bytes 0..4 are codes already
{
	if (c <= 4) return c;
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

__global__ void chain_read_kernel(const Args A)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.n_reads; r += (int64_t)gridDim.x * blockDim.x)
		for (uint64_t ci = A.chain_off[r]; ci < A.chain_off[r + 1]; ++ci) A.chain_read[ci] = (uint32_t)r;
}
__global__ void query_kernel(const Args A)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < A.n_bases; b += (uint64_t)gridDim.x * blockDim.x) {
		// the read of base b: binary search in the offsets
		int64_t lo = 0, hi = A.n_reads;
		while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (A.read_off[mid] <= b) lo = mid; else hi = mid; }
		const uint64_t b0 = A.read_off[lo], len = A.read_off[lo + 1] - b0, j = b - b0;
		const uint8_t c = base_code(A.bases[b]);
		A.qbuf[b] = c; A.qbuf[A.n_bases + b0 + (len - 1 - j)] = c;
	}
}
__global__ void window_kernel(const Args A)
{
	const cs_aln_params_t &o = A.o;
	for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < A.n_chains; ci += (int64_t)gridDim.x * blockDim.x) {
		const cs_seed_t *sd = A.cseeds + A.cseed_off[ci];
		const int ns = (int)(A.cseed_off[ci + 1] - A.cseed_off[ci]);
		A.w0[ci] = 0; A.wlen2[ci] = 0;
		if (ns <= 0) continue;
		const uint32_t r = A.chain_read[ci];
		const int l_query = (int)(A.read_off[r + 1] - A.read_off[r]);
		const int64_t l_pac = A.l_pac;
		int64_t w0 = l_pac << 1, w1 = 0;
		for (int i = 0; i < ns; ++i) {
			const cs_seed_t s = sd[i];
			const int64_t b = s.rbeg - (s.qbeg + affordable_gap(o, s.qbeg));
			const int tail = l_query - s.qbeg - s.len;
			const int64_t e = s.rbeg + s.len + (tail + affordable_gap(o, tail));
			w0 = w0 < b ? w0 : b; w1 = w1 > e ? w1 : e;
		}
		w0 = w0 > 0 ? w0 : 0; w1 = w1 < (l_pac << 1) ? w1 : (l_pac << 1);
		const int64_t mid = sd[0].rbeg;
		if (w0 < l_pac && l_pac < w1) { if (mid < l_pac) w1 = l_pac; else w0 = l_pac; } // never across the strands
		// clip to the contig of the first seed (bns_fetch_seq, bntseq.c:426-451)
		const bool rev = mid >= l_pac;
		const int64_t mid_f = rev ? (l_pac << 1) - 1 - mid : mid;
		int lo = 0, hi = A.n_ctg;                                         // last contig whose offset is <= mid_f
		while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (A.ctg_off[m] <= mid_f) lo = m; else hi = m; }
		if (mid_f < 0 || mid_f >= l_pac || !(w0 <= mid && mid < w1)) { atomicAdd(A.ctr + 3, 1ull); continue; }
		int64_t far_b = A.ctg_off[lo], far_e = far_b + A.ctg_len[lo];
		if (rev) { const int64_t x = far_b; far_b = (l_pac << 1) - far_e; far_e = (l_pac << 1) - x; }
		w0 = w0 > far_b ? w0 : far_b; w1 = w1 < far_e ? w1 : far_e;
		A.w0[ci] = w0; A.wlen2[ci] = w1 > w0 ? (uint64_t)(w1 - w0) * 2 : 0;
	}
}
// one wave per chain: the window's bases, forward strand as stored, reverse strand complemented from the mirror position; then reversed
__global__ void fill_kernel(const Args A)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t ci = wave; ci < A.n_chains; ci += n_waves) {
		const int64_t L = (int64_t)(A.wlen2[ci] >> 1), w0 = A.w0[ci];
		uint8_t *t = A.tbuf + A.tb0[ci];
		for (int64_t k = lane; k < L; k += 64) {
			const int64_t p = w0 + k;
			const uint8_t b = p < A.l_pac ? pac_base(A.pac, p) : (uint8_t)(3 - pac_base(A.pac, (A.l_pac << 1) - 1 - p));
			t[k] = b; t[L + (L - 1 - k)] = b;
		}
	}
}
// ---- regions.  A chain's seeds are extended in the order of their scores, highest first, later ones first among equals (comp_seed.cpp:1406-1411);
// that order is total, so a seed's place is the number of seeds that rank before it and every seed can be placed on its own.
constexpr int SMALL_CHAIN = 8;
__device__ __forceinline__ int seed_rank(const cs_seed_t *sd, const int32_t *sc, int ns, int i)
{
	const int si = sc ? sc[i] : sd[i].len;
	int rk = 0;
	for (int y = 0; y < ns; ++y) { const int sy = sc ? sc[y] : sd[y].len; rk += (sy > si || (sy == si && y > i)) ? 1 : 0; }
	return rk;
}
struct ChainCtx { int64_t ci, w0, L, tb0; uint64_t s0, rb0; int32_t l_query, rid, chain; float frac_rep; };
__device__ __forceinline__ ChainCtx chain_ctx(const Args &A, int64_t ci)
{
	ChainCtx X;
	const cs_chain_t c = A.chains[ci];
	const uint32_t r = A.chain_read[ci];
	X.ci = ci; X.s0 = A.cseed_off[ci]; X.rb0 = A.read_off[r]; X.l_query = (int32_t)(A.read_off[r + 1] - X.rb0);
	X.w0 = A.w0[ci]; X.L = (int64_t)(A.wlen2[ci] >> 1); X.tb0 = (int64_t)A.tb0[ci];
	X.rid = c.rid; X.frac_rep = c.frac_rep; X.chain = (int32_t)(ci - (int64_t)A.chain_off[r]);
	return X;
}
// seed i of the chain, k-th in the order of extension: its region and its (at most two) pairs, at the slots the caller reserved
__device__ __forceinline__ void emit_region(const Args &A, const ChainCtx &X, const cs_seed_t &s, int i, int k, bool left, bool right, uint64_t lslot, uint64_t rslot)
{
	const cs_aln_params_t &o = A.o;
	const uint64_t g = X.s0 + (uint64_t)k;
	cs_alnreg_t a; memset(&a, 0, sizeof a);
	a.w = o.w; a.score = a.truesc = -1; a.rid = X.rid; a.frac_rep = X.frac_rep; a.seedlen0 = s.len; a.chain = X.chain;
	a.rb = a.re = UNSET; a.qb = a.qe = UNSET;
	if (left) { // reversed read prefix against the reversed window in front of the seed
		const int64_t tl = s.rbeg - X.w0;
		const cs_ext_pair_t p = {(uint64_t)(A.n_bases + X.rb0 + (uint64_t)(X.l_query - s.qbeg)), (uint64_t)(X.tb0 + X.L + (X.L - tl)), s.qbeg, (int32_t)tl, s.len * o.a, (int32_t)g};
		A.lp[lslot] = p;
		a.qb = s.qbeg; a.rb = s.rbeg;
	} else { a.score = a.truesc = s.len * o.a; a.qb = 0; a.rb = s.rbeg; }
	if (right) { // the rest of the read against the window behind the seed
		const int qe = s.qbeg + s.len; const int64_t re = s.rbeg + s.len - X.w0;
		const cs_ext_pair_t p = {(uint64_t)(X.rb0 + (uint64_t)qe), (uint64_t)(X.tb0 + re), X.l_query - qe, (int32_t)(X.L - re), 0, (int32_t)g};
		A.rp[rslot] = p;
		a.qe = qe; a.re = X.w0 + re;
	} else { a.qe = X.l_query; a.re = s.rbeg + s.len; }
	A.regs[g] = a; A.reg_ci[g] = (uint32_t)X.ci; A.ord[g] = (uint32_t)i;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t &total)
{
	uint32_t x = v;
	for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
	total = __shfl(x, 63);
	return x - v;
}
// a chain per lane for the chains of up to SMALL_CHAIN seeds (nearly all); the pair slots of a wave's 64 chains come from one atomic per
// side (one per pair made the two counters the kernel's clock); longer chains go on a list for region_big_kernel
__global__ void region_kernel(const Args A)
{
	const int lane = threadIdx.x & 63;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < A.n_chains; base += stride) {
		const int64_t ci = base + lane;
		int ns = 0; bool mine = false; uint32_t nl = 0, nr = 0;
		ChainCtx X; const cs_seed_t *sd = nullptr;
		if (ci < A.n_chains) {
			const uint64_t s0 = A.cseed_off[ci];
			ns = (int)(A.cseed_off[ci + 1] - s0);
			if (ns > A.small_chain) A.big[atomicAdd(A.ctr + 5, 1ull)] = (uint32_t)ci;
			else if (ns > 0) {
				mine = true; X = chain_ctx(A, ci); sd = A.cseeds + s0;
				for (int i = 0; i < ns; ++i) { const cs_seed_t s = sd[i]; nl += s.qbeg != 0; nr += s.qbeg + s.len != X.l_query; }
			}
		}
		uint32_t tl, tr;
		const uint32_t el = wave_excl_scan(nl, lane, tl), er = wave_excl_scan(nr, lane, tr);
		unsigned long long bl = 0, br = 0;
		if (lane == 0) { if (tl) bl = atomicAdd(A.ctr + 0, (unsigned long long)tl); if (tr) br = atomicAdd(A.ctr + 1, (unsigned long long)tr); }
		bl = __shfl(bl, 0); br = __shfl(br, 0);
		if (mine) {
			uint64_t ls = bl + el, rs = br + er;
			const int32_t *sc = A.score ? A.score + X.s0 : nullptr;
			for (int i = 0; i < ns; ++i) {
				const cs_seed_t s = sd[i];
				const bool left = s.qbeg != 0, right = s.qbeg + s.len != X.l_query;
				emit_region(A, X, s, i, seed_rank(sd, sc, ns, i), left, right, ls, rs);
				ls += left; rs += right;
			}
		}
	}
}
// one wave per long chain (a repeat's hundreds of seeds on one diagonal band): a seed per lane, 64 at a time
__global__ void region_big_kernel(const Args A)
{
	const int lane = threadIdx.x & 63;
	const uint64_t lt = (1ull << lane) - 1ull;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	const uint64_t n_big = A.ctr[5];
	for (uint64_t e = wave; e < n_big; e += n_waves) {
		const int64_t ci = (int64_t)A.big[e];
		const ChainCtx X = chain_ctx(A, ci);
		const cs_seed_t *sd = A.cseeds + X.s0;
		const int32_t *sc = A.score ? A.score + X.s0 : nullptr;
		const int ns = (int)(A.cseed_off[ci + 1] - X.s0);
		for (int i0 = 0; i0 < ns; i0 += 64) {
			const int i = i0 + lane;
			cs_seed_t s = {0, 0, 0}; int k = 0; bool left = false, right = false;
			if (i < ns) { s = sd[i]; k = seed_rank(sd, sc, ns, i); left = s.qbeg != 0; right = s.qbeg + s.len != X.l_query; }
			const uint64_t ml = __ballot(left), mr = __ballot(right);
			unsigned long long bl = 0, br = 0;
			if (lane == 0) { if (ml) bl = atomicAdd(A.ctr + 0, (unsigned long long)__popcll(ml)); if (mr) br = atomicAdd(A.ctr + 1, (unsigned long long)__popcll(mr)); }
			bl = __shfl(bl, 0); br = __shfl(br, 0);
			if (i < ns) emit_region(A, X, s, i, k, left, right, bl + (uint64_t)__popcll(ml & lt), br + (uint64_t)__popcll(mr & lt));
		}
	}
}
__global__ void right_h0_kernel(cs_ext_pair_t *rp, uint64_t n, const cs_alnreg_t *regs) // the right side starts from what the left side reached (comp_seed.cpp:1917-1922)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) rp[i].h0 = regs[(uint32_t)rp[i].reserved].score;
}
// a side's results into the regions; pairs whose alignment may have been cut by the band go to `retry` (same pair, next try)
__global__ void apply_kernel(const Args A, const cs_ext_pair_t *pairs, const cs_ext_result_t *res, uint64_t n, int w, int attempt, int is_left, int pen_clip, cs_ext_pair_t *retry)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const cs_ext_pair_t pr = pairs[i];
		const uint32_t g = (uint32_t)pr.reserved;
		cs_alnreg_t a = A.regs[g];
		const cs_ext_result_t x = res[i];
		const int prev = a.score;
		a.score = x.score;
		// settled unless the band may have cut the alignment: same score as before, the path stayed within 3/4 of the band, or no try left
		if (a.score == prev || x.max_off < (w >> 1) + (w >> 2) || attempt == 1) {
			const bool local = x.gscore <= 0 || x.gscore <= a.score - pen_clip;   // clipping beats reaching the end of the read
			if (is_left) {
				if (local) { a.qb -= x.qle; a.rb -= x.tle; a.truesc = a.score; }
				else { a.qb = 0; a.rb -= x.gtle; a.truesc = x.gscore; }
			} else {
				if (local) { a.qe += x.qle; a.re += x.tle; a.truesc += a.score - pr.h0; }
				else { const uint32_t r = A.chain_read[A.reg_ci[g]]; a.qe = (int32_t)(A.read_off[r + 1] - A.read_off[r]); a.re += x.gtle; a.truesc += x.gscore - pr.h0; }
			}
			a.w = a.w > w ? a.w : w;
		} else retry[atomicAdd(A.ctr + 2, 1ull)] = pr;
		A.regs[g] = a;
	}
}
__global__ void seedcov_kernel(const Args A) // the chain's seeds that lie inside the final region on both axes (comp_seed.cpp:1758-1766)
{
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < A.n_seeds; g += (int64_t)gridDim.x * blockDim.x) {
		const uint32_t ci = A.reg_ci[g];
		const cs_seed_t *sd = A.cseeds + A.cseed_off[ci];
		const int ns = (int)(A.cseed_off[ci + 1] - A.cseed_off[ci]);
		cs_alnreg_t a = A.regs[g];
		int cov = 0;
		for (int i = 0; i < ns; ++i) { const cs_seed_t t = sd[i]; if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len; }
		A.regs[g].seedcov = cov;
	}
}
// comp_seed.cpp:2141-2232: walking a read's seeds in the order they were extended, a seed that lies inside an earlier, surviving region of
// the read, is not much longer than that region's seed and sits within the band of its diagonal at either end is redundant -- unless a
// higher-ranked seed of its chain that is still in play overlaps it on another diagonal.  Its region is marked qb = qe = -1.
// The walk over a read's regions is sequential (a region's fate decides about the later ones), but both tests are "is there one among
// the earlier ...": the lanes of a wave hold the earlier regions / the higher-ranked seeds and a ballot answers.
struct PReg { int64_t rb, re; int32_t qb, qe, seedlen0, w; };
template <class GapF> // gap_of(x) = affordable_gap(o, x); x is a distance inside the read here, 0 .. l_query
__device__ __forceinline__ bool purge_around(const cs_seed_t &s, const PReg &p, int l_query, GapF gap_of)
{
	if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) return false;
	if (s.len - p.seedlen0 > .1 * l_query) return false;
	int qd = s.qbeg - p.qb; int64_t rd = s.rbeg - p.rb;
	int gap = gap_of((int)(qd < rd ? qd : rd)), w = gap < p.w ? gap : p.w;
	if (qd - rd < w && rd - qd < w) return true;
	qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
	gap = gap_of((int)(qd < rd ? qd : rd)); w = gap < p.w ? gap : p.w;
	return qd - rd < w && rd - qd < w;
}
__device__ __forceinline__ bool purge_rival(const cs_seed_t &s, const cs_seed_t &t) // t: a seed ranked above s in its chain
{
	if (t.len < s.len * .95) return false;
	if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) return true;
	return t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg;
}
__device__ __forceinline__ void purge_load(const Args &A, uint64_t g0, uint64_t g, PReg &p, cs_seed_t &s, int &first)
{
	const uint64_t s0 = A.cseed_off[A.reg_ci[g]];
	first = (int)(s0 - g0);                                         // the read's regions [first, g) are the higher-ranked seeds of g's chain
	s = A.cseeds[s0 + A.ord[g]];
	const cs_alnreg_t a = A.regs[g];
	p.rb = a.rb; p.re = a.re; p.qb = a.qb; p.qe = a.qe; p.seedlen0 = a.seedlen0; p.w = a.w;
}
// reads of up to 64 regions (all but a few in a thousand): a wave per read, a region per lane, everything in registers -- the seed in
// question comes from its lane by a shuffle, which regions are still there is a mask.  Longer reads go on a list for purge_big_kernel.
__global__ void purge_kernel(const Args A)
{
	const cs_aln_params_t &o = A.o;
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	unsigned long long my = 0;
	for (int64_t r = wave; r < A.n_reads; r += n_waves) {
		const uint64_t g0 = A.cseed_off[A.chain_off[r]], g1 = A.cseed_off[A.chain_off[r + 1]];
		if (g1 - g0 < 2) continue;                                  // (a read's first region has nothing in front of it)
		if (g1 - g0 > (uint64_t)A.light_max) { if (lane == 0) A.big_reads[atomicAdd(A.ctr + 6, 1ull)] = (uint32_t)r; continue; }
		const int G = (int)(g1 - g0), l_query = (int)(A.read_off[r + 1] - A.read_off[r]);
		PReg p = {0, 0, 0, 0, 0, 0}; cs_seed_t s = {0, 0, 0}; int first = 0;
		if (lane < G) purge_load(A, g0, g0 + (uint64_t)lane, p, s, first);
		const uint64_t all = G == 64 ? ~0ull : (1ull << G) - 1ull;
		uint64_t alive = all;
		for (int g = 1; g < G; ++g) {
			cs_seed_t sg; sg.rbeg = (int64_t)__shfl((long long)s.rbeg, g); sg.qbeg = __shfl(s.qbeg, g); sg.len = __shfl(s.len, g);
			const int fg = __shfl(first, g);
			const bool before = lane < g && ((alive >> lane) & 1ull);
			if (__ballot(before && purge_around(sg, p, l_query, [&](int x) { return affordable_gap(o, x); })) == 0) continue;
			if (__ballot(before && lane >= fg && purge_rival(sg, s)) != 0) continue;
			alive &= ~(1ull << g);
		}
		const uint64_t gone = all & ~alive;
		if ((gone >> lane) & 1ull) { A.regs[g0 + (uint64_t)lane].qb = -1; A.regs[g0 + (uint64_t)lane].qe = -1; }
		if (lane == 0) my += (unsigned long long)__popcll(gone);
	}
	if (lane == 0 && my) atomicAdd(A.ctr + 4, my);
}
// a read of any size straight from HBM (the fallback of purge_big_kernel): which regions are purged so far is a byte array that is read
// around the L1 (volatile) -- lane 0 writes, the whole wave reads in the next step.  Every step is a chain of HBM round trips.
__device__ void purge_read_global(const Args &A, int64_t r, uint8_t *pflag, int lane, unsigned long long &my)
{
	const cs_aln_params_t &o = A.o;
	volatile uint8_t *pf = pflag;
	const int l_query = (int)(A.read_off[r + 1] - A.read_off[r]);
	const uint64_t g0 = A.cseed_off[A.chain_off[r]];
	uint64_t g = g0;
	for (uint64_t ci = A.chain_off[r]; ci < A.chain_off[r + 1]; ++ci) {
		const uint64_t s0 = A.cseed_off[ci];
		const cs_seed_t *sd = A.cseeds + s0;
		const uint32_t *ord = A.ord + s0;
		const int ns = (int)(A.cseed_off[ci + 1] - s0);
		for (int k = 0; k < ns; ++k, ++g) {
			const cs_seed_t s = sd[ord[k]];
			bool around = false;
			for (uint64_t base = g0; base < g && !around; base += 64) {
				const uint64_t i = base + (uint64_t)lane;
				bool hit = false;
				if (i < g && !pf[i]) { const cs_alnreg_t a = A.regs[i]; const PReg p = {a.rb, a.re, a.qb, a.qe, a.seedlen0, a.w}; hit = purge_around(s, p, l_query, [&](int x) { return affordable_gap(o, x); }); }
				around = __ballot(hit) != 0;
			}
			if (!around) continue;
			bool rival = false;
			for (int base = 0; base < k && !rival; base += 64) {
				const int v = base + lane;
				const bool hit = v < k && !pf[s0 + (uint64_t)v] && purge_rival(s, sd[ord[v]]);
				rival = __ballot(hit) != 0;
			}
			if (rival) continue;
			if (lane == 0) { pf[g] = 1; A.regs[g].qb = -1; A.regs[g].qe = -1; ++my; }
			__threadfence();
		}
	}
}
// the reads with more than 64 regions (reads inside repeats: hundreds to thousands of regions, and the walk is quadratic): a workgroup of eight
// waves per read, the read's regions and seeds staged in the LDS -- a step of the walk costs LDS round trips instead of HBM ones, 512 earlier
// regions are looked at per round, the affordable gap (two divisions in double precision) comes from a table over the read's length, both
// tests of a step are answered with ONE barrier (every wave casts its two votes, double-buffered), and the reads are handed out by a ticket
// counter (a few reads of 2,000 regions are most of the work).  All 160 KB of a CU's LDS: one read per CU at a time.
constexpr int PURGE_CAP = 3000, PURGE_GAPS = 1024;
#ifndef CS_PURGE_THREADS
#define CS_PURGE_THREADS 512
#endif
constexpr int PB = CS_PURGE_THREADS;   // threads of purge_big_kernel's workgroup
__global__ __launch_bounds__(PB) void purge_big_kernel(const Args A, uint8_t *pflag)
{
	__shared__ PReg l_reg[PURGE_CAP];
	__shared__ cs_seed_t l_seed[PURGE_CAP];
	__shared__ int32_t l_first[PURGE_CAP];
	__shared__ uint8_t l_alive[PURGE_CAP];
	__shared__ int32_t l_gap[PURGE_GAPS];
	__shared__ unsigned long long l_ticket;
	__shared__ uint32_t l_vote[2][PB / 64];
	const cs_aln_params_t &o = A.o;
	const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t n_big = A.ctr[6];
	unsigned long long my = 0;
	for (int x = tid; x < PURGE_GAPS; x += PB) l_gap[x] = affordable_gap(o, x);
	for (;;) {
		__syncthreads();                                            // (the previous read's last LDS reads are done)
		if (tid == 0) l_ticket = atomicAdd(A.ctr + 7, 1ull);
		__syncthreads();
		const uint64_t e = l_ticket;
		if (e >= n_big) break;
		const int64_t r = (int64_t)A.big_reads[e];
		const uint64_t g0 = A.cseed_off[A.chain_off[r]], g1 = A.cseed_off[A.chain_off[r + 1]];
		if (g1 - g0 > (uint64_t)A.purge_cap) { if (tid < 64) purge_read_global(A, r, pflag, lane, my); continue; }
		const int G = (int)(g1 - g0), l_query = (int)(A.read_off[r + 1] - A.read_off[r]);
		auto gap_of = [&](int x) { return (unsigned)x < (unsigned)PURGE_GAPS ? l_gap[x] : affordable_gap(o, x); };
		for (int i = tid; i < G; i += PB) { PReg p; cs_seed_t s; int first; purge_load(A, g0, g0 + (uint64_t)i, p, s, first); l_reg[i] = p; l_seed[i] = s; l_first[i] = first; l_alive[i] = 1; }
		__syncthreads();
		for (int g = 1; g < G; ++g) {
			const cs_seed_t sg = l_seed[g];
			bool around = false, rival = false;
			for (int i = tid; i < g && !around; i += PB) around = l_alive[i] && purge_around(sg, l_reg[i], l_query, gap_of);
			for (int v = l_first[g] + tid; v < g && !rival; v += PB) rival = l_alive[v] && purge_rival(sg, l_seed[v]);
			const uint32_t vote = (__ballot(around) ? 1u : 0u) | (__ballot(rival) ? 2u : 0u);
			if (lane == 0) l_vote[g & 1][wv] = vote;
			__syncthreads();
			uint32_t all = 0;
			for (int w = 0; w < PB / 64; ++w) all |= l_vote[g & 1][w];
			if (all != 1u) continue;                                // not inside an earlier region, or defended by a rival
			if (lane == 0) l_alive[g] = 0;                          // (every wave for itself: its own later reads follow its own write)
			if (tid == 0) ++my;
		}
		for (int i = tid; i < G; i += PB) if (!l_alive[i]) { A.regs[g0 + (uint64_t)i].qb = -1; A.regs[g0 + (uint64_t)i].qe = -1; }
	}
	if (lane == 0 && my) atomicAdd(A.ctr + 4, my);
}
} // namespace csa

namespace {
struct Buf { void *p = nullptr; size_t cap = 0; };
int ensure(Buf &b, size_t bytes)
{
	if (bytes <= b.cap) return CS_OK;
	if (b.p) (void)hipFree(b.p);
	b.p = nullptr; b.cap = 0;
	const size_t want = bytes + bytes / 8 + 256;
	HIP_TRYA(hipMalloc(&b.p, want));
	b.cap = want;
	return CS_OK;
}
enum { B_CHAIN_OFF, B_CSEED_OFF, B_READ_OFF, B_CHAINS, B_CSEEDS, B_SCORE, B_BASES, B_PAC, B_CTG_OFF, B_CTG_LEN, B_CHAIN_READ, B_W0, B_WLEN2, B_TB0, B_QBUF, B_TBUF, B_ORD, B_REGS,
       B_REG_CI, B_LP, B_RP, B_RETRY, B_RES, B_CTR, B_SCAN, B_PFLAG, B_BIG, B_BIG_READS, B_COUNT };
} // namespace

struct cs_aligner_gpu { int device = 0, n_cu = 256; hipStream_t s = nullptr; Buf b[B_COUNT]; bool pac_up = false; unsigned long long *h_ctr = nullptr; };

void cs_aligner_gpu_release_(cs_aligner_gpu *g)
{
	if (!g) return;
	(void)hipSetDevice(g->device);
	if (g->s) (void)hipStreamSynchronize(g->s);
	for (Buf &b : g->b) if (b.p) (void)hipFree(b.p);
	if (g->h_ctr) (void)hipHostFree(g->h_ctr);
	if (g->s) (void)hipStreamDestroy(g->s);
	delete g;
}

int cs_extend_chains_gpu_(cs_aligner_gpu **gp, int device, cs_extender_t *ext, const cs_refseq_view &R, const std::vector<uint8_t> &pac, const cs_aln_params_t &o,
                          const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases, const uint64_t *read_offsets,
                          std::vector<uint64_t> &reg_off, std::vector<cs_alnreg_t> &regs, cs_aln_stats_t &st)
{
	HIP_TRYA(hipSetDevice(device));
	if (!*gp) {
		cs_aligner_gpu *g = new cs_aligner_gpu(); g->device = device; *gp = g;
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) g->n_cu = prop.multiProcessorCount;
		HIP_TRYA(hipStreamCreateWithFlags(&g->s, hipStreamNonBlocking));
		HIP_TRYA(hipHostMalloc((void **)&g->h_ctr, 8 * sizeof(unsigned long long), hipHostMallocDefault));
	}
	cs_aligner_gpu &G = **gp;
	hipStream_t s = G.s;
#ifdef CS_ALIGN_TIMING
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { (void)hipStreamSynchronize(s); const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[cs_extend_chains] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
#else
	auto lap = [](const char *) {};
#endif
	const int64_t n = chains->n_reads, nc = (int64_t)chains->n_chains, ns = (int64_t)chains->n_seeds;
	reg_off.assign((size_t)n + 1, 0);                              // (regs keeps its size from call to call: growing a vector by 200 MB of zeroes took 6 ms per million reads)
	for (int64_t r = 0; r < n; ++r) reg_off[(size_t)r + 1] = chains->cseed_off[chains->chain_off[r + 1]];
	st.reads += (uint64_t)n;
	if (ns == 0) { regs.clear(); return CS_OK; }
	if (ns >= 0x7fffffffll || nc >= 0xffffffffll) return cs_fail_(CS_ERANGE, "cs_extend_chains: more than 2^31 regions in one call");
	const uint64_t n_bases = read_offsets[n];
	auto up = [&](int which, const void *src, size_t bytes) -> int {
		if (int rc = ensure(G.b[which], bytes + 64)) return rc;
		if (bytes) HIP_TRYA(hipMemcpyAsync(G.b[which].p, src, bytes, hipMemcpyHostToDevice, s));
		return CS_OK;
	};
	if (int rc = up(B_CHAIN_OFF, chains->chain_off, ((size_t)n + 1) * 8)) return rc;
	if (int rc = up(B_CSEED_OFF, chains->cseed_off, ((size_t)nc + 1) * 8)) return rc;
	if (int rc = up(B_READ_OFF, read_offsets, ((size_t)n + 1) * 8)) return rc;
	if (int rc = up(B_CHAINS, chains->chains, (size_t)nc * sizeof(cs_chain_t))) return rc;
	if (int rc = up(B_CSEEDS, chains->cseeds, (size_t)ns * sizeof(cs_seed_t))) return rc;
	if (cseed_score) { if (int rc = up(B_SCORE, cseed_score, (size_t)ns * 4)) return rc; }
	if (int rc = up(B_BASES, bases, (size_t)n_bases)) return rc;
	if (!G.pac_up) {
		std::vector<int64_t> co(R.offset.begin(), R.offset.end()); std::vector<int32_t> cl(R.len.begin(), R.len.end());
		if (int rc = up(B_PAC, pac.data(), pac.size())) return rc;
		if (int rc = up(B_CTG_OFF, co.data(), co.size() * 8)) return rc;
		if (int rc = up(B_CTG_LEN, cl.data(), cl.size() * 4)) return rc;
		HIP_TRYA(hipStreamSynchronize(s));          // (co / cl are locals)
		G.pac_up = true;
	}
	lap("uploads");
	for (int which : {B_CHAIN_READ, B_BIG}) if (int rc = ensure(G.b[which], (size_t)nc * 4 + 64)) return rc;
	if (int rc = ensure(G.b[B_BIG_READS], (size_t)n * 4 + 64)) return rc;
	for (int which : {B_W0, B_WLEN2, B_TB0}) if (int rc = ensure(G.b[which], ((size_t)nc + 1) * 8 + 64)) return rc;
	if (int rc = ensure(G.b[B_QBUF], (size_t)n_bases * 2 + 64)) return rc;
	for (int which : {B_ORD, B_REG_CI}) if (int rc = ensure(G.b[which], (size_t)ns * 4 + 64)) return rc;
	if (int rc = ensure(G.b[B_REGS], (size_t)ns * sizeof(cs_alnreg_t) + 64)) return rc;
	for (int which : {B_LP, B_RP, B_RETRY}) if (int rc = ensure(G.b[which], (size_t)ns * sizeof(cs_ext_pair_t) + 64)) return rc;
	if (int rc = ensure(G.b[B_RES], (size_t)ns * sizeof(cs_ext_result_t) + 64)) return rc;
	if (int rc = ensure(G.b[B_CTR], 8 * sizeof(unsigned long long))) return rc;
	HIP_TRYA(hipMemsetAsync(G.b[B_CTR].p, 0, 8 * sizeof(unsigned long long), s));

	csa::Args A;
	A.chain_off = (const uint64_t *)G.b[B_CHAIN_OFF].p; A.cseed_off = (const uint64_t *)G.b[B_CSEED_OFF].p; A.read_off = (const uint64_t *)G.b[B_READ_OFF].p;
	A.chains = (const cs_chain_t *)G.b[B_CHAINS].p; A.cseeds = (const cs_seed_t *)G.b[B_CSEEDS].p; A.score = cseed_score ? (const int32_t *)G.b[B_SCORE].p : nullptr;
	A.bases = (const uint8_t *)G.b[B_BASES].p; A.n_reads = n; A.n_chains = nc; A.n_seeds = ns; A.l_pac = R.l_pac; A.n_bases = n_bases;
	A.pac = (const uint8_t *)G.b[B_PAC].p; A.ctg_off = (const int64_t *)G.b[B_CTG_OFF].p; A.ctg_len = (const int32_t *)G.b[B_CTG_LEN].p; A.n_ctg = (int32_t)R.offset.size();
	A.o = o;
	A.chain_read = (uint32_t *)G.b[B_CHAIN_READ].p; A.w0 = (int64_t *)G.b[B_W0].p; A.wlen2 = (uint64_t *)G.b[B_WLEN2].p; A.tb0 = (uint64_t *)G.b[B_TB0].p;
	A.qbuf = (uint8_t *)G.b[B_QBUF].p; A.tbuf = nullptr; A.ord = (uint32_t *)G.b[B_ORD].p; A.regs = (cs_alnreg_t *)G.b[B_REGS].p; A.reg_ci = (uint32_t *)G.b[B_REG_CI].p;
	A.lp = (cs_ext_pair_t *)G.b[B_LP].p; A.rp = (cs_ext_pair_t *)G.b[B_RP].p; A.ctr = (unsigned long long *)G.b[B_CTR].p;
	A.big = (uint32_t *)G.b[B_BIG].p; A.big_reads = (uint32_t *)G.b[B_BIG_READS].p;
	A.small_chain = (o.flags & CS_ALN_NO_LIGHT_PATHS) ? 0 : csa::SMALL_CHAIN; A.light_max = (o.flags & CS_ALN_NO_LIGHT_PATHS) ? 1 : 64;
	A.purge_cap = (o.flags & CS_ALN_PURGE_FROM_HBM) ? 0 : csa::PURGE_CAP;
	auto grid = [&](int64_t items, int per_block = 256) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((items + per_block - 1) / per_block, (int64_t)G.n_cu * 16))); };

	lap("buffers");
	hipLaunchKernelGGL(csa::chain_read_kernel, grid(n), dim3(256), 0, s, A);
	hipLaunchKernelGGL(csa::query_kernel, grid((int64_t)n_bases), dim3(256), 0, s, A);
	hipLaunchKernelGGL(csa::window_kernel, grid(nc), dim3(256), 0, s, A);
	HIP_TRYA(hipGetLastError());
	{ // every window's place in the target buffer: exclusive scan of 2 x length (one element more: the total)
		HIP_TRYA(hipMemsetAsync(A.wlen2 + nc, 0, 8, s));
		size_t tb = 0;
		HIP_TRYA(rocprim::exclusive_scan(nullptr, tb, A.wlen2, A.tb0, (uint64_t)0, (size_t)nc + 1, rocprim::plus<uint64_t>(), s));
		if (int rc = ensure(G.b[B_SCAN], tb + 16)) return rc;
		HIP_TRYA(rocprim::exclusive_scan(G.b[B_SCAN].p, tb, A.wlen2, A.tb0, (uint64_t)0, (size_t)nc + 1, rocprim::plus<uint64_t>(), s));
	}
	HIP_TRYA(hipMemcpyAsync(G.h_ctr, A.tb0 + nc, 8, hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipMemcpyAsync(G.h_ctr + 1, A.ctr + 3, 8, hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipStreamSynchronize(s));
	if (G.h_ctr[1]) return cs_fail_(CS_EINVAL, "cs_extend_chains: a chain's first seed lies outside the reference");
	lap("queries, windows, scan");
	const uint64_t t_bytes = G.h_ctr[0];
	if (int rc = ensure(G.b[B_TBUF], (size_t)t_bytes + 64)) return rc;
	A.tbuf = (uint8_t *)G.b[B_TBUF].p;
	hipLaunchKernelGGL(csa::fill_kernel, grid(nc * 64), dim3(256), 0, s, A);
	lap("fill");
	hipLaunchKernelGGL(csa::region_kernel, grid(nc), dim3(256), 0, s, A);
	hipLaunchKernelGGL(csa::region_big_kernel, dim3((unsigned)G.n_cu * 8), dim3(256), 0, s, A);
	HIP_TRYA(hipGetLastError());
	HIP_TRYA(hipMemcpyAsync(G.h_ctr, A.ctr, 2 * 8, hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipStreamSynchronize(s));
	const uint64_t n_left = G.h_ctr[0], n_right = G.h_ctr[1];
	lap("regions");

	// ---- the dynamic programming (extend.hip), each side: band w, then 2w for the pairs whose path came close to the band's edge
	auto run_side = [&](cs_ext_pair_t *pairs, uint64_t cnt, bool is_left, int pen_clip) -> int {
		cs_ext_pair_t *cur = pairs, *nxt = (cs_ext_pair_t *)G.b[B_RETRY].p;
		for (int attempt = 0; attempt < 2 && cnt; ++attempt) { // MAX_BAND_TRY (comp_seed.cpp:423)
			const int w = o.w << attempt;
			const int rc = cs_extend_batch_device(ext, (int64_t)cnt, cur, A.qbuf, n_bases * 2, A.tbuf, t_bytes, w, (cs_ext_result_t *)G.b[B_RES].p);
			if (rc != CS_OK) return rc;
			st.pairs += cnt; st.launches++;
			HIP_TRYA(hipMemsetAsync(A.ctr + 2, 0, 8, s));
			hipLaunchKernelGGL(csa::apply_kernel, grid((int64_t)cnt), dim3(256), 0, s, A, (const cs_ext_pair_t *)cur, (const cs_ext_result_t *)G.b[B_RES].p, cnt, w, attempt, is_left ? 1 : 0, pen_clip, nxt);
			HIP_TRYA(hipGetLastError());
			HIP_TRYA(hipMemcpyAsync(G.h_ctr, A.ctr + 2, 8, hipMemcpyDeviceToHost, s));
			HIP_TRYA(hipStreamSynchronize(s));
			cnt = G.h_ctr[0]; st.retries += cnt;
			std::swap(cur, nxt);
		}
		return CS_OK;
	};
	if (int rc = run_side(A.lp, n_left, true, o.pen_clip5)) return rc;
	lap("left side");
	if (n_right) { hipLaunchKernelGGL(csa::right_h0_kernel, grid((int64_t)n_right), dim3(256), 0, s, A.rp, n_right, (const cs_alnreg_t *)A.regs); HIP_TRYA(hipGetLastError()); HIP_TRYA(hipStreamSynchronize(s)); }
	if (int rc = run_side(A.rp, n_right, false, o.pen_clip3)) return rc;

	lap("right side");
	hipLaunchKernelGGL(csa::seedcov_kernel, grid(ns), dim3(256), 0, s, A);
	lap("seedcov");
	if (int rc = ensure(G.b[B_PFLAG], (size_t)ns + 64)) return rc;
	HIP_TRYA(hipMemsetAsync(G.b[B_PFLAG].p, 0, (size_t)ns, s));
	hipLaunchKernelGGL(csa::purge_kernel, grid(n * 64), dim3(256), 0, s, A);
	lap("purge, light reads");
	hipLaunchKernelGGL(csa::purge_big_kernel, dim3((unsigned)G.n_cu), dim3(csa::PB), 0, s, A, (uint8_t *)G.b[B_PFLAG].p);
	HIP_TRYA(hipGetLastError());
	lap("purge");
	regs.resize((size_t)ns);
	lap("host resize");
	HIP_TRYA(hipMemcpyAsync(regs.data(), A.regs, (size_t)ns * sizeof(cs_alnreg_t), hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipMemcpyAsync(G.h_ctr, A.ctr + 4, 8, hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipStreamSynchronize(s));
	lap("download");
	st.purged += G.h_ctr[0]; st.regions += (uint64_t)ns;
	return CS_OK;
}
