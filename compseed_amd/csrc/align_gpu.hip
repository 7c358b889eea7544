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
	unsigned long long *ctr;               // [0] left pairs [1] right pairs [2] retries [3] bad chains [4] purged
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
__global__ void region_kernel(const Args A)
{
	const cs_aln_params_t &o = A.o;
	for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < A.n_chains; ci += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t s0 = A.cseed_off[ci];
		const cs_seed_t *sd = A.cseeds + s0;
		const int32_t *sc = A.score ? A.score + s0 : nullptr;
		const int ns = (int)(A.cseed_off[ci + 1] - s0);
		if (ns <= 0) continue;
		const cs_chain_t c = A.chains[ci];
		const uint32_t r = A.chain_read[ci];
		const uint64_t rb0 = A.read_off[r];
		const int l_query = (int)(A.read_off[r + 1] - rb0);
		const int64_t w0 = A.w0[ci], L = (int64_t)(A.wlen2[ci] >> 1), tb0 = (int64_t)A.tb0[ci];
		uint32_t *ord = A.ord + s0;
		// seeds by score, highest first, later ones first among equals: insertion sort (chains have a handful of seeds, rarely hundreds)
		for (int i = 0; i < ns; ++i) {
			const int si = sc ? sc[i] : sd[i].len;
			int k = i;
			while (k > 0) {
				const uint32_t y = ord[k - 1];
				const int sy = sc ? sc[y] : sd[y].len;
				if (sy > si || (sy == si && (int)y > i)) break;        // y ranks before i
				ord[k] = y; --k;
			}
			ord[k] = (uint32_t)i;
		}
		for (int k = 0; k < ns; ++k) {
			const cs_seed_t s = sd[ord[k]];
			const uint64_t g = s0 + (uint64_t)k;
			cs_alnreg_t a; memset(&a, 0, sizeof a);
			a.w = o.w; a.score = a.truesc = -1; a.rid = c.rid; a.frac_rep = c.frac_rep; a.seedlen0 = s.len; a.chain = (int32_t)(ci - (int64_t)A.chain_off[r]);
			a.rb = a.re = UNSET; a.qb = a.qe = UNSET;
			if (s.qbeg) { // left: reversed read prefix against the reversed window in front of the seed
				const int64_t tl = s.rbeg - w0;
				const cs_ext_pair_t p = {(uint64_t)(A.n_bases + rb0 + (uint64_t)(l_query - s.qbeg)), (uint64_t)(tb0 + L + (L - tl)), s.qbeg, (int32_t)tl, s.len * o.a, (int32_t)g};
				A.lp[atomicAdd(A.ctr + 0, 1ull)] = p;
				a.qb = s.qbeg; a.rb = s.rbeg;
			} else { a.score = a.truesc = s.len * o.a; a.qb = 0; a.rb = s.rbeg; }
			if (s.qbeg + s.len != l_query) { // right: the rest of the read against the window behind the seed
				const int qe = s.qbeg + s.len; const int64_t re = s.rbeg + s.len - w0;
				const cs_ext_pair_t p = {(uint64_t)(rb0 + (uint64_t)qe), (uint64_t)(tb0 + re), l_query - qe, (int32_t)(L - re), 0, (int32_t)g};
				A.rp[atomicAdd(A.ctr + 1, 1ull)] = p;
				a.qe = qe; a.re = w0 + re;
			} else { a.qe = l_query; a.re = s.rbeg + s.len; }
			A.regs[g] = a; A.reg_ci[g] = (uint32_t)ci;
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
// One WAVE per read: the walk over the read's regions is sequential (a region's fate decides about the later ones), but both tests are
// "is there one among the earlier ...", so the lanes take the earlier regions / the higher-ranked seeds 64 at a time and a ballot answers.
// (One thread per read took 1.5 s for 400,000 reads with 68 regions each; this takes a few ms.)  Which regions are purged so far is kept
// in a byte array that is read around the L1 (volatile): lane 0 writes, the whole wave reads in the next step.
__global__ void purge_kernel(const Args A, uint8_t *pflag)
{
	const cs_aln_params_t &o = A.o;
	volatile uint8_t *pf = pflag;
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	unsigned long long my = 0;
	for (int64_t r = wave; r < A.n_reads; r += n_waves) {
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
				for (uint64_t base = g0; base < g && !around; base += 64) { // the earlier regions of the read that survive
					const uint64_t i = base + (uint64_t)lane;
					bool hit = false;
					if (i < g && !pf[i]) {
						const cs_alnreg_t p = A.regs[i];
						if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) && !(s.len - p.seedlen0 > .1 * l_query)) {
							int qd = s.qbeg - p.qb; int64_t rd = s.rbeg - p.rb;
							int gap = affordable_gap(o, (int)(qd < rd ? qd : rd)), w = gap < p.w ? gap : p.w;
							if (qd - rd < w && rd - qd < w) hit = true;
							else {
								qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
								gap = affordable_gap(o, (int)(qd < rd ? qd : rd)); w = gap < p.w ? gap : p.w;
								if (qd - rd < w && rd - qd < w) hit = true;
							}
						}
					}
					around = __ballot(hit) != 0;
				}
				if (!around) continue;
				bool rival = false;
				for (int base = 0; base < k && !rival; base += 64) { // seeds ranked above this one that are still in play
					const int v = base + lane;
					bool hit = false;
					if (v < k && !pf[s0 + (uint64_t)v]) {
						const cs_seed_t t = sd[ord[v]];
						if (!(t.len < s.len * .95)) {
							if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) hit = true;
							else if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) hit = true;
						}
					}
					rival = __ballot(hit) != 0;
				}
				if (rival) continue;
				if (lane == 0) { pf[g] = 1; A.regs[g].qb = -1; A.regs[g].qe = -1; ++my; }
				__threadfence();
			}
		}
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
       B_REG_CI, B_LP, B_RP, B_RETRY, B_RES, B_CTR, B_SCAN, B_PFLAG, B_COUNT };
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
	reg_off.assign((size_t)n + 1, 0); regs.clear();
	for (int64_t r = 0; r < n; ++r) reg_off[(size_t)r + 1] = chains->cseed_off[chains->chain_off[r + 1]];
	st.reads += (uint64_t)n;
	if (ns == 0) return CS_OK;
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
	for (int which : {B_CHAIN_READ}) if (int rc = ensure(G.b[which], (size_t)nc * 4 + 64)) return rc;
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
	hipLaunchKernelGGL(csa::region_kernel, grid(nc), dim3(256), 0, s, A);
	HIP_TRYA(hipGetLastError());
	HIP_TRYA(hipMemcpyAsync(G.h_ctr, A.ctr, 2 * 8, hipMemcpyDeviceToHost, s));
	HIP_TRYA(hipStreamSynchronize(s));
	const uint64_t n_left = G.h_ctr[0], n_right = G.h_ctr[1];
	lap("fill, regions");

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
	hipLaunchKernelGGL(csa::purge_kernel, grid(n * 64), dim3(256), 0, s, A, (uint8_t *)G.b[B_PFLAG].p);
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
