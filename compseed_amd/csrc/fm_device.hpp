// fm_device.hpp -- FM-index backward-search primitives as gfx950 device code.
//
// Replaces (integer-exact) the reference's bwt_occ4 / bwt_2occ4 / bwt_extend / bwt_set_intv / bwt_occ /
// bwt_invPsi / bwt_sa (FM_index/bwt.c:169-186, 189-220, 262-275, 107-129, 53-59, 86-96; bwt.h:82).
//
// Device layout of the Occ-sampled BWT (relayout_kernel converts the file layout once at upload, in place, same size).
// The file (FM_index/bwt.h:73-80) samples Occ every 128 rows: 4 x u64 counts + 128 two-bit bases = one 64-byte line.
// On the device every 64 rows get their own self-sufficient 32-byte record, two records per line:
//     quad 0: count(A), count(C), count(G), count(T) before the record's first row   (u32 each)
//     quad 1: low-bit plane of the 64 bases (2 words, word w bit j = base 32w + j), then the high-bit plane (2 words)
// One lane owns one query.  An Occ lookup is then TWO global_load_dwordx4 of one half line (the memory pipeline
// charges a fully divergent wave-instruction per lane, and the kernels were measured to be stalled on VMEM issue, so
// halving the instructions per lookup matters more than the bytes), and counting the bases up to a row is 2 words x
// (3 logic + 3 masked v_bcnt_u32_b32) with a "first t bits" mask per word; no 256-entry LUT (bwt.c:42-51,165).  A is
// derived from the number of bases taken.  32-bit counts hold as long as no single base occurs 2^32 times in
// forward + reverse text (checked at engine creation; hg19: 1.8e9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csd {

constexpr uint64_t NONE64 = ~0ull;
constexpr int OCC_SHIFT = 6;            // rows per record = 64
constexpr uint32_t OCC_MASK = 63;

struct DevIndex {
	const uint4    *bwt;      // 32-byte records (2 quads) of 64 rows each
	const uint64_t *sa;       // sampled SA, sa[0] = -1
	uint64_t primary, seq_len, n_sa, n_blocks; // n_blocks = 128-row blocks of the file layout
	uint64_t L2[5];
	uint32_t sa_mask, sa_shift;
	// full suffix array materialised in HBM at engine creation (one entry per BWT row, 4 or 8 bytes): SAL becomes ONE
	// gather instead of a walk of up to sa_intv-1 dependent Occ reads.  Null when disabled / out of memory.
	const uint32_t *fsa32;
	const uint64_t *fsa64;
	// "text mode" for unique matches (smem_split.hpp): the text itself, 2 bits per base (16 bases per word, base j in bits
	// 2j..2j+1), and the inverse suffix array.  Null when disabled / out of memory.
	const uint32_t *text2;
	const uint32_t *isa32;
	const uint64_t *isa64;
	// Re-seeding from the text (smem_split.hpp, r2text_kernel): lcp[r] = min(255, LCP(suffix of row r-1, suffix of row r))
	// for rows 1..seq_len (lcp[0] = lcp[seq_len+1] = 0), and rep[p] = max(lcp[ISA[p]], lcp[ISA[p]+1]) = length of the longest
	// substring starting at text position p that occurs at least twice (capped at 255).  Null when disabled.
	const uint8_t *lcp;
	const uint8_t *rep;
};

// ---- byte model of the kernels (bench.py's roofline): every read of an index-side array is an EVENT, counted per lane in a
// register and summed over the wave once, at the end of the kernel, into evc[kernel][event] (no atomics in the hot loop).
// The engine turns events into bytes (cs_engine_traffic_model).
enum : int { EV_REC = 0 /* 32-B Occ record */, EV_JUMP /* 16-B jump-table entry */, EV_BLOOM /* 8-B filter word */, EV_SA /* full-SA entry */,
             EV_ISA /* inverse-SA entry */, EV_TEXT /* 4-B word of the 2-bit text */, EV_REP /* 8-B load of rep[] */, EV_LCP /* byte of lcp[] */,
             EV_LEP /* 16-B LEP entry read or written */, EV_MEM /* 32-B mem record read back */, N_EV };
enum : int { KID_FWD0 = 0, KID_FWD, KID_BWD_WIN, KID_BWD_WIN0, KID_BWD_WIDE, KID_BWD_ALL, KID_R2TEXT, KID_R3TEXT, KID_FUSED, N_KID };
// The counters cost registers the hot kernels do not have to spare (their launch bounds leave no slack: the counting build of
// fwd_kernel spills dozens of VGPRs instead of 9), so the kernels exist twice: WaveCtrT<false> compiles to nothing and is what
// every timed call runs; cs_params_t.count_traffic selects the counting instantiation for a pass whose only purpose is the model.
template <bool ON> struct WaveCtrT;
template <> struct WaveCtrT<true> {
	uint32_t v[N_EV] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                               // this LANE's events
#ifdef CS_STEP_HIST // experiment build (make variant DEFS=-DCS_STEP_HIST): slots 0..7 = iterations of bwd_win_kernel's wave loop by the number of lanes that extend in it (0, 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64), slot 8 = the extensions
	__device__ __forceinline__ void add(int, uint32_t) {}
	__device__ __forceinline__ void addn(int, uint32_t) {}
	__device__ __forceinline__ void rec(bool) {}
	__device__ __forceinline__ void hist(uint32_t) {}
	__device__ __forceinline__ void steps(uint32_t n) { if ((threadIdx.x & 63u) == 0) { v[n == 0 ? 0 : n == 1 ? 1 : n == 2 ? 2 : n <= 4 ? 3 : n <= 8 ? 4 : n <= 16 ? 5 : n <= 32 ? 6 : 7] += 1u; v[8] += n; } }
#elif defined(CS_X2_HIST) // experiment build (make variant DEFS=-DCS_X2_HIST): the ten event slots hold a histogram of the interval sizes extend1 is asked for
	__device__ __forceinline__ void add(int, uint32_t) {}
	__device__ __forceinline__ void addn(int, uint32_t) {}
	__device__ __forceinline__ void rec(bool) {}
	__device__ __forceinline__ void hist(uint32_t x2) { v[x2 <= 1 ? 0 : x2 <= 2 ? 1 : x2 <= 4 ? 2 : x2 <= 8 ? 3 : x2 <= 16 ? 4 : x2 <= 64 ? 5 : x2 <= 256 ? 6 : x2 <= 4096 ? 7 : x2 <= 65536 ? 8 : 9] += 1u; }
	__device__ __forceinline__ void steps(uint32_t) {}
#else
	__device__ __forceinline__ void add(int ev, uint32_t n) { v[ev] += n; }
	__device__ __forceinline__ void addn(int ev, uint32_t n) { v[ev] += n; }
	__device__ __forceinline__ void rec(bool two) { v[EV_REC] += two ? 2u : 1u; }
	__device__ __forceinline__ void hist(uint32_t) {}
	__device__ __forceinline__ void steps(uint32_t) {}
#endif
	__device__ __forceinline__ void flush(unsigned long long *evc, int kid) const                    // every lane of the wave must call it
	{
		for (int ev = 0; ev < N_EV; ++ev) {
			unsigned long long t = v[ev];
			for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
			if ((threadIdx.x & 63u) == 0 && evc && t) atomicAdd(evc + kid * N_EV + ev, t);
		}
	}
};
template <> struct WaveCtrT<false> {
	__device__ __forceinline__ void add(int, uint32_t) {}
	__device__ __forceinline__ void addn(int, uint32_t) {}
	__device__ __forceinline__ void rec(bool) {}
	__device__ __forceinline__ void hist(uint32_t) {}
	__device__ __forceinline__ void steps(uint32_t) {}
	__device__ __forceinline__ void flush(unsigned long long *, int) const {}
};
using WaveCtr = WaveCtrT<true>;
using NoCtr = WaveCtrT<false>;
template <class WC> __device__ __forceinline__ void wc_add(WC &W, int ev, uint32_t per_lane = 1u) { W.add(ev, per_lane); }
template <class WC> __device__ __forceinline__ void wc_flush(const WC &W, unsigned long long *evc, int kid) { W.flush(evc, kid); }

struct Intv { uint64_t x0, x1; uint32_t x2; }; // the search's bi-interval; x2 < 2^32: no base occurs 2^32 times (checked at upload)
struct Intv64 { uint64_t x0, x1, x2; };           // the primitive entry points take any interval, e.g. the whole text

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return (uint64_t)hi << 32 | lo; }

// file layout -> device layout, in place, one thread per 128-row block of the file (run once per engine)
__global__ void relayout_kernel(uint4 *bwt, uint64_t n_blocks, unsigned long long *overflow)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	uint4 q0 = bwt[b * 4], q1 = bwt[b * 4 + 1], q2 = bwt[b * 4 + 2], q3 = bwt[b * 4 + 3];
	const uint32_t w[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
	uint64_t h[4] = {u64_of(q0.x, q0.y), u64_of(q0.z, q0.w), u64_of(q1.x, q1.y), u64_of(q1.z, q1.w)};
	uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, first64[4] = {0, 0, 0, 0};
	for (int i = 0; i < 128; ++i) {
		uint32_t code = (w[i >> 4] >> ((15 - (i & 15)) << 1)) & 3u; // bwt_B0, bwt.h:80
		lo[i >> 5] |= (code & 1u) << (i & 31);
		hi[i >> 5] |= (code >> 1) << (i & 31);
		if (i < 64) ++first64[code];
	}
	for (int c = 0; c < 4; ++c) if ((h[c] + first64[c]) >> 32) atomicAdd(overflow, 1ull);
	bwt[b * 4]     = make_uint4((uint32_t)h[0], (uint32_t)h[1], (uint32_t)h[2], (uint32_t)h[3]);
	bwt[b * 4 + 1] = make_uint4(lo[0], lo[1], hi[0], hi[1]);
	bwt[b * 4 + 2] = make_uint4((uint32_t)(h[0] + first64[0]), (uint32_t)(h[1] + first64[1]), (uint32_t)(h[2] + first64[2]), (uint32_t)(h[3] + first64[3]));
	bwt[b * 4 + 3] = make_uint4(lo[2], lo[3], hi[2], hi[3]);
}

struct Block { uint4 cnt, pl; }; // counts A,C,G,T | lo0, lo1, hi0, hi1

__device__ __forceinline__ Block load_block(const DevIndex &ix, uint64_t rec)
{
	const uint4 *p = ix.bwt + (rec << 1);
	Block k;
	k.cnt = p[0]; k.pl = p[1];
	return k;
}

// occurrences of C, G, T among the first `nb` (1..64) bases of a record
__device__ __forceinline__ void count_cgt(const Block &k, uint32_t nb, uint32_t &c1, uint32_t &c2, uint32_t &c3)
{
	// masks of the first nb bases in the two 32-base words; popcounts accumulate (v_bcnt_u32_b32 adds for free):
	// T = lo & hi, C = lo without T, G = hi without T
	const uint64_t m = ~0ull >> (64u - nb); // nb in 1..64: one 64-bit shift instead of two compare/select chains
	const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
	const uint32_t lo0 = k.pl.x & m0, lo1 = k.pl.y & m1, hi0 = k.pl.z & m0, hi1 = k.pl.w & m1;
	c3 = __builtin_popcount(lo0 & hi0) + __builtin_popcount(lo1 & hi1);
	c1 = __builtin_popcount(lo0) + __builtin_popcount(lo1) - c3;
	c2 = __builtin_popcount(hi0) + __builtin_popcount(hi1) - c3;
}

// counts of A,C,G,T in rows [0, row] given the block that holds `row` (row already primary-adjusted)
__device__ __forceinline__ void occ4_in_block(const Block &k, uint64_t row, uint64_t cnt[4])
{
	uint32_t nb = ((uint32_t)row & OCC_MASK) + 1, c1, c2, c3;
	count_cgt(k, nb, c1, c2, c3);
	cnt[0] = (uint64_t)k.cnt.x + (nb - c1 - c2 - c3);
	cnt[1] = (uint64_t)k.cnt.y + c1;
	cnt[2] = (uint64_t)k.cnt.z + c2;
	cnt[3] = (uint64_t)k.cnt.w + c3;
}

// bwt_occ4 (bwt.c:169-186)
__device__ __forceinline__ void occ4(const DevIndex &ix, uint64_t k, uint64_t cnt[4])
{
	bool none = (k == NONE64);
	uint64_t row = none ? 0 : k - (k >= ix.primary);
	Block b = load_block(ix, row >> OCC_SHIFT);
	occ4_in_block(b, row, cnt);
	if (none) cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
}

// bwt_2occ4 (bwt.c:189-220).  Both rows are fetched up front so the two 64-byte lines are in flight together;
// when they share a block the second fetch is the same line (L1 hit).  Returns the number of distinct lines (0..2),
// which is what the reference touches (bwt.c:194).
__device__ __forceinline__ int occ2x4(const DevIndex &ix, uint64_t k, uint64_t l, uint64_t tk[4], uint64_t tl[4])
{
	bool kn = (k == NONE64), ln = (l == NONE64);
	uint64_t rk = kn ? 0 : k - (k >= ix.primary);
	uint64_t rl = ln ? 0 : l - (l >= ix.primary);
	Block bk = load_block(ix, rk >> OCC_SHIFT);
	Block bl = load_block(ix, rl >> OCC_SHIFT);
	occ4_in_block(bk, rk, tk);
	occ4_in_block(bl, rl, tl);
	if (kn) tk[0] = tk[1] = tk[2] = tk[3] = 0;
	if (ln) tl[0] = tl[1] = tl[2] = tl[3] = 0;
	return (kn && ln) ? 0 : ((kn || ln) ? 1 : ((rk >> 7) != (rl >> 7) ? 2 : 1)); // 128-row blocks of the reference
}

__device__ __forceinline__ uint64_t sel4(int c, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3)
{
	return c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3;
}

// bwt_set_intv (bwt.h:82)
__device__ __forceinline__ Intv set_intv(const DevIndex &ix, int c)
{
	Intv v;
	uint64_t lc  = sel4(c, ix.L2[0], ix.L2[1], ix.L2[2], ix.L2[3]);
	uint64_t lc1 = sel4(c, ix.L2[1], ix.L2[2], ix.L2[3], ix.L2[4]);
	uint64_t lr  = sel4(c, ix.L2[3], ix.L2[2], ix.L2[1], ix.L2[0]);
	v.x0 = lc + 1; v.x2 = lc1 - lc; v.x1 = lr + 1;
	return v;
}

// bwt_extend (bwt.c:262-275), all four children
__device__ __forceinline__ int extend4(const DevIndex &ix, const Intv64 &ik, bool is_back, Intv64 ok[4])
{
	uint64_t xa = is_back ? ik.x0 : ik.x1, xb = is_back ? ik.x1 : ik.x0;
	uint64_t tk[4], tl[4];
	int lines = occ2x4(ix, xa - 1, xa - 1 + ik.x2, tk, tl);
	uint64_t ya[4], yb[4], y2[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) { ya[c] = ix.L2[c] + 1 + tk[c]; y2[c] = tl[c] - tk[c]; }
	yb[3] = xb + ((xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) ? 1 : 0);
	yb[2] = yb[3] + y2[3];
	yb[1] = yb[2] + y2[2];
	yb[0] = yb[1] + y2[1];
#pragma unroll
	for (int c = 0; c < 4; ++c) { ok[c].x0 = is_back ? ya[c] : yb[c]; ok[c].x1 = is_back ? yb[c] : ya[c]; ok[c].x2 = y2[c]; }
	return lines;
}

// bwt_extend restricted to the one child the SMEM search uses: child `c` (0..3) of ik in direction IS_BACK
// (forward extension by read base q uses c = 3 - q, bwt.c:309-315; backward uses c = q, bwt.c:327).  Both rows of the
// searched coordinate are primary-adjusted and fetched before anything is counted, so the two lines travel together.
// The searched coordinate is never 0 for a real bi-interval (it starts at L2[c]+1 >= 1), so the k == -1 sentinel of
// bwt_2occ4 cannot occur here; the four-child form above keeps it for the primitive-level entry points.
template <bool IS_BACK, class WC>
__device__ __forceinline__ Intv extend1(const DevIndex &ix, const Intv &ik, int c, WC &W)
{
	uint64_t xa = IS_BACK ? ik.x0 : ik.x1, xb = IS_BACK ? ik.x1 : ik.x0;
	uint64_t k = xa - 1, l = xa - 1 + ik.x2;
	uint64_t rk = k - (k >= ix.primary), rl = l - (l >= ix.primary);
	// Both rows usually fall into ONE record once the interval is small (bwt.c:194): then one record is requested, not
	// two.  The second load sits under the lane's predicate and is issued right behind the first, so when it is needed the
	// two still travel together; nothing waits until the first use below.
	const bool two = (rk >> OCC_SHIFT) != (rl >> OCC_SHIFT);
	W.rec(two); // records requested: one per lane, two where the rows straddle
	W.hist(ik.x2);
	Block bk = load_block(ix, rk >> OCC_SHIFT);
	Block b2;
	if (two) b2 = load_block(ix, rl >> OCC_SHIFT);
	// the select below must not be folded into the branch above (the compiler would then wait for the first record before
	// issuing the second in order to copy it): hide the predicate behind an empty asm
	uint32_t two_sel = two ? 1u : 0u;
	asm volatile("" : "+v"(two_sel));
	Block bl;
	bl.cnt = two_sel ? b2.cnt : bk.cnt; bl.pl = two_sel ? b2.pl : bk.pl;
	uint32_t nk = ((uint32_t)rk & OCC_MASK) + 1, nl = ((uint32_t)rl & OCC_MASK) + 1, k1, k2, k3, l1, l2, l3;
	count_cgt(bk, nk, k1, k2, k3);
	count_cgt(bl, nl, l1, l2, l3);
	// Everything up to the final coordinates fits 32 bits: no base occurs 2^32 times (checked at upload), so Occ values
	// and child sizes are below 2^32 and their differences are exact modulo 2^32.  (This function is the VALU hot spot of
	// the backward kernels, which are issue-bound: 64-bit adds and selects cost two instructions each.)
	const uint32_t tk0 = bk.cnt.x + (nk - k1 - k2 - k3), tk1 = bk.cnt.y + k1, tk2 = bk.cnt.z + k2, tk3 = bk.cnt.w + k3;
	const uint32_t s0 = bl.cnt.x + (nl - l1 - l2 - l3) - tk0, s1 = bl.cnt.y + l1 - tk1, s2 = bl.cnt.z + l2 - tk2, s3 = bl.cnt.w + l3 - tk3;
	const uint32_t tkc = c == 0 ? tk0 : c == 1 ? tk1 : c == 2 ? tk2 : tk3;
	const uint32_t sc  = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
	uint64_t l2c = sel4(c, ix.L2[0], ix.L2[1], ix.L2[2], ix.L2[3]);
	// children of larger bases sit in front of child c on the shifted coordinate (cascade of bwt.c:271-274)
	uint64_t above = (uint64_t)(c < 3 ? s3 : 0u) + (uint64_t)(c < 2 ? s2 : 0u) + (uint64_t)(c < 1 ? s1 : 0u);
	uint64_t ya = l2c + 1 + tkc;
	uint64_t yb = xb + ((xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) ? 1 : 0) + above;
	Intv o;
	o.x0 = IS_BACK ? ya : yb; o.x1 = IS_BACK ? yb : ya; o.x2 = (uint64_t)sc;
	return o;
}

// direction chosen at run time (the fused single-kernel path): forward extension is backward extension with the two
// coordinates swapped (bwt.c:265-274 is symmetric in x[!is_back] / x[is_back])
template <class WC>
__device__ __forceinline__ Intv extend1_rt(const DevIndex &ix, const Intv &ik, bool is_back, int c, WC &W)
{
	Intv s;
	s.x0 = is_back ? ik.x0 : ik.x1; s.x1 = is_back ? ik.x1 : ik.x0; s.x2 = ik.x2;
	Intv y = extend1<true>(ix, s, c, W), o;
	o.x0 = is_back ? y.x0 : y.x1; o.x1 = is_back ? y.x1 : y.x0; o.x2 = y.x2;
	return o;
}

// one bwt_invPsi step (bwt.c:53-59): the base at row k and its Occ come from the same 64-byte block
__device__ __forceinline__ uint64_t inv_psi(const DevIndex &ix, uint64_t k)
{
	if (k == ix.primary) return 0;
	uint64_t row = k - (k > ix.primary);
	Block b = load_block(ix, row >> OCC_SHIFT);
	uint32_t p = (uint32_t)row & OCC_MASK, w = p >> 5, bit = p & 31;
	uint32_t lo = w == 0 ? b.pl.x : b.pl.y;
	uint32_t hi = w == 0 ? b.pl.z : b.pl.w;
	int c = (int)(((lo >> bit) & 1u) | (((hi >> bit) & 1u) << 1));
	uint64_t cnt[4];
	occ4_in_block(b, row, cnt);
	return sel4(c, ix.L2[0], ix.L2[1], ix.L2[2], ix.L2[3]) + sel4(c, cnt[0], cnt[1], cnt[2], cnt[3]);
}

// bwt_sa through the full suffix array when it is resident (same integers: SA[k] is SA[k] however it is obtained)
__device__ __forceinline__ uint64_t sa_direct(const DevIndex &ix, uint64_t k)
{
	if (k == 0) return ~0ull; // sa[0] = -1 (bwt.c:83)
	return ix.fsa32 ? (uint64_t)ix.fsa32[k] : ix.fsa64[k];
}

__device__ __forceinline__ uint64_t isa_direct(const DevIndex &ix, uint64_t pos) { return ix.isa32 ? (uint64_t)ix.isa32[pos] : ix.isa64[pos]; }

// one-time preparation of the text-mode arrays from the full suffix array: T[SA[r] - 1] is the BWT character of row r
template <typename T>
__global__ void text_isa_fill_kernel(const DevIndex ix, const T *fsa, uint8_t *tbytes, T *isa)
{
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= ix.seq_len; r += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t s = (uint64_t)fsa[r];
		isa[s] = (T)r;
		if (r == ix.primary) continue; // the row of the whole text: its BWT character is the sentinel
		uint64_t row = r - (r > ix.primary);
		Block b = load_block(ix, row >> OCC_SHIFT);
		uint32_t p = (uint32_t)row & OCC_MASK, w = p >> 5, bit = p & 31;
		uint32_t lo = w == 0 ? b.pl.x : b.pl.y, hi = w == 0 ? b.pl.z : b.pl.w;
		tbytes[s - 1] = (uint8_t)(((lo >> bit) & 1u) | (((hi >> bit) & 1u) << 1));
	}
}
__global__ void text_pack_kernel(const uint8_t *tbytes, uint64_t n, uint32_t *text2)
{
	uint64_t nw = (n + 15) >> 4;
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t v = 0;
		for (int j = 0; j < 16; ++j) { uint64_t p = w * 16 + j; if (p < n) v |= (uint32_t)(tbytes[p] & 3) << (2 * j); }
		text2[w] = v;
	}
}

// 32 text bases from position pos on (2 bits each, base j in bits 2j..2j+1); the text buffer is padded
__device__ __forceinline__ uint64_t text_win(const DevIndex &ix, uint64_t pos)
{
	const uint64_t *t = reinterpret_cast<const uint64_t *>(ix.text2);
	uint64_t k = pos >> 5; uint32_t sh = (uint32_t)(pos & 31) << 1;
	uint64_t w0 = t[k], w1 = t[k + 1];
	return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
}
// length of the common prefix of the suffixes at text positions a and b, capped (the sentinel matches nothing)
__device__ __forceinline__ uint32_t text_lcp(const DevIndex &ix, uint64_t a, uint64_t b, uint32_t cap)
{
	uint64_t room = ix.seq_len - (a > b ? a : b);
	uint32_t lim = room < cap ? (uint32_t)room : cap, l = 0;
	while (l < lim) {
		uint64_t x = text_win(ix, a + l) ^ text_win(ix, b + l);
		uint32_t m = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 1 : 32u;
		l += m;
		if (m < 32) break;
	}
	return l < lim ? l : lim;
}
template <typename T>
__global__ void lcp_fill_kernel(const DevIndex ix, const T *fsa, uint8_t *lcp)
{
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= ix.seq_len + 1; r += (uint64_t)gridDim.x * blockDim.x)
		lcp[r] = (r == 0 || r > ix.seq_len) ? 0 : (uint8_t)text_lcp(ix, (uint64_t)fsa[r - 1], (uint64_t)fsa[r], 255u);
}
template <typename T>
__global__ void rep_fill_kernel(const DevIndex ix, const T *fsa, const uint8_t *lcp, uint8_t *rep)
{
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= ix.seq_len; r += (uint64_t)gridDim.x * blockDim.x) {
		uint8_t a = lcp[r], b = lcp[r + 1];
		rep[(uint64_t)fsa[r]] = a > b ? a : b; // row 0 (the empty suffix) writes rep[seq_len] = 0
	}
}

// bwt_sa (bwt.c:86-96)
__device__ __forceinline__ uint64_t sa_lookup(const DevIndex &ix, uint64_t k, uint32_t *steps = nullptr)
{
	uint64_t s = 0;
	while (k & ix.sa_mask) { ++s; k = inv_psi(ix, k); }
	if (steps) *steps = (uint32_t)s;
	return s + ix.sa[k >> ix.sa_shift];
}

} // namespace csd
