// fm_device.hpp -- FM-index backward-search primitives as gfx950 device code.
//
// Replaces (integer-exact) the reference's bwt_occ4 / bwt_2occ4 / bwt_extend / bwt_set_intv / bwt_occ /
// bwt_invPsi / bwt_sa (FM_index/bwt.c:169-186, 189-220, 262-275, 107-129, 53-59, 86-96; bwt.h:82).
//
// Layout (unchanged from <prefix>.bwt, FM_index/bwt.h:73-80): block b covers BWT rows 128b .. 128b+127
// ($ removed) and is 64 bytes = 4 x 16-byte quads:
//     quad 0: count(A), count(C)   (u64 each, occurrences before the block)
//     quad 1: count(G), count(T)
//     quad 2: bases   0..63        (4 words, base j of a word in bits (15-j)*2, i.e. first base in the top bits)
//     quad 3: bases  64..127
// One lane owns one query, so a block is four global_load_dwordx4 of one 64-byte line.  The reference counts with a
// 256-entry byte LUT (bwt.c:42-51,165); here each word is reduced with 2-bit lane compares and v_bcnt_u32_b32
// (popcount-accumulate), A being derived from the number of bases taken -- same integers, no table, no LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csd {

constexpr uint64_t NONE64 = ~0ull;

struct DevIndex {
	const uint4    *bwt;      // 64-byte blocks as 4 quads
	const uint64_t *sa;       // sampled SA, sa[0] = -1
	uint64_t primary, seq_len, n_sa, n_blocks;
	uint64_t L2[5];
	uint32_t sa_mask, sa_shift;
	// full suffix array materialised in HBM at engine creation (one entry per BWT row, 4 or 8 bytes): SAL becomes ONE
	// gather instead of a walk of up to sa_intv-1 dependent Occ reads.  Null when disabled / out of memory.
	const uint32_t *fsa32;
	const uint64_t *fsa64;
};

struct Intv { uint64_t x0, x1, x2; };

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return (uint64_t)hi << 32 | lo; }

// occurrences of C, G, T among the first `nb` (0..128) bases of a block's 8 words
__device__ __forceinline__ void count_cgt(const uint4 &qa, const uint4 &qb, uint32_t nb, uint32_t &c1, uint32_t &c2, uint32_t &c3)
{
	const uint32_t w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
	c1 = c2 = c3 = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		int rem = (int)nb - 16 * i;                    // bases wanted from this word
		rem = rem < 0 ? 0 : (rem > 16 ? 16 : rem);
		// low bit of every wanted 2-bit lane: the (16-rem) trailing lanes are cut off
		uint32_t m = (uint32_t)(0x55555555ull << ((16 - rem) << 1)) & 0x55555555u;
		uint32_t lo = w[i], hi = w[i] >> 1;
		c1 += __builtin_popcount(lo & ~hi & m);        // 01
		c2 += __builtin_popcount(hi & ~lo & m);        // 10
		c3 += __builtin_popcount(lo & hi & m);         // 11
	}
}

struct Block { uint4 h0, h1, w0, w1; };

__device__ __forceinline__ Block load_block(const DevIndex &ix, uint64_t b)
{
	const uint4 *p = ix.bwt + (b << 2);
	Block k;
	k.h0 = p[0]; k.h1 = p[1]; k.w0 = p[2]; k.w1 = p[3];
	return k;
}

// counts of A,C,G,T in rows [0, row] given the block that holds `row` (row already primary-adjusted)
__device__ __forceinline__ void occ4_in_block(const Block &k, uint64_t row, uint64_t cnt[4])
{
	uint32_t nb = (uint32_t)(row & 127) + 1, c1, c2, c3;
	count_cgt(k.w0, k.w1, nb, c1, c2, c3);
	cnt[0] = u64_of(k.h0.x, k.h0.y) + (nb - c1 - c2 - c3);
	cnt[1] = u64_of(k.h0.z, k.h0.w) + c1;
	cnt[2] = u64_of(k.h1.x, k.h1.y) + c2;
	cnt[3] = u64_of(k.h1.z, k.h1.w) + c3;
}

// bwt_occ4 (bwt.c:169-186)
__device__ __forceinline__ void occ4(const DevIndex &ix, uint64_t k, uint64_t cnt[4])
{
	bool none = (k == NONE64);
	uint64_t row = none ? 0 : k - (k >= ix.primary);
	Block b = load_block(ix, row >> 7);
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
	Block bk = load_block(ix, rk >> 7);
	Block bl = load_block(ix, rl >> 7);
	occ4_in_block(bk, rk, tk);
	occ4_in_block(bl, rl, tl);
	if (kn) tk[0] = tk[1] = tk[2] = tk[3] = 0;
	if (ln) tl[0] = tl[1] = tl[2] = tl[3] = 0;
	return (kn && ln) ? 0 : ((kn || ln) ? 1 : ((rk >> 7) != (rl >> 7) ? 2 : 1));
}

// bwt_set_intv (bwt.h:82)
__device__ __forceinline__ Intv set_intv(const DevIndex &ix, int c)
{
	Intv v;
	uint64_t lc  = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	uint64_t lc1 = c == 0 ? ix.L2[1] : c == 1 ? ix.L2[2] : c == 2 ? ix.L2[3] : ix.L2[4];
	uint64_t lr  = c == 0 ? ix.L2[3] : c == 1 ? ix.L2[2] : c == 2 ? ix.L2[1] : ix.L2[0];
	v.x0 = lc + 1; v.x2 = lc1 - lc; v.x1 = lr + 1;
	return v;
}

// bwt_extend (bwt.c:262-275), all four children
__device__ __forceinline__ int extend4(const DevIndex &ix, const Intv &ik, bool is_back, Intv ok[4])
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

// bwt_extend restricted to the one child the SMEM search uses: child `c` (0..3) of ik in direction is_back
// (forward extension by read base q uses c = 3 - q, bwt.c:309-315; backward uses c = q, bwt.c:327).
__device__ __forceinline__ Intv extend1(const DevIndex &ix, const Intv &ik, bool is_back, int c)
{
	uint64_t xa = is_back ? ik.x0 : ik.x1, xb = is_back ? ik.x1 : ik.x0;
	uint64_t tk[4], tl[4];
	occ2x4(ix, xa - 1, xa - 1 + ik.x2, tk, tl);
	uint64_t s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3], s0 = tl[0] - tk[0];
	uint64_t tkc = c == 0 ? tk[0] : c == 1 ? tk[1] : c == 2 ? tk[2] : tk[3];
	uint64_t sc  = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
	uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	// children of larger bases sit in front of child c on the shifted coordinate (cascade of bwt.c:271-274)
	uint64_t above = (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
	uint64_t ya = l2c + 1 + tkc;
	uint64_t yb = xb + ((xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) ? 1 : 0) + above;
	Intv o;
	o.x0 = is_back ? ya : yb; o.x1 = is_back ? yb : ya; o.x2 = sc;
	return o;
}

// one bwt_invPsi step (bwt.c:53-59): the base at row k and its Occ come from the same 64-byte block
__device__ __forceinline__ uint64_t inv_psi(const DevIndex &ix, uint64_t k)
{
	if (k == ix.primary) return 0;
	uint64_t row = k - (k > ix.primary);
	Block b = load_block(ix, row >> 7);
	uint32_t p = (uint32_t)(row & 127);
	uint32_t wsel = p >> 4;
	uint32_t word = wsel == 0 ? b.w0.x : wsel == 1 ? b.w0.y : wsel == 2 ? b.w0.z : wsel == 3 ? b.w0.w
	              : wsel == 4 ? b.w1.x : wsel == 5 ? b.w1.y : wsel == 6 ? b.w1.z : b.w1.w;
	int c = (word >> ((~p & 15) << 1)) & 3;
	uint64_t cnt[4];
	occ4_in_block(b, row, cnt);
	uint64_t occ = c == 0 ? cnt[0] : c == 1 ? cnt[1] : c == 2 ? cnt[2] : cnt[3];
	uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	return l2c + occ;
}

// bwt_sa through the full suffix array when it is resident (same integers: SA[k] is SA[k] however it is obtained)
__device__ __forceinline__ uint64_t sa_direct(const DevIndex &ix, uint64_t k)
{
	if (k == 0) return ~0ull; // sa[0] = -1 (bwt.c:83)
	return ix.fsa32 ? (uint64_t)ix.fsa32[k] : ix.fsa64[k];
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
