// extend.hip -- banded Smith-Waterman seed extension on gfx950 behind cs_extend_batch (include/compseed_amd.h).
//
// What it replaces: the extensions mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319) hands, batch by batch, to
// BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (mapping/bandedSWA.cpp:412, 1117, 242; call sites
// comp_seed.cpp:1719, 1790, 1859, 1942, 2003, 2074): for a query (the read's bases beyond one end of a seed), a target (the reference
// window beyond it) and the score h0 the seed has reached, the best score of an alignment that starts at the seed and extends into
// both, and where it ends.  The definition is ksw_extend2 (bwalib/ksw.c:380-479): an affine-gap DP over a band of +-w diagonals that
// also shrinks to the columns still alive, stops at a Z-drop, and reports six numbers.  Results are bit-identical to that definition
// (tests/test_gpu_extend.py: every extension the reference performed on the golden read sets + its scalar code's known answers).
//
// Mapping: ONE WAVEFRONT PER PAIR, one query column per lane, rows in sequence.  A row of the reference's inner loop looks serial --
// F(i, j+1) depends on F(i, j) -- but insertions open from the diagonal score M only (ksw.c:436,446: "100M3I3D20M" is disallowed), and
// M(i, j) = H(i-1, j-1) + S depends on the previous row alone.  So with g(j) = max(M(j) - o_ins - e_ins, 0)
//     F(j) = max over beg <= k < j of g(k) - (j - 1 - k) e_ins      =  [prefix maximum of g(k) + k e_ins]  -  (j - 1) e_ins,
// one exclusive max-scan across the wave (6 DPP steps), and everything else in the row is element-wise: H = max(M, E, F), the new E, the
// shift of H by one column for the next row's diagonal (one DPP move), the row maximum and its LAST column (a reduction + a ballot),
// and the first / last live column for the adaptive band (two ballots).  The band state lives where the reference keeps it -- two
// arrays over the query columns, here in LDS -- including what dropped columns last held, because a band that grows again reads it.
// Queries longer than 64 columns take several chunks per row with the scan's carry handed on; queries too long for LDS use the same
// code over a scratch area in HBM.  The target row base is a lane-indexed register window, re-read every 64 rows.
//
// This kernel is integer-compute-bound (VALU + cross-lane), not HBM-bound: a pair reads its two sequences once and writes 24 bytes.
#include "cs_internal.hpp"

#include <algorithm>
#include <cstring>
#include <string>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#define HIP_TRYX(expr)                                                                              \
	do {                                                                                            \
		hipError_t e__ = (expr);                                                                    \
		if (e__ != hipSuccess) {                                                                    \
			(void)hipGetLastError();                                                                \
			return cs_fail_(e__ == hipErrorOutOfMemory ? CS_ENOMEM : CS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
		}                                                                                           \
	} while (0)

namespace cse {

constexpr int NEG = -0x40000000; // identity of max for scores (|score| < 2^30: checked on the host)
constexpr int DECLINED = (int)0x80000000; // result.score written by extend16_kernel for a pair it leaves to extend_kernel

// ---- cross-lane primitives (gfx9 DPP: row shifts inside the four rows of 16 lanes, then row broadcasts to combine the rows)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_take(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wave_scan_max(int v) // inclusive: lane l gets max(v[0..l])
{
	// Six v_max_i32_dpp in place: a lane without a source (row start, rows a broadcast does not address) is simply not written and keeps its
	// own value.  Written as assembly because the compiler emits identity move + DPP move + max per step (this scan runs twice per 64 DP
	// cells and the kernel is VALU-bound); the s_nop are the two wait states a DPP read needs behind a VALU write of the same register.
	asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	             "s_nop 1"
	             : "+v"(v));
	return v;
}
__device__ __forceinline__ int wave_shift_up(int v, int lane0) { return dpp_take<0x138, 0xf>(lane0, v); } // lane l gets v[l-1], lane 0 gets lane0 (wave_shr:1)

struct ExtParams { int32_t match, mismatch, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, best; }; // match / mismatch = mat[0] / mat[1]; best = largest matrix entry

struct ExtArgs {
	const cs_ext_pair_t *pairs; int64_t n;
	const uint8_t *qbuf, *tbuf; uint64_t q_bytes, t_bytes;
	cs_ext_result_t *out;
	int32_t w, max_qlen;
	int32_t min_qlen16;                // extend16_kernel leaves queries of up to this many bases to extend_kernel
	int32_t packed16;                  // extend16_kernel takes the pairs of the 16-bit class; extend_kernel skips what that one has done
	ExtParams P;
	const int8_t *mat;                 // the 5 x 5 matrix in device memory (read by the scalar rule only)
	int32_t *scratch;                  // HBM variant: (max_qlen + 2) x 2 ints per wave
	unsigned long long *err;           // pairs whose offsets leave the buffers / qlen < 1 (skipped, result zeroed)
	unsigned long long *stat;          // [0] DP cells computed, [1] rows
};

// the two scoring rules of the reference: its vectorised code compares codes (mapping/bandedSWA.cpp:286-290),
// its scalar code -- used for pairs of 32768 bases or more, comp_seed.cpp:1569-1577 -- indexes the matrix (ksw.c:392-395)
__device__ __forceinline__ int pair_score(const ExtParams &P, const int8_t *mat, bool vec_rule, int t, int q)
{
	if (vec_rule) return (t == 4 || q == 4) ? -1 : (t == q ? P.match : P.mismatch);
	const int k = 5 * t + q;
	return k < 25 ? (int)mat[k] : 0; // (a code-5 query base against target code 4 reads past the reference's matrix: undefined there, 0 here)
}

template <bool LDS>
__global__ __launch_bounds__(256, 8) void extend_kernel(const ExtArgs A)
{
	extern __shared__ int32_t smem[];
	// (readfirstlane: the wave number is the same in all 64 lanes, and telling the compiler so keeps everything derived from it -- the pair,
	// the row loop, the band -- in scalar registers)
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
	const int64_t wave = (int64_t)blockIdx.x * wpb + wv, n_waves = (int64_t)gridDim.x * wpb;
	const int cols = A.max_qlen + 2;
	int32_t *Hd, *Ev; uint8_t *qs = nullptr;
	if (LDS) { // per wave: Hd[cols], Ev[cols], then the query bytes
		const int per_wave = 2 * cols + ((A.max_qlen + 3) >> 2);
		Hd = smem + (size_t)wv * per_wave; Ev = Hd + cols; qs = reinterpret_cast<uint8_t *>(Ev + cols);
	} else { Hd = A.scratch + (size_t)wave * 2 * cols; Ev = Hd + cols; }
	const ExtParams &P = A.P;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_ins = P.e_ins, e_del = P.e_del;
	unsigned long long my_cells = 0, my_rows = 0;
	for (int64_t p = wave; p < A.n; p += n_waves) { // wave-uniform
		const cs_ext_pair_t pr = A.pairs[p];
		const int qlen = pr.qlen, tlen = pr.tlen, h0 = pr.h0;
		if (qlen < 1 || tlen < 0 || qlen > A.max_qlen || pr.q_off > A.q_bytes || (uint64_t)qlen > A.q_bytes - pr.q_off || pr.t_off > A.t_bytes ||
		    (uint64_t)tlen > A.t_bytes - pr.t_off) {
			if (lane == 0) { atomicAdd(A.err, 1ull); cs_ext_result_t z = {0, 0, 0, 0, 0, 0}; A.out[p] = z; }
			continue;
		}
		const uint8_t *qg = A.qbuf + pr.q_off, *tg = A.tbuf + pr.t_off;
		const bool vec_rule = qlen < 32768 && tlen < 32768 && h0 + (qlen < tlen ? qlen : tlen) * P.match < 32768;
		if (A.packed16 && A.out[p].score != DECLINED) continue;   // extend16_kernel ran first and took this pair
		// ---- row "-1" (ksw.c:398-400): the seed's score decays along the query by one insertion; both arrays cleared
		const int v1 = h0 > oe_ins ? h0 - oe_ins : 0;
		for (int j = lane; j <= qlen + 1; j += 64) {
			Hd[j] = j == 0 ? h0 : j <= qlen ? imax(v1 - (j - 1) * e_ins, 0) : 0;
			Ev[j] = 0;
		}
		if (LDS) for (int j = lane; j < qlen; j += 64) qs[j] = qg[j];
		// ---- the band cannot usefully be wider than the longest gap the best possible score pays for (ksw.c:402-410)
		int w = A.w;
		{
			int lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_ins) / (double)e_ins + 1.);
			lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
			lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_del) / (double)e_del + 1.);
			lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
		}
		__builtin_amdgcn_wave_barrier();
		int top = h0, top_i = -1, top_j = -1, g_best = -1, g_row = -1, off = 0, beg = 0, end = qlen;
		int tv = 4; // target bases i0 .. i0 + 63, one per lane
		for (int i = 0; i < tlen; ++i) { // wave-uniform
			if ((i & 63) == 0) tv = i + lane < tlen ? (int)tg[i + lane] : 4;
			const int ti = __builtin_amdgcn_readlane(tv, i & 63);
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			const int left = beg == 0 ? imax(h0 - (P.o_del + e_del * (i + 1)), 0) : 0; // H(i, beg - 1) (ksw.c:419-423)
			int row_max = 0, row_arg = -1, hcarry = left, fcarry = 0, hlast = left, first_nz = 0x7fffffff, last_nz = -1;
			for (int cb = beg; cb < end; cb += 64) { // wave-uniform: 64 columns at a time
				const int j = cb + lane; const bool act = j < end;
				const int jc = act ? j : end - 1;                                  // (every lane loads: no branch around three LDS reads)
				const int hd = Hd[jc], ev = act ? Ev[jc] : 0;
				const int qj = LDS ? (int)qs[jc] : (int)qg[jc];
				int sc = (qj == 4 || ti == 4) ? -1 : (qj == ti ? P.match : P.mismatch);
				if (!vec_rule) { const int k = 5 * ti + qj; sc = k < 25 ? (int)A.mat[k] : 0; } // (wave-uniform branch; pairs of 32768 bases or more only)
				const int M = (act && hd != 0) ? hd + sc : 0;                       // a path may not restart from a zero cell (ksw.c:436)
				const int g = act ? imax(M - oe_ins, 0) : NEG;
				// F: exclusive max-scan of g(k) + k e_ins, and the carry from the chunks to the left decaying by e_ins per column
				const int incl = wave_scan_max(act ? g + lane * e_ins : NEG);
				const int excl = wave_shift_up(incl, NEG);
				int F = imax(fcarry - lane * e_ins, excl - (lane - 1) * e_ins);
				if (lane == 0) F = fcarry;
				const int h = imax(imax(M, ev), F);
				const int hprev = wave_shift_up(h, hcarry);                       // H(i, j-1): the diagonal of column j in the next row
				const int evn = imax(ev - e_del, imax(M - oe_del, 0));
				if (act) { Hd[j] = hprev; Ev[j] = evn; }
				// row maximum and the LAST column that reaches it (ksw.c:440-441)
				const int hm = act ? h : -1;
				const int cm = __builtin_amdgcn_readlane(wave_scan_max(hm), 63);
				if (cm >= row_max) { row_max = cm; row_arg = cb + 63 - __builtin_clzll(__ballot(hm == cm)); }
				// live columns for the adaptive band (ksw.c:470-473 look at the arrays as this row leaves them)
				const unsigned long long nzm = __ballot(act && (hprev | evn) != 0);
				if (nzm) { if (first_nz == 0x7fffffff) first_nz = cb + __builtin_ctzll(nzm); last_nz = cb + 63 - __builtin_clzll(nzm); }
				const int nact = end - cb < 64 ? end - cb : 64;
				hlast = __builtin_amdgcn_readlane(h, nact - 1);
				hcarry = hlast;
				fcarry = __builtin_amdgcn_readlane(imax(F - e_ins, g), 63);        // F at the first column of the next chunk
				my_cells += (unsigned)nact;
			}
			if (lane == 0) { Hd[end] = hlast; Ev[end] = 0; }
			__builtin_amdgcn_wave_barrier();
			++my_rows;
			if ((beg < end ? end : beg) == qlen) { // the row reached the end of the query (ksw.c:452-455)
				if (!(g_best > hlast)) g_row = i;
				g_best = imax(g_best, hlast);
			}
			if (row_max == 0) break;
			if (row_max > top) {
				top = row_max; top_i = i; top_j = row_arg;
				const int d = row_arg - i; off = imax(off, d < 0 ? -d : d);
			} else if (P.zdrop > 0) { // Z-drop with the diagonal shift priced as a gap extension (ksw.c:461-467)
				const int di = i - top_i, dj = row_arg - top_j;
				if (di > dj) { if (top - row_max - (di - dj) * e_del > P.zdrop) break; }
				else if (top - row_max - (dj - di) * e_ins > P.zdrop) break;
			}
			const int nbeg = first_nz < end ? first_nz : end;
			const int jz = hlast != 0 ? end : (last_nz >= nbeg ? last_nz : nbeg - 1);
			beg = nbeg; end = jz + 2 < qlen ? jz + 2 : qlen;
		}
		if (lane == 0) { cs_ext_result_t r = {top, top_j + 1, top_i + 1, g_row + 1, g_best, off}; A.out[p] = r; }
		__builtin_amdgcn_wave_barrier();
	}
	if (lane == 0 && A.stat) { atomicAdd(A.stat, my_cells); atomicAdd(A.stat + 1, my_rows); }
}

// ---- the same algorithm with TWO query columns per lane, scores as packed int16 -- OFF by default (cs_ext_params_t.flags): measured on
// the bench workload it is exact but not faster (84 ms against 75 ms per 2 M pairs): one 128-column chunk costs 92 VALU instructions
// against 60 for a 64-column chunk of extend_kernel -- the two max-scans, the ballots and the scalar bookkeeping do not shrink, and
// unpacking for them eats what the packed element-wise part saves -- and the adaptive band of a 150-bp read's extension rarely needs more
// than one and a half 64-column chunks per row.  Kept as a second, independently written implementation of the same definition (all
// fixtures run through both) and as the starting point for longer reads, where rows are several chunks wide.
// The same algorithm with TWO query columns per lane, scores as packed int16 (v_pk_add_i16 / v_pk_max_i16 / v_pk_mad_i16).
// Exact for the pairs whose scores fit: the reference's own 16-bit class, h0 + min(qlen, tlen) x match < 32768 (comp_seed.cpp:1572), with
// codes below 8 on both sides; everything else is left to extend_kernel (result.score = DECLINED).  A 150-bp read's extension is one chunk
// of 128 columns per row.  What is packed: M, E, H, the new E, the shift of H by one COLUMN (the previous lane's high half and this lane's
// low half: one DPP move + v_alignbit), and the scoring -- the query is kept as one 16-bit word per column, bit 2t = "matches target
// code t", bit 2t+1 = "ambiguous against t", so a row's scores for both columns are one packed shift, two masks and two v_pk_mad_i16.
// What is not: the max-scan for F runs on one 32-bit value per lane (the larger of its two columns' g + column x e_ins), the row maximum
// on keys (h << 16 | column), so that the last column of the maximum falls out of the same reduction.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s(uint32_t x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u16x2 as_us(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ uint32_t as_u(s16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ s16x2 splat(int v) { s16x2 r = {(short)v, (short)v}; return r; }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32

__global__ __launch_bounds__(256, 8) void extend16_kernel(const ExtArgs A)
{
	extern __shared__ int32_t smem[];
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
	const int64_t wave = (int64_t)blockIdx.x * wpb + wv, n_waves = (int64_t)gridDim.x * wpb;
	const int W = (A.max_qlen + 4) >> 1;                               // 32-bit words per array: columns 0 .. max_qlen + 1 and a spare
	uint32_t *Hw = reinterpret_cast<uint32_t *>(smem) + (size_t)wv * 3 * W, *Ew = Hw + W, *Qw = Ew + W;
	const ExtParams &P = A.P;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_ins = P.e_ins, e_del = P.e_del;
	const s16x2 k_oe_ins = splat(oe_ins), k_oe_del = splat(oe_del), k_e_del = splat(e_del), k_zero = splat(0);
	const s16x2 k_mm = splat(P.mismatch), k_m_minus_mm = splat(P.match - P.mismatch), k_amb_minus_mm = splat(-1 - P.mismatch);
	const int c0e = 2 * lane * e_ins;                                  // column (relative to the chunk) x e_ins of this lane's low column
	unsigned long long my_cells = 0, my_rows = 0;
	for (int64_t p = wave; p < A.n; p += n_waves) { // wave-uniform
		const cs_ext_pair_t pr = A.pairs[p];
		const int qlen = pr.qlen, tlen = pr.tlen, h0 = pr.h0;
		if (qlen < 1 || tlen < 0 || qlen > A.max_qlen || pr.q_off > A.q_bytes || (uint64_t)qlen > A.q_bytes - pr.q_off || pr.t_off > A.t_bytes ||
		    (uint64_t)tlen > A.t_bytes - pr.t_off || h0 < 0 || !(qlen < 32768 && tlen < 32768 && h0 + (qlen < tlen ? qlen : tlen) * P.match < 32768) ||
		    qlen <= A.min_qlen16) { // (a query that fits one 64-column chunk is no slower one column per lane: measured)
			if (lane == 0) { cs_ext_result_t z = {DECLINED, 0, 0, 0, 0, 0}; A.out[p] = z; }   // (bad pairs are reported by extend_kernel)
			continue;
		}
		const uint8_t *qg = A.qbuf + pr.q_off, *tg = A.tbuf + pr.t_off;
		// ---- row "-1" and the query's match / ambiguity bits; a code of 8 or more anywhere: not this kernel's pair
		const int v1 = h0 > oe_ins ? h0 - oe_ins : 0;
		bool odd_code = false;
		for (int wi = lane; wi < W; wi += 64) {
			uint32_t hw = 0, qw = 0;
#pragma unroll
			for (int hlf = 0; hlf < 2; ++hlf) {
				const int j = 2 * wi + hlf;
				const int hv = j == 0 ? h0 : j <= qlen ? imax(v1 - (j - 1) * e_ins, 0) : 0;
				uint32_t code = 0;
				if (j < qlen) {
					const uint32_t q = qg[j];
					odd_code |= q > 7u;
					code = q == 4u ? 0xaaaau : ((q < 8u && q != 4u) ? (1u << (2u * q)) : 0u) | 0x0200u; // ambiguous against everything | matches t == q; target code 4 is ambiguous for every q
				}
				hw |= (uint32_t)hv << (16 * hlf); qw |= code << (16 * hlf);
			}
			Hw[wi] = hw; Ew[wi] = 0; Qw[wi] = qw;
		}
		for (int i = lane; i < tlen; i += 64) odd_code |= tg[i] > 7u;
		if (__ballot(odd_code)) {
			if (lane == 0) { cs_ext_result_t z = {DECLINED, 0, 0, 0, 0, 0}; A.out[p] = z; }
			__builtin_amdgcn_wave_barrier();
			continue;
		}
		int w = A.w;
		{
			int lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_ins) / (double)e_ins + 1.);
			lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
			lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_del) / (double)e_del + 1.);
			lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
		}
		__builtin_amdgcn_wave_barrier();
		int top = h0, top_i = -1, top_j = -1, g_best = -1, g_row = -1, off = 0, beg = 0, end = qlen;
		int tv = 4;
		for (int i = 0; i < tlen; ++i) { // wave-uniform
			if ((i & 63) == 0) tv = i + lane < tlen ? (int)tg[i + lane] : 4;
			const int ti = __builtin_amdgcn_readlane(tv, i & 63);
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			const int left = beg == 0 ? imax(h0 - (P.o_del + e_del * (i + 1)), 0) : 0;
			int row_max = 0, row_arg = -1, hcarry = left, hlast = left, first_nz = 0x7fffffff, last_nz = -1;
			int fcarry = (beg & 1) ? e_ins : 0;                       // F(beg) = 0: with an odd beg the chunk starts one (idle) column earlier
			const u16x2 shift = {(unsigned short)(2 * ti), (unsigned short)(2 * ti)};
			for (int cb = beg & ~1; cb < end; cb += 128) { // wave-uniform: 128 columns at a time
				const int j0 = cb + 2 * lane, j1 = j0 + 1;
				const bool a0 = j0 >= beg && j0 < end, a1 = j1 < end;
				const uint32_t amask = (a0 ? 0xffffu : 0u) | (a1 ? 0xffff0000u : 0u);
				int wi = (cb >> 1) + lane; wi = wi < W ? wi : W - 1;
				const uint32_t hdw = Hw[wi], evw_old = Ew[wi], qcw = Qw[wi];
				// scores of the two columns against this row's base
				const u16x2 x = as_us(qcw) >> shift;
				const s16x2 mb = as_s(as_u(x) & 0x00010001u), ab = as_s((as_u(x) >> 1) & 0x00010001u);
				const s16x2 sc = mb * k_m_minus_mm + (ab * k_amb_minus_mm + k_mm);
				// M = Hd ? Hd + S : 0, per half
				const u16x2 one = {1, 1}, ffff = {0xffff, 0xffff};
				const uint32_t nzmask = as_u(__builtin_elementwise_min(as_us(hdw), one) * ffff) & amask;
				const s16x2 Mv = as_s(as_u(as_s(hdw) + sc) & nzmask);
				const s16x2 gv = pk_max(Mv - k_oe_ins, k_zero);
				const s16x2 ev = as_s(evw_old & amask);
				// F: one 32-bit exclusive max-scan per lane over max(g0 + c0 e, g1 + c1 e)
				const int g0 = a0 ? (int)gv.x : NEG, g1 = a1 ? (int)gv.y : NEG;
				const int incl = wave_scan_max(imax(g0 + c0e, g1 + c0e + e_ins));
				const int excl = wave_shift_up(incl, NEG);
				int F0 = imax(fcarry - c0e, excl - (c0e - e_ins));
				if (lane == 0) F0 = fcarry;
				const int F1 = imax(F0 - e_ins, g0);
				const s16x2 Fv = {(short)F0, (short)F1};
				uint32_t hw = as_u(pk_max(pk_max(Mv, ev), Fv));
				if (lane == 0 && !a0) hw = (hw & 0xffff0000u) | (uint32_t)hcarry;   // (odd beg: the idle column in front of the band carries H(i, beg-1))
				const uint32_t prevw = (uint32_t)wave_shift_up((int)hw, (int)((uint32_t)hcarry << 16));
				const uint32_t hpw = __builtin_amdgcn_alignbit(hw, prevw, 16);   // low half: the previous column's H, high half: this lane's low column's
				const uint32_t evn = as_u(pk_max(ev - k_e_del, pk_max(Mv - k_oe_del, k_zero)));
				if (amask) { Hw[wi] = bfi(amask, hpw, hdw); Ew[wi] = bfi(amask, evn, evw_old); }
				// row maximum with its LAST column: keys h << 16 | column
				const int k0 = a0 ? (int)((hw << 16) | (uint32_t)j0) : -1, k1 = a1 ? (int)((hw & 0xffff0000u) | (uint32_t)j1) : -1;
				const int K = __builtin_amdgcn_readlane(wave_scan_max(imax(k0, k1)), 63);
				if ((K >> 16) >= row_max) { row_max = K >> 16; row_arg = K & 0xffff; }
				// live columns for the adaptive band
				const uint32_t nzw = (hpw | evn) & amask;
				const unsigned long long nzm = __ballot(nzw != 0);
				if (nzm) {
					const int fl = __builtin_ctzll(nzm), ll = 63 - __builtin_clzll(nzm);
					const uint32_t wf = (uint32_t)__builtin_amdgcn_readlane((int)nzw, fl), wl = (uint32_t)__builtin_amdgcn_readlane((int)nzw, ll);
					if (first_nz == 0x7fffffff) first_nz = cb + 2 * fl + ((wf & 0xffffu) ? 0 : 1);
					last_nz = cb + 2 * ll + ((wl >> 16) ? 1 : 0);
				}
				const int nact = end - cb < 128 ? end - cb : 128;                 // columns of this chunk inside [.., end)
				const uint32_t wlast = (uint32_t)__builtin_amdgcn_readlane((int)hw, (nact - 1) >> 1);
				hlast = ((nact - 1) & 1) ? (int)(wlast >> 16) : (int)(wlast & 0xffffu);
				hcarry = hlast;
				fcarry = __builtin_amdgcn_readlane(imax(F1 - e_ins, g1), 63);
				my_cells += (unsigned)(nact - (cb < beg ? 1 : 0));
			}
			if (lane == 0) { reinterpret_cast<uint16_t *>(Hw)[end] = (uint16_t)hlast; reinterpret_cast<uint16_t *>(Ew)[end] = 0; }
			__builtin_amdgcn_wave_barrier();
			++my_rows;
			if ((beg < end ? end : beg) == qlen) {
				if (!(g_best > hlast)) g_row = i;
				g_best = imax(g_best, hlast);
			}
			if (row_max == 0) break;
			if (row_max > top) {
				top = row_max; top_i = i; top_j = row_arg;
				const int d = row_arg - i; off = imax(off, d < 0 ? -d : d);
			} else if (P.zdrop > 0) {
				const int di = i - top_i, dj = row_arg - top_j;
				if (di > dj) { if (top - row_max - (di - dj) * e_del > P.zdrop) break; }
				else if (top - row_max - (dj - di) * e_ins > P.zdrop) break;
			}
			const int nbeg = first_nz < end ? first_nz : end;
			const int jz = hlast != 0 ? end : (last_nz >= nbeg ? last_nz : nbeg - 1);
			beg = nbeg; end = jz + 2 < qlen ? jz + 2 : qlen;
		}
		if (lane == 0) { cs_ext_result_t r = {top, top_j + 1, top_i + 1, g_row + 1, g_best, off}; A.out[p] = r; }
		__builtin_amdgcn_wave_barrier();
	}
	if (lane == 0 && A.stat) { atomicAdd(A.stat, my_cells); atomicAdd(A.stat + 1, my_rows); }
}

// ---- one pair per LANE (the default for short queries).  The kernels above spread ONE extension over a wave, which costs two max-scans,
// three ballots and the band bookkeeping per 64-column chunk and leaves the lanes beyond the band idle: ~2 wave instructions per DP
// cell of a 150-bp read's extension.  Here every lane runs ksw_extend2 (bwalib/ksw.c:380-479) for a pair of its own, cell by cell, the
// way the reference's vectorised code runs one pair per SIMD lane (mapping/bandedSWA.cpp): no scans, no ballots, ~28 lane instructions
// per cell.  What it needs instead: pairs of similar shape in a wave (the caller sorts them by query-length class and target length:
// a wave takes as many rows as its longest target and as many columns per row as its widest band), and the two score arrays of every
// lane in LDS -- H and E of a column packed into one dword (scores of this class are below 2^15), the lanes interleaved so that a
// wave's access to "its column" never conflicts: (qmax + 2) x 256 bytes per wave, plus the query bytes.  Pairs outside the class
// (longer queries, scores that may reach 2^15, the scalar scoring rule) are left to extend_kernel.
constexpr int LANES_NCLASS = 8;
constexpr int LANES_QCLASS[LANES_NCLASS] = {32, 48, 64, 80, 96, 112, 136, 160};          // query-length classes: one launch (and LDS size: 8.8 .. 51.7 KB per wave) each
__device__ __host__ __forceinline__ int lanes_class_of(int qlen)
{
	return qlen <= 64 ? (qlen <= 32 ? 0 : qlen <= 48 ? 1 : 2) : qlen <= 112 ? (qlen <= 80 ? 3 : qlen <= 96 ? 4 : 5) : qlen <= 136 ? 6 : qlen <= 160 ? 7 : LANES_NCLASS;
}

// sort key of a pair for the lane kernel: class | target length | query length; 0xffffffff = not for the lane kernel (its result's score
// is set to DECLINED so that extend_kernel takes it)
__global__ void lanes_keys_kernel(const ExtArgs A, uint32_t *keys, uint32_t *idx, unsigned long long *class_cnt)
{
	__shared__ unsigned int cnt[2 * LANES_NCLASS];                                        // (one atomic per block and class on the global counters: 2 M lanes on four words took 12 ms)
	if (threadIdx.x < 2 * LANES_NCLASS) cnt[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < A.n; p += (int64_t)gridDim.x * blockDim.x) {
		const cs_ext_pair_t pr = A.pairs[p];
		const bool ok = pr.qlen >= 1 && pr.tlen >= 0 && pr.tlen < 32768 && pr.q_off <= A.q_bytes && (uint64_t)pr.qlen <= A.q_bytes - pr.q_off && pr.t_off <= A.t_bytes &&
		                (uint64_t)pr.tlen <= A.t_bytes - pr.t_off && pr.h0 >= 0;
		const int cls = ok ? lanes_class_of(pr.qlen) : LANES_NCLASS;
		// every score of the extension stays below h0 + qlen x match: it has to fit 15 bits (and the reference's vectorised scoring rule applies)
		const bool fits = cls < LANES_NCLASS && (int64_t)pr.h0 + (int64_t)pr.qlen * A.P.best < 32000;
		const bool narrow = (int64_t)pr.h0 + (int64_t)pr.qlen * A.P.best <= 255;   // 8-bit scores: classes 0 .. 7; 16-bit: 8 .. 15
		const int kc = narrow ? cls : cls + LANES_NCLASS;
		if (fits) keys[p] = (uint32_t)kc << 27 | (uint32_t)pr.tlen << 12 | (uint32_t)pr.qlen;
		else { keys[p] = 0xffffffffu; A.out[p].score = DECLINED; }
		idx[p] = (uint32_t)p;
		for (int c = 0; c < 2 * LANES_NCLASS; ++c) { // (the lanes of a wave that are in this iteration, counted per class by one of them)
			const unsigned long long mk = __ballot(fits && kc == c);
			if (mk && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true))) atomicAdd(&cnt[c], (unsigned)__builtin_popcountll(mk));
		}
	}
	__syncthreads();
	if (threadIdx.x < 2 * LANES_NCLASS && cnt[threadIdx.x]) atomicAdd(class_cnt + threadIdx.x, (unsigned long long)cnt[threadIdx.x]);
}

// CellT uint32_t: H | E << 16 (scores below 2^15); uint16_t: H | E << 8 for the pairs whose scores stay below 256 -- h0 + qlen x match <= 255,
// the reference's own 8-bit class (getScores8), which is most extensions of a 150-bp read: half the LDS, twice the waves per CU
// QIN (with 8-bit scores only): the query base sits in the cell's third byte -- a dword per column that is the lane's own (no bank
// conflicts, one LDS read per cell instead of two) for a third more LDS than the 16-bit cell + query byte
template <typename CellT, bool QIN>
__global__ __launch_bounds__(64) void extend_lanes_kernel(const ExtArgs A, const uint32_t *order, int64_t first, int64_t count, int qmax)
{
	extern __shared__ uint32_t lsm[];
	constexpr int SH = QIN ? 8 : sizeof(CellT) * 4;                      // bits of a score field
	constexpr uint32_t FM = (1u << SH) - 1u;
	const int lane = threadIdx.x;
	CellT *eh = reinterpret_cast<CellT *>(lsm) + lane;                   // column j of this lane: eh[j * 64] = H | E << SH
	uint8_t *qs = reinterpret_cast<uint8_t *>(reinterpret_cast<CellT *>(lsm) + (size_t)(qmax + 2) * 64) + lane; // query base j of this lane: qs[j * 64]
	const int64_t k = (int64_t)blockIdx.x * 64 + lane;
	const bool have = k < count;
	const int64_t p = have ? (int64_t)order[first + k] : 0;
	const cs_ext_pair_t pr = have ? A.pairs[p] : cs_ext_pair_t{0, 0, 1, 0, 0, 0};
	const ExtParams &P = A.P;
	const int qlen = pr.qlen, tlen = have ? pr.tlen : 0, h0 = pr.h0;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_ins = P.e_ins, e_del = P.e_del;
	const uint8_t *qg = A.qbuf + pr.q_off, *tg = A.tbuf + pr.t_off;
	// row "-1" (ksw.c:398-400) and the query
	if (have) {
		const int v1 = h0 > oe_ins ? h0 - oe_ins : 0;
		for (int j = 0; j <= qlen + 1; ++j) eh[j * 64] = (CellT)((uint32_t)(j == 0 ? h0 : j <= qlen ? imax(v1 - (j - 1) * e_ins, 0) : 0) | (QIN && j < qlen ? (uint32_t)qg[j] << 16 : 0u));
		if (!QIN) for (int j = 0; j < qlen; ++j) qs[j * 64] = qg[j];
	}
	int w = A.w;
	{ // the band cannot usefully be wider than the longest gap the best possible score pays for (ksw.c:402-410)
		int lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_ins) / (double)e_ins + 1.);
		lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
		lim = (int)((double)(qlen * P.best + P.end_bonus - P.o_del) / (double)e_del + 1.);
		lim = lim > 1 ? lim : 1; w = w < lim ? w : lim;
	}
	int top = h0, top_i = -1, top_j = -1, g_best = -1, g_row = -1, off = 0, beg = 0, end = qlen;
	unsigned long long my_cells = 0, my_rows = 0;
	int tnext = tlen > 0 ? (int)tg[0] : 4;
	for (int i = 0; i < tlen; ++i) {
		const int ti = tnext;
		if (i + 1 < tlen) tnext = (int)tg[i + 1];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1 = beg == 0 ? imax(h0 - (P.o_del + e_del * (i + 1)), 0) : 0;   // H(i, beg - 1) (ksw.c:419-423)
		int f = 0, best = 0;                                                  // best = row maximum << 16 | its LAST column (ksw.c:440-441): one max per cell
		const int s_eq = ti == 4 ? -1 : P.match, s_ne = ti == 4 ? -1 : P.mismatch;
		uint32_t cell_n = (uint32_t)eh[beg * 64]; int q_n = QIN ? 0 : (int)qs[beg * 64];         // (column j + 1 is fetched while column j is computed: a lane has no neighbours to hide LDS latency behind)
#pragma unroll 4
		for (int j = beg; j < end; ++j) {
			const uint32_t cell = cell_n; const int qj = QIN ? (int)((cell >> 16) & 0xffu) : q_n;
			cell_n = (uint32_t)eh[(j + 1) * 64]; if (!QIN) q_n = (int)qs[(j + 1) * 64];
			int M = (int)(cell & FM), e = (int)((cell >> SH) & FM);
			const int sc = qj == 4 ? -1 : (qj == ti ? s_eq : s_ne);
			M = M != 0 ? M + sc : 0;                                        // a path may not restart from a zero cell (ksw.c:436)
			const int h = imax(imax(M, e), f);
			e = imax(e - e_del, imax(M - oe_del, 0));
			if (QIN) *reinterpret_cast<uint16_t *>(&eh[j * 64]) = (uint16_t)((uint32_t)h1 | (uint32_t)e << 8);   // (the low half: the base stays)
			else eh[j * 64] = (CellT)((uint32_t)h1 | (uint32_t)e << SH);                // H(i, j-1): the diagonal of column j in the next row; E(i+1, j)
			h1 = h;
			best = imax(best, h << 16 | j);
			f = imax(f - e_ins, imax(M - oe_ins, 0));
		}
		const int m = best >> 16, mj = best & 0xffff;
		if (QIN) *reinterpret_cast<uint16_t *>(&eh[end * 64]) = (uint16_t)h1; else eh[end * 64] = (CellT)h1;
		my_cells += (unsigned)(end > beg ? end - beg : 0); ++my_rows;
		if ((beg < end ? end : beg) == qlen) { // the row reached the end of the query (ksw.c:452-455)
			if (!(g_best > h1)) g_row = i;
			g_best = imax(g_best, h1);
		}
		if (m == 0) break;
		if (m > top) {
			top = m; top_i = i; top_j = mj;
			const int d = mj - i; off = imax(off, d < 0 ? -d : d);
		} else if (P.zdrop > 0) { // Z-drop with the diagonal shift priced as a gap extension (ksw.c:461-467)
			const int di = i - top_i, dj = mj - top_j;
			if (di > dj) { if (top - m - (di - dj) * e_del > P.zdrop) break; }
			else if (top - m - (dj - di) * e_ins > P.zdrop) break;
		}
		// the live columns (ksw.c:470-473 look at the arrays as this row leaves them): the zero cells at either edge are usually none or one,
		// so two short scans are cheaper than keeping track of them in every cell
		int nbeg = beg;
		while (nbeg < end && (QIN ? ((uint32_t)eh[nbeg * 64] & 0xffffu) : (uint32_t)eh[nbeg * 64]) == 0) ++nbeg;
		int jz = end;
		if (h1 == 0) { jz = end - 1; while (jz >= nbeg && (QIN ? ((uint32_t)eh[jz * 64] & 0xffffu) : (uint32_t)eh[jz * 64]) == 0) --jz; }
		beg = nbeg; end = jz + 2 < qlen ? jz + 2 : qlen;
	}
	if (have) { cs_ext_result_t r = {top, top_j + 1, top_i + 1, g_row + 1, g_best, off}; A.out[p] = r; }
	if (A.stat) {
		for (int o = 32; o > 0; o >>= 1) { my_cells += __shfl_xor(my_cells, o); my_rows += __shfl_xor(my_rows, o); }
		if (lane == 0) { atomicAdd(A.stat, my_cells); atomicAdd(A.stat + 1, my_rows); }
	}
}

__global__ void max_qlen_kernel(const cs_ext_pair_t *pairs, int64_t n, unsigned long long *out)
{
	unsigned long long m = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const int q = pairs[i].qlen; if (q > 0 && (unsigned long long)q > m) m = (unsigned long long)q; }
	for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(m, o); m = x > m ? x : m; }
	if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

} // namespace cse

struct cs_extender {
	int device = 0, n_cu = 256;
	cse::ExtParams P{};
	bool packed16 = true;                 // the two-columns-per-lane int16 kernel may be used (parameters fit; not switched off)
	int min_qlen16 = 64;
	bool lanes = true;                    // short queries go through extend_lanes_kernel (one pair per lane)
	bool qin = false;                     // ... with the query base inside the cell (CS_EXT_LANES_QIN: experiment)
	void *d_keys = nullptr, *d_keys2 = nullptr, *d_idx = nullptr, *d_idx2 = nullptr, *d_sort = nullptr; size_t c_keys = 0, c_keys2 = 0, c_idx = 0, c_idx2 = 0, c_sort = 0;
	unsigned long long *d_cls = nullptr, *h_cls = nullptr; // pairs per query-length class of the lane kernel
	hipStream_t stream = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
	hipStream_t side[2] = {nullptr, nullptr}; hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr}; // the lane kernel's class launches run side by side
	void *d_pairs = nullptr, *d_q = nullptr, *d_t = nullptr, *d_out = nullptr, *d_scratch = nullptr; size_t c_pairs = 0, c_q = 0, c_t = 0, c_out = 0, c_scratch = 0;
	uint64_t res_q = 0, res_t = 0;                         // bytes of the sequence buffers uploaded by cs_extender_upload
	unsigned long long *d_ctr = nullptr, *h_ctr = nullptr; // device words: [0] skipped pairs [1] cells [2] rows [3] longest query; [4..7] hold the matrix
	cs_ext_stats_t st{};
};

static int grow(void **p, size_t *cap, size_t need)
{
	if (need <= *cap) return CS_OK;
	if (*p) (void)hipFree(*p);
	*p = nullptr; *cap = 0;
	const size_t want = need + need / 4 + 256;
	HIP_TRYX(hipMalloc(p, want));
	*cap = want;
	return CS_OK;
}

extern "C" void cs_ext_params_default(cs_ext_params_t *p)
{
	if (!p) return;
	// mem_opt_init (mapping/comp_seed.cpp:26-58): a = 1, b = 4, o_del = o_ins = 6, e_del = e_ins = 1, zdrop = 100, pen_clip5 = pen_clip3 = 5; bwa_fill_scmat (bwalib/bwa.c:17-29)
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) p->mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? 1 : -4);
	p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1; p->zdrop = 100; p->end_bonus = 5; p->flags = 0;
}

extern "C" int cs_extender_create(int device, const cs_ext_params_t *par, cs_extender_t **out)
{
	if (!out) return cs_fail_(CS_EINVAL, "cs_extender_create: null argument");
	*out = nullptr;
	cs_ext_params_t dp;
	if (!par) { cs_ext_params_default(&dp); par = &dp; }
	if (par->e_del < 1 || par->e_ins < 1 || par->o_del < 0 || par->o_ins < 0 || par->zdrop < 0) return cs_fail_(CS_EINVAL, "cs_extender_create: gap extension penalties must be positive, gap opens and zdrop non-negative");
	int ndev = 0;
	HIP_TRYX(hipGetDeviceCount(&ndev));
	if (ndev <= 0) return cs_fail_(CS_EDEVICE, "no HIP device: the extension kernel has no CPU path");
	if (device < 0 || device >= ndev) return cs_fail_(CS_EINVAL, "device ordinal out of range");
	HIP_TRYX(hipSetDevice(device));
	cs_extender *x = new cs_extender();
	x->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) x->n_cu = prop.multiProcessorCount;
	x->P.match = par->mat[0]; x->P.mismatch = par->mat[1];
	x->P.o_del = par->o_del; x->P.e_del = par->e_del; x->P.o_ins = par->o_ins; x->P.e_ins = par->e_ins; x->P.zdrop = par->zdrop; x->P.end_bonus = par->end_bonus;
	x->P.best = 0;
	for (int k = 0; k < 25; ++k) x->P.best = std::max<int>(x->P.best, par->mat[k]);
	// extend16_kernel: int16 arithmetic incl. the F carry's decay over a 128-column chunk, and the vectorised scoring rule written as
	// mismatch + match-bit x (match - mismatch) + ambiguity-bit x (-1 - mismatch)
	x->min_qlen16 = (par->flags & CS_EXT_PACKED16_ALL) ? 0 : 64;
	x->packed16 = (par->flags & (CS_EXT_PACKED16 | CS_EXT_PACKED16_ALL)) && par->e_ins <= 200 && par->e_del <= 200 && par->o_ins + par->e_ins < 16000 && par->o_del + par->e_del < 16000 &&
	              par->mat[0] >= 0 && par->mat[0] <= 100 && par->mat[1] <= 0 && par->mat[1] >= -100;
	hipError_t e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreate(&x->ev0);
	if (e == hipSuccess) e = hipEventCreate(&x->ev1);
	for (int k = 0; k < 2 && e == hipSuccess; ++k) { e = hipStreamCreateWithFlags(&x->side[k], hipStreamNonBlocking); if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_join[k], hipEventDisableTiming); }
	if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipMalloc((void **)&x->d_ctr, 8 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipHostMalloc((void **)&x->h_ctr, 8 * sizeof(unsigned long long), hipHostMallocDefault);
	if (e == hipSuccess) e = hipMemcpy(x->d_ctr + 4, par->mat, 25, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMalloc((void **)&x->d_cls, 16 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipHostMalloc((void **)&x->h_cls, 16 * sizeof(unsigned long long), hipHostMallocDefault);
	x->lanes = !x->packed16 && !(par->flags & CS_EXT_NO_LANES);
	x->qin = (par->flags & CS_EXT_LANES_QIN) != 0;
	if (e != hipSuccess) { (void)hipGetLastError(); cs_extender_destroy(x); return cs_fail_(CS_EDEVICE, std::string("cs_extender_create: ") + hipGetErrorString(e)); }
	*out = x;
	return CS_OK;
}

extern "C" void cs_extender_destroy(cs_extender_t *x)
{
	if (!x) return;
	(void)hipSetDevice(x->device);
	if (x->stream) (void)hipStreamSynchronize(x->stream);
	for (void *p : {x->d_pairs, x->d_q, x->d_t, x->d_out, x->d_scratch, (void *)x->d_ctr, x->d_keys, x->d_keys2, x->d_idx, x->d_idx2, x->d_sort, (void *)x->d_cls}) if (p) (void)hipFree(p);
	if (x->h_ctr) (void)hipHostFree(x->h_ctr);
	if (x->h_cls) (void)hipHostFree(x->h_cls);
	if (x->ev0) (void)hipEventDestroy(x->ev0);
	if (x->ev1) (void)hipEventDestroy(x->ev1);
	for (int k = 0; k < 2; ++k) { if (x->ev_join[k]) (void)hipEventDestroy(x->ev_join[k]); if (x->side[k]) { (void)hipStreamSynchronize(x->side[k]); (void)hipStreamDestroy(x->side[k]); } }
	if (x->ev_fork) (void)hipEventDestroy(x->ev_fork);
	if (x->stream) (void)hipStreamDestroy(x->stream);
	delete x;
}

// the launch: pairs, sequences and results are device memory here
static int extend_device(cs_extender *x, int64_t n, const cs_ext_pair_t *d_pairs, const uint8_t *d_q, uint64_t q_bytes, const uint8_t *d_t, uint64_t t_bytes,
                         int32_t w, cs_ext_result_t *d_out)
{
	hipStream_t s = x->stream;
	HIP_TRYX(hipMemsetAsync(x->d_ctr, 0, 4 * sizeof(unsigned long long), s));
	hipLaunchKernelGGL(cse::max_qlen_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)x->n_cu * 8)), dim3(256), 0, s, d_pairs, n, x->d_ctr + 3);
	HIP_TRYX(hipGetLastError());
	HIP_TRYX(hipMemcpyAsync(x->h_ctr, x->d_ctr, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	HIP_TRYX(hipStreamSynchronize(s));
	const int64_t max_q = (int64_t)x->h_ctr[3];
	if (max_q > 65535) return cs_fail_(CS_ERANGE, "cs_extend_batch: a query is longer than 65535 bases (MAX_READ_LEN, mapping/comp_seed.h:39)");
	if ((int64_t)x->P.best * max_q > (1 << 29)) return cs_fail_(CS_ERANGE, "cs_extend_batch: scores would not fit the kernel's 30-bit range");
	cse::ExtArgs A;
	A.pairs = d_pairs; A.n = n; A.qbuf = d_q; A.tbuf = d_t; A.q_bytes = q_bytes; A.t_bytes = t_bytes; A.out = d_out; A.w = w; A.max_qlen = (int32_t)std::max<int64_t>(max_q, 1);
	A.P = x->P; A.mat = (const int8_t *)(x->d_ctr + 4); A.scratch = nullptr; A.err = x->d_ctr; A.stat = x->d_ctr + 1;
	const size_t per_wave = ((size_t)2 * (A.max_qlen + 2) + ((A.max_qlen + 3) >> 2)) * 4; // bytes of LDS per wave
	HIP_TRYX(hipEventRecord(x->ev0, s));
	A.packed16 = 0; A.min_qlen16 = x->min_qlen16;
	int64_t n_lanes = 0;
	if (x->lanes && n < (1ll << 32)) { // short queries: one pair per lane, pairs sorted by (query-length class, target length, query length)
		if (int rc = grow(&x->d_keys, &x->c_keys, (size_t)n * 4)) return rc;
		if (int rc = grow(&x->d_keys2, &x->c_keys2, (size_t)n * 4)) return rc;
		if (int rc = grow(&x->d_idx, &x->c_idx, (size_t)n * 4)) return rc;
		if (int rc = grow(&x->d_idx2, &x->c_idx2, (size_t)n * 4)) return rc;
		HIP_TRYX(hipMemsetAsync(x->d_cls, 0, 16 * sizeof(unsigned long long), s));
		hipLaunchKernelGGL(cse::lanes_keys_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)x->n_cu * 8)), dim3(256), 0, s, A, (uint32_t *)x->d_keys, (uint32_t *)x->d_idx, x->d_cls);
		HIP_TRYX(hipGetLastError());
		HIP_TRYX(hipMemcpyAsync(x->h_cls, x->d_cls, 2 * cse::LANES_NCLASS * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		size_t tb = 0;
		HIP_TRYX(rocprim::radix_sort_pairs(nullptr, tb, (uint32_t *)x->d_keys, (uint32_t *)x->d_keys2, (uint32_t *)x->d_idx, (uint32_t *)x->d_idx2, (size_t)n, 0u, 32u, s));
		if (int rc = grow(&x->d_sort, &x->c_sort, tb + 16)) return rc;
		HIP_TRYX(rocprim::radix_sort_pairs(x->d_sort, tb, (uint32_t *)x->d_keys, (uint32_t *)x->d_keys2, (uint32_t *)x->d_idx, (uint32_t *)x->d_idx2, (size_t)n, 0u, 32u, s));
		HIP_TRYX(hipStreamSynchronize(s));
		int64_t first = 0;
		// one launch per class (its LDS size), the largest first, spread over three streams: the tail of one class is filled by the next
		HIP_TRYX(hipEventRecord(x->ev_fork, s));
		for (int k = 0; k < 2; ++k) HIP_TRYX(hipStreamWaitEvent(x->side[k], x->ev_fork, 0));
		int64_t start[2 * cse::LANES_NCLASS]; int by_size[2 * cse::LANES_NCLASS];
		for (int c = 0; c < 2 * cse::LANES_NCLASS; ++c) { start[c] = first; first += (int64_t)x->h_cls[c]; by_size[c] = c; }
		std::sort(by_size, by_size + 2 * cse::LANES_NCLASS, [&](int a, int b) { return x->h_cls[a] > x->h_cls[b]; });
		int nl = 0;
		for (int o = 0; o < 2 * cse::LANES_NCLASS; ++o) {
			const int c = by_size[o];
			const int64_t cnt = (int64_t)x->h_cls[c];
			if (cnt <= 0) continue;
			const bool narrow = c < cse::LANES_NCLASS;
			const int qmax = cse::LANES_QCLASS[c % cse::LANES_NCLASS];
			const bool qin = narrow && x->qin;
			const size_t lds = qin ? (size_t)(qmax + 3) * 256 : (size_t)(qmax + 2) * 64 * (narrow ? 2 : 4) + (size_t)(qmax + 1) * 64;   // (+ 1 query row: the loop fetches one column ahead)
			hipStream_t ls = nl % 3 == 0 ? s : x->side[nl % 3 - 1]; ++nl;
			if (qin) hipLaunchKernelGGL((cse::extend_lanes_kernel<uint32_t, true>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), lds, ls, A, (const uint32_t *)x->d_idx2, start[c], cnt, qmax);
			else if (narrow) hipLaunchKernelGGL((cse::extend_lanes_kernel<uint16_t, false>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), lds, ls, A, (const uint32_t *)x->d_idx2, start[c], cnt, qmax);
			else hipLaunchKernelGGL((cse::extend_lanes_kernel<uint32_t, false>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), lds, ls, A, (const uint32_t *)x->d_idx2, start[c], cnt, qmax);
			HIP_TRYX(hipGetLastError());
		}
		for (int k = 0; k < 2; ++k) { HIP_TRYX(hipEventRecord(x->ev_join[k], x->side[k])); HIP_TRYX(hipStreamWaitEvent(s, x->ev_join[k], 0)); }
		n_lanes = first;
		A.packed16 = 1;                    // extend_kernel below takes only what the key kernel marked DECLINED
	}
	if (n_lanes == n) goto done;          // (nothing left for the wave-per-pair kernels)
	{ // the pairs of the 16-bit class first, two columns per lane (declines what does not fit: those get DECLINED as their score) ...
		const size_t per_wave16 = (size_t)3 * ((A.max_qlen + 4) >> 1) * 4;
		if (x->packed16 && per_wave16 <= 60 * 1024) {
			const int wpb = per_wave16 <= 15 * 1024 ? 4 : per_wave16 <= 30 * 1024 ? 2 : 1;
			const int64_t blocks = std::min<int64_t>((n + wpb - 1) / wpb, (int64_t)x->n_cu * (wpb == 4 ? 8 : wpb == 2 ? 4 : 2));
			hipLaunchKernelGGL(cse::extend16_kernel, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(64 * wpb), per_wave16 * wpb, s, A);
			HIP_TRYX(hipGetLastError());
			A.packed16 = 1;
		}
	}
	// ... then everything else (and every pair, where the packed kernel is off) one column per lane
	if (per_wave <= 60 * 1024) {
		const int wpb = per_wave <= 15 * 1024 ? 4 : per_wave <= 30 * 1024 ? 2 : 1;
		const int64_t blocks = std::min<int64_t>((n + wpb - 1) / wpb, (int64_t)x->n_cu * (wpb == 4 ? 8 : wpb == 2 ? 4 : 2));
		hipLaunchKernelGGL(cse::extend_kernel<true>, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(64 * wpb), per_wave * wpb, s, A);
	} else { // long queries: the band state in HBM
		const int64_t blocks = std::min<int64_t>((n + 3) / 4, (int64_t)x->n_cu * 4);
		if (int rc = grow(&x->d_scratch, &x->c_scratch, (size_t)std::max<int64_t>(blocks, 1) * 4 * 2 * (A.max_qlen + 2) * sizeof(int32_t))) return rc;
		A.scratch = (int32_t *)x->d_scratch;
		hipLaunchKernelGGL(cse::extend_kernel<false>, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), 0, s, A);
	}
	HIP_TRYX(hipGetLastError());
done:
	HIP_TRYX(hipEventRecord(x->ev1, s));
	HIP_TRYX(hipMemcpyAsync(x->h_ctr, x->d_ctr, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	HIP_TRYX(hipStreamSynchronize(s));
	float ms = 0.f;
	HIP_TRYX(hipEventElapsedTime(&ms, x->ev0, x->ev1));
	x->st.pairs += (uint64_t)n; x->st.cells += x->h_ctr[1]; x->st.rows += x->h_ctr[2]; x->st.kernel_ms += ms; x->st.launches++;
	if (x->h_ctr[0]) return cs_fail_(CS_EINVAL, "cs_extend_batch: " + std::to_string(x->h_ctr[0]) + " pair(s) with qlen < 1, tlen < 0 or offsets outside the sequence buffers (their results are zero)");
	return CS_OK;
}

extern "C" int cs_extend_batch_device(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *d_pairs, const uint8_t *d_qbuf, uint64_t q_bytes,
                                      const uint8_t *d_tbuf, uint64_t t_bytes, int32_t w, cs_ext_result_t *d_out)
{
	if (!x || n_pairs < 0 || w < 0 || (n_pairs > 0 && (!d_pairs || !d_out))) return cs_fail_(CS_EINVAL, "cs_extend_batch_device: bad argument");
	if (n_pairs == 0) return CS_OK;
	HIP_TRYX(hipSetDevice(x->device));
	return extend_device(x, n_pairs, d_pairs, d_qbuf, q_bytes, d_tbuf, t_bytes, w, d_out);
}

extern "C" int cs_extend_batch(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *pairs, const uint8_t *qbuf, uint64_t q_bytes,
                               const uint8_t *tbuf, uint64_t t_bytes, int32_t w, cs_ext_result_t *out)
{
	if (!x || n_pairs < 0 || w < 0 || (n_pairs > 0 && (!pairs || !out)) || (q_bytes && !qbuf) || (t_bytes && !tbuf)) return cs_fail_(CS_EINVAL, "cs_extend_batch: bad argument");
	if (n_pairs == 0) return CS_OK;
	HIP_TRYX(hipSetDevice(x->device));
	if (int rc = grow(&x->d_pairs, &x->c_pairs, (size_t)n_pairs * sizeof(cs_ext_pair_t))) return rc;
	if (int rc = grow(&x->d_out, &x->c_out, (size_t)n_pairs * sizeof(cs_ext_result_t))) return rc;
	if (int rc = grow(&x->d_q, &x->c_q, (size_t)q_bytes + 64)) return rc;
	if (int rc = grow(&x->d_t, &x->c_t, (size_t)t_bytes + 64)) return rc;
	hipStream_t s = x->stream;
	HIP_TRYX(hipMemcpyAsync(x->d_pairs, pairs, (size_t)n_pairs * sizeof(cs_ext_pair_t), hipMemcpyHostToDevice, s));
	if (q_bytes) HIP_TRYX(hipMemcpyAsync(x->d_q, qbuf, (size_t)q_bytes, hipMemcpyHostToDevice, s));
	if (t_bytes) HIP_TRYX(hipMemcpyAsync(x->d_t, tbuf, (size_t)t_bytes, hipMemcpyHostToDevice, s));
	const int rc = extend_device(x, n_pairs, (const cs_ext_pair_t *)x->d_pairs, (const uint8_t *)x->d_q, q_bytes, (const uint8_t *)x->d_t, t_bytes, w, (cs_ext_result_t *)x->d_out);
	const std::string keep = rc != CS_OK ? std::string(cs_last_error()) : std::string();
	if (rc == CS_OK || rc == CS_EINVAL) { // (CS_EINVAL from skipped pairs: the other results are valid and are delivered)
		hipError_t e = hipMemcpyAsync(out, x->d_out, (size_t)n_pairs * sizeof(cs_ext_result_t), hipMemcpyDeviceToHost, s);
		if (e == hipSuccess) e = hipStreamSynchronize(s);
		if (e != hipSuccess) { (void)hipGetLastError(); return cs_fail_(CS_EDEVICE, hipGetErrorString(e)); }
	}
	if (rc != CS_OK) return cs_fail_(rc, keep);
	return CS_OK;
}

// The band retries of the reference extend the SAME sequences again with another pair list (comp_seed.cpp:1717-1776): the two buffers
// are uploaded once and stay resident in the extender; every try then moves only its pairs in and its results out.
extern "C" int cs_extender_upload(cs_extender_t *x, const uint8_t *qbuf, uint64_t q_bytes, const uint8_t *tbuf, uint64_t t_bytes)
{
	if (!x || (q_bytes && !qbuf) || (t_bytes && !tbuf)) return cs_fail_(CS_EINVAL, "cs_extender_upload: bad argument");
	HIP_TRYX(hipSetDevice(x->device));
	if (int rc = grow(&x->d_q, &x->c_q, (size_t)q_bytes + 64)) return rc;
	if (int rc = grow(&x->d_t, &x->c_t, (size_t)t_bytes + 64)) return rc;
	if (q_bytes) HIP_TRYX(hipMemcpyAsync(x->d_q, qbuf, (size_t)q_bytes, hipMemcpyHostToDevice, x->stream));
	if (t_bytes) HIP_TRYX(hipMemcpyAsync(x->d_t, tbuf, (size_t)t_bytes, hipMemcpyHostToDevice, x->stream));
	HIP_TRYX(hipStreamSynchronize(x->stream));
	x->res_q = q_bytes; x->res_t = t_bytes;
	return CS_OK;
}
extern "C" int cs_extend_batch_resident(cs_extender_t *x, int64_t n_pairs, const cs_ext_pair_t *pairs, int32_t w, cs_ext_result_t *out)
{
	if (!x || n_pairs < 0 || w < 0 || (n_pairs > 0 && (!pairs || !out))) return cs_fail_(CS_EINVAL, "cs_extend_batch_resident: bad argument");
	if (n_pairs == 0) return CS_OK;
	HIP_TRYX(hipSetDevice(x->device));
	if (int rc = grow(&x->d_pairs, &x->c_pairs, (size_t)n_pairs * sizeof(cs_ext_pair_t))) return rc;
	if (int rc = grow(&x->d_out, &x->c_out, (size_t)n_pairs * sizeof(cs_ext_result_t))) return rc;
	hipStream_t s = x->stream;
	HIP_TRYX(hipMemcpyAsync(x->d_pairs, pairs, (size_t)n_pairs * sizeof(cs_ext_pair_t), hipMemcpyHostToDevice, s));
	const int rc = extend_device(x, n_pairs, (const cs_ext_pair_t *)x->d_pairs, (const uint8_t *)x->d_q, x->res_q, (const uint8_t *)x->d_t, x->res_t, w, (cs_ext_result_t *)x->d_out);
	const std::string keep = rc != CS_OK ? std::string(cs_last_error()) : std::string();
	if (rc == CS_OK || rc == CS_EINVAL) {
		hipError_t e = hipMemcpyAsync(out, x->d_out, (size_t)n_pairs * sizeof(cs_ext_result_t), hipMemcpyDeviceToHost, s);
		if (e == hipSuccess) e = hipStreamSynchronize(s);
		if (e != hipSuccess) { (void)hipGetLastError(); return cs_fail_(CS_EDEVICE, hipGetErrorString(e)); }
	}
	if (rc != CS_OK) return cs_fail_(rc, keep);
	return CS_OK;
}

extern "C" int cs_extender_stats(const cs_extender_t *x, cs_ext_stats_t *st)
{
	if (!x || !st) return cs_fail_(CS_EINVAL, "null argument");
	*st = x->st;
	return CS_OK;
}
