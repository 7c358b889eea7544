// smem_split.hpp -- the SMEM collection as a forward kernel and wavefront-cooperative backward kernels (gfx950).
//
// Same result as smem_kernel (seed_kernels.hpp) and as the reference's three rounds (mapping/bwamem.c:218-272 ==
// mapping/comp_seed.cpp:2262-2301), organised around what the hardware is good at.  Measured on MI355X the fused
// one-lane-per-read state machine is bound by instruction issue and by the one-chain-per-lane latency, not by HBM
// (a bare dependent chain of random 64-byte reads runs at 57 G lines/s, 3.6 TB/s; the fused kernel reached 35 % of
// that), and 54 % of all bwt_extend calls belong to backward sweeps that extend ~9 independent intervals per step.
//
//   fwd_kernel   one LANE per task: the forward pass of one SMEM call (bwt.c:300-320), or the whole round-3 chain of a
//                read when round 3 runs on the index only.  Starts from the k-mer jump table, leaves the index for the
//                2-bit text once the match is unique, finishes calls whose sweep is trivial or can be read off the text.
//   bwd_win0_kernel / bwd_win_kernel / bwd_wide_kernel   the backward sweeps under the window scheme (default): short match
//                ends are settled through the jump table, stored LEPs walk in groups of 32/64 lanes on one clock.
//   bwd_all_kernel   the literal sweep (bwt.c:325-345), a GROUP of G lanes (16/32/64) per call; lane g holds LEP n-1-g in
//                registers; every step extends ALL live intervals at once and the reference's sequential keep/emit rules
//                are evaluated with a ballot and one shuffle: occurrence counts are monotone along the list (a longer
//                match cannot occur more often), so the intervals that stop form a prefix, only the first of them can be
//                a new SMEM, and "differs from the last kept size" is a comparison with the previous surviving lane.
//   r2text_kernel / r3text_kernel   re-seeding calls and round-3 seeds answered from the text-side arrays.
//   (DESIGN.md section 4.2 states each shortcut and why it is exact.)
//
// Calls are chained through task queues in HBM: a finished backward sweep of round 1 enqueues the forward pass at the
// next pivot (bwamem.c:226-236), and every emitted round-1 SMEM that is long and rare enough enqueues its re-seeding
// call (bwamem.c:241-249) -- those depend only on that one SMEM, so all of them run in parallel.  The host alternates
// fwd / bwd launches until the queues are empty.  Mems of a read are appended with one atomic per mem and sorted
// afterwards (comp_seed.cpp:2301), so the order in which tasks finish is irrelevant to the output.
#pragma once
#include "fm_device.hpp"
#include "seed_kernels.hpp"

namespace csd {

enum : uint32_t { TK_ROUND1 = 0, TK_ROUND2 = 1, TK_ROUND3 = 2, TK_NOP = 3, TK_TEXT = 4 /* fwd_kernel-internal: a round-1 call in text mode */ };

// window scheme (bwd_win_run): lanes 0..WIN_LANES-1 of a group hold the short matches, the rest hold LEPs
constexpr int WIN_LANES = 18, WIN_G32_LEPS = 32 - WIN_LANES, WIN_G64_LEPS = 64 - WIN_LANES;

// forward task, 8 bytes: read | pivot | min_intv | kind
__device__ __host__ __forceinline__ uint64_t ftask_pack(uint32_t r, uint32_t x, uint32_t min_intv, uint32_t kind)
{
	return (uint64_t)r | (uint64_t)x << 32 | (uint64_t)min_intv << 48 | (uint64_t)kind << 62;
}

struct BTask { uint32_t r; uint16_t x, mi_kind, n, ret; uint32_t cls; };  // 16 bytes, stored at the forward task's slot;
                                                                           // cls = size class 0..3, 0xffffffff = no call
struct OvfRec { OutMem m; uint32_t r, pad; };                              // a mem beyond a read's first `cap`

struct SplitArgs {
	DevIndex ix;
	const uint8_t  *seq;
	const uint4    *seqp;                             // the reads as 16-byte records of 32 bases (pack_reads_kernel), already offset to this launch's first read
	const uint64_t *off;
	int64_t   n_reads;
	OutMem   *out; uint32_t *out_cnt; uint32_t cap;
	OvfRec   *ovf; unsigned long long *ovf_cnt; uint64_t ovf_cap;
	int32_t   min_seed_len, split_len;
	uint32_t  split_width;
	uint64_t  max_mem_intv;
	const uint64_t *fq; uint64_t n_f;                 // forward tasks of this launch
	uint64_t *fq_next; unsigned long long *n_f_next; uint64_t fq_cap;
	unsigned long long *n_text_sweeps;                // backward sweeps answered from the text (fwd_kernel)
	unsigned long long *n_r2_quick;                   // re-seeding calls settled by fwd0_kernel itself (counted with r2text_kernel's)
	const uint64_t *bloom; uint32_t bloom_bits;       // k-mer filter of the text for k = min_seed_len (kmer_filter_*), or null
	int32_t   win;                                    // window scheme for the backward sweeps (bwd_win_run) is on
	int32_t   text_sweep;                             // that shortcut is enabled (CS_TEXT_SWEEP, default on)
	unsigned long long *n_btasks;                     // backward calls created by this forward launch (0: the backward kernels return at once)
	uint64_t *aux_next;                               // side word of fq_next[slot] for re-seeding calls (r2text_kernel), or null
	BTask    *bq;                                     // backward task of forward task t: bq[t] (no atomics: 1:1)
	uint4    *lep; uint32_t lep_stride;               // LEP list of forward task t: lep + t*lep_stride
	unsigned long long *task_ctr;
	unsigned long long *n_queries;
	unsigned long long *err;                          // sticky: a queue overflowed
	unsigned long long *n_sst_hits;                   // bwt_extend queries answered by the on-device SST
	int32_t   sst;                                    // cs_params_t.sst_mode
	uint4    *sst2;                                   // second SST level (global, SST2_ENTRIES)
	const uint4 *jump; int32_t jump_k;                // round-3 jump table: bi-interval of every jump_k-mer (or null)
	unsigned long long *evc;                          // byte-model event counters [N_KID][N_EV] (fm_device.hpp), or null
};

__device__ __forceinline__ void emit_mem(const SplitArgs &A, uint32_t r, const Intv &v, uint32_t beg, uint32_t end)
{
	OutMem m = {v.x0, v.x1, v.x2, (uint64_t)beg << 32 | end};
	uint32_t k = atomicAdd(&A.out_cnt[r], 1u);
	if (k < A.cap) A.out[(size_t)r * A.cap + k] = m;
	else {
		unsigned long long s = atomicAdd(A.ovf_cnt, 1ull);
		if (s < A.ovf_cap) { OvfRec o = {m, r, 0}; A.ovf[s] = o; } else atomicMax(A.err, 1ull);
	}
}
__device__ __forceinline__ void push_ftask(const SplitArgs &A, uint64_t t, uint64_t aux = ~0ull) // one atomic per task: rare paths only
{
	unsigned long long s = atomicAdd(A.n_f_next, 1ull);
	if (s < A.fq_cap) { A.fq_next[s] = t; if (aux != ~0ull) A.aux_next[s] = aux; } else atomicMax(A.err, 2ull);
}
constexpr uint64_t FTASK_NONE = ~0ull; // kind bits = TK_NOP
constexpr uint64_t AUX_NONE = ~0ull, POS_NONE = ~0ull;
// an SMEM of a round-1/2 call: length filter (bwamem.c:232,246); returns the re-seeding call a round-1 SMEM triggers
// (bwamem.c:241-249) or FTASK_NONE.  aux: for the re-seeding call of a UNIQUE SMEM (min_intv 2), what r2text_kernel needs
// to find the SMEM in the text: x0 | beg << 37 | parity(beg + end) << 53.
__device__ __forceinline__ uint64_t emit_smem(const SplitArgs &A, uint32_t r, uint32_t kind, const Intv &v, int beg, uint32_t end, uint64_t &aux)
{
	int len = (int)end - beg;
	aux = AUX_NONE;
	if (len < A.min_seed_len) return FTASK_NONE;
	emit_mem(A, r, v, (uint32_t)beg, end);
	if (kind == TK_ROUND1 && len >= A.split_len && v.x2 <= A.split_width) {
		if (v.x2 == 1 && A.aux_next) aux = v.x0 | (uint64_t)beg << 37 | (uint64_t)(((uint32_t)beg + end) & 1u) << 53;
		return ftask_pack(r, (uint32_t)(beg + (int)end) >> 1, (uint32_t)v.x2 + 1, TK_ROUND2);
	}
	return FTASK_NONE;
}

// ------------------------------------------------------------------------------------------------------------------
// On-device SST (mapping/SST.h on the CPU): a transparent memo of bwt_extend, resident in LDS.
//
// The CPU SST is two tries (forward / backward) of bi-intervals keyed by the path of bases, reset every 512 reads.  A
// bi-interval is a function of the STRING alone, whichever direction it was reached from, so on the device one table
// keyed by the string serves both directions: entry (len, code) holds the interval of the string whose 2-bit packed
// bases are `code`.  It covers every string of up to SST_K bases (sum 4^d = 1364 entries x 16 B = 21.8 KB per
// workgroup, so occupancy is untouched), starts empty in every workgroup and is filled lazily: a miss costs exactly the
// bwt_extend it would have cost anyway and publishes the child; a hit answers from LDS with no HBM/L2 round trip.  Racing
// writers store identical values (the memoised function is pure), entries are single 16-byte LDS accesses.  Deeper
// strings are not cached: beyond ~12 bases every extension is a distinct random line whether a trie node or an Occ
// block answers it, so only an LDS-resident level set saves anything.
constexpr int SST_K = 5;
constexpr int SST_ENTRIES = 4 + 16 + 64 + 256 + 1024;
// Optional second level: strings of SST_K+1 .. SST2_K bases in a table in global memory (L2-resident, persistent across
// launches).  MEASURED AND SWITCHED OFF (SST2_K == SST_K): with SST2_K = 8 (1.4 MB) the hit rate rose from 8.8 % to
// 18.8 % on the bench workload but the SMEM stage got 10 % SLOWER (224 vs 204 ms per 10 M reads) -- the Occ records of
// such short strings are L2 hits already, so a hit only trades two record reads for one table read plus divergence.
// Only a level that answers without leaving the CU (LDS) pays.  The code path is kept for the record.
#ifndef CS_SST2_K
#define CS_SST2_K 5
#endif
constexpr int SST2_K = CS_SST2_K;
constexpr int SST2_ENTRIES = SST2_K > SST_K ? ((1 << (2 * (SST2_K + 1))) - 4096) / 3 : 16;
__device__ __forceinline__ int sst2_index(int len, uint32_t code) { return ((1 << (2 * len)) - 4096) / 3 + (int)code; } // len in 6..8
__device__ __forceinline__ int sst_index(int len, uint32_t code) // len in 1..SST_K
{
	return ((1 << (2 * len)) - 4) / 3 + (int)code; // 4 + 16 + ... + 4^(len-1) entries precede length `len`
}
__device__ __forceinline__ void sst_clear(uint4 *sst)
{
	for (int t = threadIdx.x; t < SST_ENTRIES; t += blockDim.x) sst[t] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
	__syncthreads();
}
__device__ __forceinline__ bool sst_get(const uint4 *sst, uint4 *sst2, int len, uint32_t code, Intv &v)
{
	uint4 e = len <= SST_K ? sst[sst_index(len, code)] : sst2[sst2_index(len, code)];
	if (e.w == 0xffffffffu) return false; // empty: a stored entry keeps its top 16 bits (the unused query end) zero
	uint32_t end; unpack_lep(e, v, end);
	return true;
}
__device__ __forceinline__ void sst_put(uint4 *sst, uint4 *sst2, int len, uint32_t code, const Intv &v)
{
	if (len <= SST_K) sst[sst_index(len, code)] = pack_lep(v, 0); else sst2[sst2_index(len, code)] = pack_lep(v, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// Round-3 jump table.  bwt_seed_strategy1 (bwt.c:358-379) extends forward from a start x and looks at the interval only
// once i - x >= min_seed_len, so the first min_seed_len bases of every segment are pure pointer chasing whose
// intermediate intervals nobody reads -- and at those depths the interval is still wide, so every step costs TWO random
// records.  The bi-interval of every k-mer (k = jump_k <= min_seed_len) is therefore precomputed once per engine into a
// table in HBM (4^15 x 16 B = 17 GB by default: this is what 288 GB are for) and a segment starts with ONE lookup instead of k - 1
// extensions.  A k-mer that does not occur has size 0 and every later extension keeps it at 0, exactly as in the
// reference, so the emitted seeds are unchanged; the skipped steps are counted as queries answered by the cache.
__global__ void jump_fill_kernel(const DevIndex ix, int k, uint4 *table)
{
	uint64_t n = 1ull << (2 * k);
	for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < n; m += (uint64_t)gridDim.x * blockDim.x) {
		Intv v = set_intv(ix, (int)((m >> (2 * (k - 1))) & 3));
		NoCtr W;
		for (int j = k - 2; j >= 0; --j) v = extend1<false>(ix, v, 3 - (int)((m >> (2 * j)) & 3), W);
		table[m] = pack_lep(v, 0);
	}
}

// The reads a second time, packed: one 16-byte record per 32 bases -- .x/.y the bases, 2 bits each, base j in bits 2j..2j+1 (the
// order of the 2-bit text), .z one bit per base that is ambiguous or lies behind the end of the read.  Record k of read r is
// rec[(off[r] >> 5) + r + k]: no second offset array, and read r owns at least len/32 + 1 records, so the record of position
// len exists and says "end" there.  Why: at 6-8 waves per SIMD the lanes in flight touch more lines than the L2 holds, so every
// 8-byte window of a byte-per-base read and every 4-byte word of the text came from HBM again (fwd0_kernel fetched 30 lines per
// read for 12 lines' worth of data); with 32 bases per load there is one fetch per record.
// RAW: straight from the caller's bytes (ASCII or nt4 codes; nst_nt4_table, FM_index/bntseq.c:46-63, codes 0..4 pass through as in
// comp_seed.cpp:2259) -- the byte-per-base nt4 copy is then only made when the fused kernel has to step in.  n_bases bounds the loads.
template <bool RAW>
__global__ void pack_reads_kernel(const uint8_t *seq, const uint64_t *off, int64_t n_reads, uint64_t n_bases, uint4 *rec)
{
	// letters -> codes through a 256-byte table in LDS (((c >> 1) ^ (c >> 2)) & 3 for A C G T in either case, the codes 0..3 as they
	// are, 4 for everything else): one LDS read per base instead of a dozen instructions
	__shared__ uint8_t lut[256];
	if (RAW) {
		for (uint32_t c = threadIdx.x; c < 256u; c += blockDim.x) {
			const uint32_t t = (c & 0xdfu) - 0x41u;
			const bool letter = t < 20u && ((0x80045u >> t) & 1u);
			lut[c] = (uint8_t)(c < 4u ? c : letter ? ((c >> 1) ^ (c >> 2)) & 3u : 4u);
		}
		__syncthreads();
	}
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = ((uint64_t)gridDim.x * blockDim.x) >> 3;
	for (uint64_t r = gid >> 3; r < (uint64_t)n_reads; r += gstride) { // eight lanes per read, a record each
		const uint64_t rb = off[r], re = off[r + 1], len = re - rb;
		const uint64_t w0 = (rb >> 5) + r, nrec = (re >> 5) + r + 1 - w0;
		for (uint64_t k = gid & 7; k < nrec; k += 8) {
			uint4 o = {0u, 0u, ~0u, 0u};
			if (k * 32 < len) {
				const uint64_t a = rb + k * 32, a0 = a & ~7ull;                // (the nt4 copy is padded by 64 bytes; the caller's buffer is not)
				const uint64_t *w = reinterpret_cast<const uint64_t *>(seq + a0);
				const uint32_t sh = (uint32_t)(a - a0) << 3;
				uint64_t v[5];
#pragma unroll
				for (int q = 0; q < 5; ++q) {
					const uint64_t wa = a0 + 8u * (uint32_t)q;
					if (!RAW || wa + 8 <= n_bases) v[q] = w[q];
					else { // the word that holds the caller's last bytes: assembled bytewise, nothing behind n_bases is read
						uint64_t t = 0x0404040404040404ull;
						for (uint64_t z = wa; z < n_bases; ++z) t = (t & ~(0xffull << ((z - wa) << 3))) | (uint64_t)seq[z] << ((z - wa) << 3);
						v[q] = t;
					}
				}
				uint64_t bases = 0; uint32_t bad = 0;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					uint64_t b8 = sh ? (v[q] >> sh) | (v[q + 1] << (64u - sh)) : v[q]; // bases 8q .. 8q+7, a byte each
					if (RAW) {
						uint64_t c8 = 0;
#pragma unroll
						for (int z = 0; z < 8; ++z) c8 |= (uint64_t)lut[(uint32_t)(b8 >> (8 * z)) & 0xffu] << (8 * z);
						b8 = c8;
					}
					bad |= (uint32_t)((((b8 >> 2) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * q);
					b8 &= 0x0303030303030303ull;
					b8 = (b8 | (b8 >> 6)) & 0x000F000F000F000Full;
					b8 = (b8 | (b8 >> 12)) & 0x000000FF000000FFull;
					bases |= ((b8 | (b8 >> 24)) & 0xFFFFull) << (16 * q);
				}
				const uint64_t left = len - k * 32;
				if (left < 32) { bad |= ~0u << (uint32_t)left; bases &= (1ull << (2 * (uint32_t)left)) - 1ull; }
				o.x = (uint32_t)bases; o.y = (uint32_t)(bases >> 32); o.z = bad;
			}
			rec[w0 + k] = o;
		}
	}
}
// The way back, for the rare case that the host made the records (host_pack.cpp) and the fused kernel has to step in: a byte per
// base, codes 0..3, 4 for an ambiguous base.
__global__ void unpack_reads_kernel(const uint4 *rec, const uint64_t *off, int64_t n_reads, uint8_t *out)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = ((uint64_t)gridDim.x * blockDim.x) >> 3;
	for (uint64_t r = gid >> 3; r < (uint64_t)n_reads; r += gstride) {
		const uint64_t rb = off[r], len = off[r + 1] - rb;
		const uint4 *rr = rec + (rb >> 5) + r;
		for (uint64_t k = gid & 7; k * 32 < len; k += 8) {
			const uint4 v = rr[k];
			const uint64_t bases = (uint64_t)v.x | (uint64_t)v.y << 32, left = len - k * 32;
			for (uint32_t j = 0; j < 32u && j < left; ++j) out[rb + k * 32 + j] = (uint8_t)((v.z >> j) & 1u ? 4u : (uint32_t)(bases >> (2 * j)) & 3u);
		}
	}
}
// reader over those records: any position of the read, one load per record entered
struct PackedReader {
	const uint4 *rec; uint64_t bases; uint32_t bad; int wk;
	__device__ __forceinline__ void load() { const uint4 v = rec[wk]; bases = (uint64_t)v.x | (uint64_t)v.y << 32; bad = v.z; }
	__device__ __forceinline__ void start(const uint4 *recs, uint64_t rb, uint32_t r, int pos)
	{
		rec = recs + (rb >> 5) + r; wk = (pos < 0 ? 0 : pos) >> 5; load();
	}
	__device__ __forceinline__ void seek(int pos) { if ((pos >> 5) != wk) { wk = pos >> 5; load(); } }
	__device__ __forceinline__ uint32_t at(int pos) // 0..3, or 4: ambiguous base / behind the end (pos <= len)
	{
		seek(pos);
		const uint32_t j = (uint32_t)pos & 31u;
		return (bad >> j) & 1u ? 4u : (uint32_t)(bases >> (j << 1)) & 3u;
	}
	// the 32 bases from pos on (2 bits each, the first least significant) and their ambiguity bits; the nb (<= 32) first of them must
	// lie inside the read.  The reader stays on the record of pos.
	__device__ __forceinline__ uint64_t window(int pos, int nb, uint32_t &badw)
	{
		seek(pos);
		const uint32_t j = (uint32_t)pos & 31u;
		uint64_t w = bases >> (j << 1); badw = bad >> j;
		if (j + (uint32_t)nb > 32u) { const uint4 v = rec[wk + 1]; w |= ((uint64_t)v.x | (uint64_t)v.y << 32) << ((32u - j) << 1); badw |= v.z << (32u - j); }
		return w;
	}
	// the jk (<= 16) bases from pos on as a jump-table code (first base most significant); pos + jk <= len
	__device__ __forceinline__ uint32_t kmer(int pos, int jk, uint32_t &badk)
	{
		seek(pos);
		const uint32_t j = (uint32_t)pos & 31u;
		uint64_t w = bases >> (j << 1); uint32_t bd = bad >> j;
		if (j + (uint32_t)jk > 32u) { // j >= 17: the code runs into the next record (left loaded: the caller goes on from there)
			++wk; load();
			w |= bases << ((32u - j) << 1); bd |= bad << (32u - j);
		}
		badk = (bd & ((1u << jk) - 1u)) ? 4u : 0u;
		uint32_t rv = __brev((uint32_t)w);                           // group q at 2(15-q), its two bits swapped
		rv = ((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1);
		return rv >> (32 - 2 * jk);
	}
};
// text mode: up to 32 read bases from i against the text from tpos (<= seq_len); true when the match ends here.  Counts what the
// reference would have performed: one bwt_extend per base that is compared (the last one, at a mismatch or the text's end, returns
// size 0); none at an ambiguous base or the read's end (bwt.c:309-316).
template <class WC>
__device__ __forceinline__ bool text_step(const DevIndex &ix, PackedReader &rd, int &i, uint64_t &tpos, uint32_t &my_q, uint32_t &my_hits, WC &W)
{
	rd.seek(i);
	const uint32_t j = (uint32_t)i & 31u, avail = 32u - j;
	const uint64_t x = (rd.bases >> (j << 1)) ^ text_win(ix, tpos); wc_add(W, EV_TEXT, 4u);
	const uint64_t d = (x | x >> 1) & 0x5555555555555555ull;
	uint32_t m = d ? (uint32_t)(__ffsll((long long)d) - 1) >> 1 : 32u;   // first base that differs
	const uint64_t room = ix.seq_len - tpos;
	if (room < m) m = (uint32_t)room;                                    // ... or has no text base to agree with
	const uint32_t bb = rd.bad >> j, m_bad = bb ? (uint32_t)__ffs((int)bb) - 1u : 32u;
	uint32_t n = m < m_bad ? m : m_bad;
	if (n > avail) n = avail;
	i += (int)n; tpos += n; my_q += n; my_hits += n;
	if (n == avail) return false;                                        // the record is used up: on with the next one
	if (n != m_bad) { ++my_q; ++my_hits; }
	return true;
}

// ------------------------------------------------------------------------------------------------------------------
// initial tasks: round-1 call at the first unambiguous base, and the round-3 chain (bwamem.c:226, 253)
__global__ void init_tasks_kernel(const SplitArgs A, uint64_t *fq, uint64_t *fq_r3)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= A.n_reads) return;
	uint64_t b = A.off[r]; int len = (int)(A.off[r + 1] - b), x = 0;
	PackedReader rd; rd.start(A.seqp, b, (uint32_t)r, 0);
	while (x < len && rd.at(x) > 3) ++x;
	// the round-3 chains get a queue of their own: they depend on nothing and run on a second stream (engine.hip)
	fq[r] = x < len ? ftask_pack((uint32_t)r, (uint32_t)x, 1, TK_ROUND1) : ftask_pack((uint32_t)r, 0, 0, TK_NOP);
	fq_r3[r] = (len > 0 && A.max_mem_intv > 0) ? ftask_pack((uint32_t)r, 0, 0, TK_ROUND3) : ftask_pack((uint32_t)r, 0, 0, TK_NOP);
}

// Task dispenser.  One returning atomic on a single word costs ~11 ns and the word saturates near 88 M dequeues/s
// (MI355X_MICROARCH.md "dequeue"), far below the millions of short tasks per launch here, so a wave draws REFILL task
// ids at a time with ONE atomic and hands them to its lanes with a ballot + popcount.  All state is wave-uniform.
struct WavePool { uint64_t cur, end; bool exhausted; };

template <int REFILL>
__device__ __forceinline__ bool pool_take(WavePool &P, bool want, unsigned long long *ctr, uint64_t n_tasks, uint64_t &task)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint64_t m = __ballot(want);
	if (m == 0) return false;
	if (P.cur == P.end && !P.exhausted) {
		int src = __ffsll((long long)m) - 1;
		unsigned long long base = 0;
		if ((int)lane == src) base = atomicAdd(ctr, (unsigned long long)REFILL);
		base = __shfl(base, src);
		if (base >= n_tasks) P.exhausted = true;
		else { P.cur = base; P.end = base + REFILL < n_tasks ? base + REFILL : n_tasks; }
	}
	uint64_t avail = P.end - P.cur, cnt = (uint64_t)__popcll(m);
	uint64_t rank = (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
	task = P.cur + rank;
	P.cur += cnt < avail ? cnt : avail;
	return want && rank < avail;
}

// The reverse direction: a wave reserves RES slots of the next forward queue with one atomic and its lanes fill them
// (ballot + popcount); slots left over when the wave moves on are filled with no-op tasks.
struct WaveOut { uint64_t cur, end; };
template <int RES>
__device__ __forceinline__ void wave_push(WaveOut &O, bool want, uint64_t task, const SplitArgs &A, uint64_t aux = AUX_NONE)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint64_t m = __ballot(want);
	if (m == 0) return;
	uint64_t cnt = (uint64_t)__popcll(m);
	if (O.end - O.cur < cnt) {
		uint64_t rem = O.end - O.cur;
		if (lane < rem) A.fq_next[O.cur + lane] = FTASK_NONE;
		int src = __ffsll((long long)m) - 1;
		unsigned long long base = 0;
		if ((int)lane == src) base = atomicAdd(A.n_f_next, (unsigned long long)RES);
		base = __shfl(base, src);
		if (base + RES > A.fq_cap) { if ((int)lane == src) atomicMax(A.err, 2ull); O.cur = O.end = 0; return; }
		O.cur = base; O.end = base + RES;
	}
	uint64_t rank = (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
	if (want) { A.fq_next[O.cur + rank] = task; if (aux != AUX_NONE) A.aux_next[O.cur + rank] = aux; }
	O.cur += cnt;
}
__device__ __forceinline__ void wave_push_finish(WaveOut &O, const SplitArgs &A)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint64_t rem = O.end - O.cur;
	if (lane < rem) A.fq_next[O.cur + lane] = FTASK_NONE;
	O.cur = O.end;
}

// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// A filter over ALL min_seed_len-mers of the text (both strands: the text holds both): 2^bloom_bits 64-bit words, two bits
// per k-mer inside one word chosen by a hash.  "Both bits set" has ~1 % false positives, "not both" is exact: that k-mer
// does not occur.  The window lanes of the backward kernels ask it before anything else -- a chance window exists with
// probability 2 %, so 97 % of them end after one 8-byte read instead of a jump-table read and two to four extensions.
// Codes are 2 bits per base, first base least significant (the order of the 2-bit text), k <= 32.
__device__ __forceinline__ uint64_t kmer_hash(uint64_t code) { uint64_t h = code * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
__device__ __forceinline__ bool kmer_filter_has(const uint64_t *bloom, uint32_t bits, uint64_t code)
{
	const uint64_t h = kmer_hash(code);
	const uint64_t w = bloom[h >> (64u - bits)];
	return ((w >> (h & 63u)) & (w >> ((h >> 6) & 63u)) & 1ull) != 0;
}
__global__ void kmer_filter_fill_kernel(const DevIndex ix, int k, uint64_t *bloom, uint32_t bits)
{
	const uint64_t n = ix.seq_len >= (uint64_t)k ? ix.seq_len - (uint64_t)k + 1 : 0;
	const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t h = kmer_hash(text_win(ix, p) & mask);
		atomicOr((unsigned long long *)&bloom[h >> (64u - bits)], (1ull << (h & 63u)) | (1ull << ((h >> 6) & 63u)));
	}
}
// The re-seeding call of a unique SMEM, the part most calls come to: in sequence that is not repeated, rep[] stays
// below min_seed_len around the pivot.  If it does at all the min_seed_len offsets up to the pivot (and is never 0), the sweep of
// r2_by_text ends within them (o + rep[o] <= pivot at the latest at o = pivot - min_seed_len + 1), touches neither end of the SMEM and
// reports nothing: the call is answered by three or four words of rep[].  0: answered; 1: needs the sweep.  P: text position of the
// SMEM's first base, len its length, po the pivot's offset in it; nw: words read.
__device__ __forceinline__ int r2_quick_rep(const DevIndex &ix, uint64_t P, int len, int po, int k, uint32_t &nw)
{
	nw = 0;
	if (po < k || po + k > len || k > 32) return 1;
	const uint64_t lo = P + (uint64_t)(po - k + 1), hi = P + (uint64_t)po;   // the bytes rep[lo .. hi]
	const uint64_t *wp = reinterpret_cast<const uint64_t *>(ix.rep) + (lo >> 3);
	nw = (uint32_t)((hi >> 3) - (lo >> 3)) + 1u;                               // 3..5 aligned words for k <= 32
	uint64_t wd[5];
#pragma unroll
	for (int q = 0; q < 5; ++q) wd[q] = (uint32_t)q < nw ? wp[q] : 0x0101010101010101ull;
	const uint64_t ones = 0x0101010101010101ull, top = 0x8080808080808080ull;
	const uint64_t fl = ((lo & 7) ? ~0ull << ((lo & 7) << 3) : ~0ull), fh = ~0ull >> ((7 - (hi & 7)) << 3); // bytes of the first / last word that count
	uint64_t bad = 0;
#pragma unroll
	for (int q = 0; q < 5; ++q) {
		const uint64_t w = wd[q];
		uint64_t f = ((((w & ~top) + (uint64_t)(0x80 - k) * ones) | w) & top)    // a byte >= k
		           | ((w - ones) & ~w & top);                                    // a byte == 0
		if (q == 0) f &= fl;
		if ((uint32_t)q + 1u == nw) f &= fh;
		if ((uint32_t)q < nw) bad |= f;
	}
	return bad ? 1 : 0;
}
// the call that follows a finished round-1 call at pivot x_cur: next pivot = end of the longest forward match, ambiguous
// bases skipped.  A round-1 task carries, in the field that holds min_intv for round 2 (round 1 always uses 1), the
// distance to the previous pivot + 1 when the forward pass ended ON the new pivot (no ambiguous base in between): no
// match that starts at or before the previous pivot reaches beyond the new one, which bounds the new call's sweep.
__device__ __forceinline__ uint64_t chain_round1(PackedReader &rd, uint32_t r, int len, int ret, int x_cur)
{
	int x = ret;
	while (x < len && rd.at(x) > 3) ++x;
	uint32_t d = (x == ret && ret - x_cur < 16382) ? (uint32_t)(ret - x_cur) : 0u;
	return x < len ? ftask_pack(r, (uint32_t)x, 1u + d, TK_ROUND1) : FTASK_NONE;
}

template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, 6) void fwd_kernel(const SplitArgs A)
{
	const DevIndex &ix = A.ix;
	bool active = false;
	uint64_t tslot = 0; uint32_t r = 0, kind = 0, min_intv = 1, dprev = 0;
	int len = 0, x = 0, i = 0, n = 0;
	Intv ik = {0, 0, 0};
	PackedReader rd;
	uint4 *lep = nullptr;
	uint32_t my_q = 0, my_hits = 0, my_sw = 0; // per-lane counters (a lane sees a few thousand extensions at most); bit 31 of my_sw: created a backward task
	WavePool P = {0, 0, false};
	WaveOut O = {0, 0};
	WaveCtrT<COUNT> W;
	__shared__ uint4 sst[SST_ENTRIES];
	sst_clear(sst);
	const bool use_sst = A.sst != 0;
	int slen = 0; uint32_t scode = 0; // the string matched so far, while it is short enough for the SST
	// Text mode.  Once the forward match of an SMEM call occurs exactly once, every further bwt_extend only re-ranks that one
	// occurrence: its size stays 1 until the read and the text disagree (bwt.c:309-316 pushes nothing in between).  So the
	// lane looks the occurrence's text position up in the suffix array once, compares the read against the 2-bit text eight
	// bases per iteration without touching the index, and at the end takes the reverse-strand coordinate from the inverse
	// suffix array (the forward coordinate of a unique match does not move).  Two random reads replace ~80 per read.
	const bool text_on = use_sst && ix.text2 != nullptr;
	// (text mode is kind == TK_TEXT; the text cursor lives in ik.x1)
	const int jump_k = (use_sst && A.jump && A.jump_k <= A.min_seed_len) ? A.jump_k : 0;
	// start a round-3 segment at x: through the jump table when the next jump_k bases are all A/C/G/T, else base by base
	auto r3_start = [&]() -> bool { // true: the jump table was used
		if (jump_k && x + jump_k <= len) {
			uint32_t bad; const uint32_t code = rd.kmer(x, jump_k, bad);
			if (bad == 0) {
				uint32_t e; unpack_lep(A.jump[code], ik, e); wc_add(W, EV_JUMP);
				i = x + jump_k; slen = jump_k; scode = 0;
				my_q += (unsigned)(jump_k - 1); my_hits += (unsigned)(jump_k - 1);
				return true;
			}
		}
		scode = rd.at(x); slen = 1;
		ik = set_intv(ix, (int)scode); i = x + 1;
		return false;
	};
	for (;;) {
		uint64_t t_id = 0;
		bool got = pool_take<256>(P, !active, A.task_ctr, A.n_f, t_id);
		if (!active && got) {
			tslot = t_id;
			uint64_t t = A.fq[tslot];
			kind = (uint32_t)(t >> 62);
			r = (uint32_t)t; x = (int)((t >> 32) & 0xffffu); min_intv = (uint32_t)((t >> 48) & 0x3fffu);
			dprev = 0;
			if (kind == TK_ROUND1) { dprev = min_intv - 1; min_intv = 1; } // see chain_round1
			if (kind != TK_NOP && (int64_t)r < A.n_reads) {
				uint64_t rb = A.off[r]; len = (int)(A.off[r + 1] - rb);
				if (x < len) {
					rd.start(A.seqp, rb, r, x);
					lep = A.lep + tslot * A.lep_stride; n = 0;
					if (kind == TK_ROUND3) while (x < len && rd.at(x) > 3) ++x; // first start (bwamem.c:255-256)
					if (x < len) {
						if (kind == TK_ROUND3) r3_start();
						else {
							// A round-1 call at pivot 0 keeps no LEPs (below), so its first jump_k steps can come from the
							// jump table too, provided the jump_k-mer occurs at all (otherwise: step by step, to find where it stops)
							// (the same holds for every call under the window scheme: LEPs shorter than min_seed_len are not stored)
							bool start = true;
							if (((x == 0 && kind == TK_ROUND1) || A.win) && jump_k) {
								start = false;
								if (r3_start() && ik.x2 < min_intv) { my_q -= (unsigned)(jump_k - 1); my_hits -= (unsigned)(jump_k - 1); start = true; }
							}
							if (start) { scode = rd.at(x); slen = 1; ik = set_intv(ix, (int)scode); i = x + 1; }
						}
						active = true;
					}
				}
			}
		}
		if (P.exhausted && __ballot(active) == 0) break; // wave-uniform exit
		uint64_t push0 = FTASK_NONE, push1 = FTASK_NONE, aux0 = AUX_NONE; // forward tasks this lane spawns in this step
		if (active) {
			bool fin = false; // the forward pass of an SMEM call ends in this iteration with ik = [x, i)
			if (kind == TK_TEXT) {
				uint64_t tpos = ik.x1;
				fin = text_step(ix, rd, i, tpos, my_q, my_hits, W);        // up to 32 bases against the text
				ik.x1 = tpos;
				if (fin) {
					ik.x1 = isa_direct(ix, ix.seq_len - tpos); kind = TK_ROUND1; // rank of the reverse complement of [x, i)
					wc_add(W, EV_ISA);
					// The sweep of this call cannot pass the previous pivot x - dprev (chain_round1), and while the unique
					// match keeps agreeing with the text in front of it, it stays the longest survivor and nothing else is
					// reported (bwt.c:328-336).  So if the dprev - 1 bases between the pivots agree, the whole sweep reports
					// exactly one SMEM, [x - dprev + 1, i), and its bi-interval comes from the inverse suffix array.
					if (dprev > 0 && A.text_sweep) {
						const int nb = (int)dprev - 1;
						const uint64_t px = tpos - (uint64_t)(i - x); // text position of read base x
						bool same = px >= (uint64_t)nb;
						for (int done = 0; same && done < nb;) { // read [x - nb, x) against the text in front of px, a record at a time
							const int p = x - nb + done;
							rd.seek(p);
							const uint32_t j = (uint32_t)p & 31u, room = 32u - j, n = (uint32_t)(nb - done) < room ? (uint32_t)(nb - done) : room;
							const uint64_t dx = (rd.bases >> (j << 1)) ^ text_win(ix, px - (uint64_t)(nb - done)); wc_add(W, EV_TEXT, 4u);
							const uint64_t keep = n >= 32u ? ~0ull : (1ull << (n << 1)) - 1ull;
							same = (dx & keep) == 0 && ((rd.bad >> j) & (uint32_t)(n >= 32u ? ~0u : (1u << n) - 1u)) == 0;
							done += (int)n;
						}
						if (same) {
							Intv m = {isa_direct(ix, px - (uint64_t)nb), ik.x1, 1}; wc_add(W, EV_ISA);
							push0 = emit_smem(A, r, TK_ROUND1, m, x - nb, (uint32_t)i, aux0);
							push1 = chain_round1(rd, r, len, i, x);
							++my_sw; active = false; fin = false;
						}
					}
				}
			} else {
				// ---- the one extension site: forward by read base i (bwt.c:309-311 / 368-369)
				uint32_t b = i < len ? rd.at(i) : 4u;
				Intv y = ik;
				bool cached = false, cacheable = use_sst && b <= 3 && slen < SST2_K;
				uint32_t ccode = scode << 2 | b;                    // the string extended by read base b
				if (cacheable) cached = sst_get(sst, A.sst2, slen + 1, ccode, y);
				if (b <= 3) {
					++my_q;
					if (cached) ++my_hits;
					else { y = extend1<false>(ix, ik, 3 - (int)b, W); if (cacheable) sst_put(sst, A.sst2, slen + 1, ccode, y); }
					scode = ccode; ++slen;                          // slen keeps counting; only values < SST2_K are looked at
				}
				if (kind == TK_ROUND3) { // bwt_seed_strategy1, bwt.c:366-377
					if (b <= 3 && !(y.x2 < A.max_mem_intv && i - x >= A.min_seed_len)) { ik = y; ++i; }
					else if (b > 3 && i >= len) active = false;
					else {
						if (b <= 3 && y.x2 > 0) emit_mem(A, r, y, (uint32_t)x, (uint32_t)(i + 1));
						x = i + 1; // restart behind the seed / the ambiguous base
						while (x < len && rd.at(x) > 3) ++x;
						if (x >= len) active = false; else r3_start();
					}
				} else { // ---- forward pass of an SMEM call, bwt.c:303-320
					const bool changed = b > 3 || y.x2 != ik.x2;            // read end (i == len), ambiguous base, or size change
					fin = b > 3 || (y.x2 != ik.x2 && y.x2 < min_intv);
					if (changed && !fin && x != 0 && (!A.win || i - x >= A.min_seed_len)) { lep[n++] = pack_lep(ik, (uint32_t)i); wc_add(W, EV_LEP); }
					if (!fin) {
						ik = y; ++i;
						if (text_on && ik.x2 == 1 && kind == TK_ROUND1) { // unique from here on: continue on the text
							const uint64_t tp = sa_direct(ix, ik.x0) + (uint64_t)(i - x); // text cursor: the base that has to equal read base i
							wc_add(W, EV_SA);
							if (tp <= ix.seq_len) { kind = TK_TEXT; ik.x1 = tp; }  // (always: the match lies inside the text)
						}
					}
				}
			}
			if (fin) { // ik = the longest forward match [x, i): the last LEP (bwt.c:307/315/320)
				// A call at pivot 0 has a trivial backward sweep (bwt.c:325 starts at i = -1): its only SMEM is the longest
				// forward match, so it needs no LEP list, no backward task, and finishes right here.
				if (x == 0) {
					push0 = emit_smem(A, r, kind, ik, 0, (uint32_t)i, aux0);
					if (kind == TK_ROUND1) push1 = chain_round1(rd, r, len, i, x);
				} else { // hand the list to the backward kernel of its size class; ret = end of the longest match = next pivot
					if (!A.win || i - x >= A.min_seed_len) { lep[n++] = pack_lep(ik, (uint32_t)i); wc_add(W, EV_LEP); }
					uint32_t cls = n <= 16 ? 0u : n <= 32 ? 1u : n <= 64 ? 2u : 3u;
					if (A.win) cls = n == 0 ? 6u : n <= WIN_G32_LEPS ? 4u : n <= WIN_G64_LEPS ? 5u : 3u;
					BTask bt = {r, (uint16_t)x, (uint16_t)(min_intv | kind << 14), (uint16_t)n, (uint16_t)i, cls};
					A.bq[tslot] = bt;
					my_sw |= 0x80000000u;
				}
				active = false;
			}
		}
		wave_push<64>(O, push0 != FTASK_NONE, push0, A, aux0);
		wave_push<64>(O, push1 != FTASK_NONE, push1, A);
	}
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, (unsigned long long)my_q);
	if (my_hits) atomicAdd(A.n_sst_hits, (unsigned long long)my_hits);
	if (my_sw & 0x80000000u) atomicAdd(A.n_btasks, 1ull);
	if (my_sw & 0x7fffffffu) atomicAdd(A.n_text_sweeps, (unsigned long long)(my_sw & 0x7fffffffu));
	wc_flush(W, A.evc, KID_FWD);
}


// The first launch of a batch: every call is a round-1 call at the read's first base, and almost all of them go jump table
// -> a few extensions until the match is unique -> text mode -> one SMEM, the next pivot, a re-seeding candidate.  None of
// that needs LEPs, backward tasks, the SST or round 3, so these calls get a kernel without them: half the registers of
// fwd_kernel (the calls are latency-bound, so resident waves are what counts).  It takes the calls it can start from the
// jump table and replaces them by no-ops in the queue; whatever is left (reads that begin with an ambiguous base, are
// shorter than jump_k, or whose first jump_k-mer does not occur) is fwd_kernel's, launched over the same queue afterwards.
template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, 8) void fwd0_kernel(const SplitArgs A, uint64_t *fq)
{
	const DevIndex &ix = A.ix;
	const int jk = A.jump_k;
	bool active = false, textm = false;
	uint32_t r = 0;
	int len = 0, i = 0;
	Intv ik = {0, 0, 0};
	PackedReader rd;
	uint32_t my_q = 0, my_hits = 0, my_r2 = 0;
	WavePool P = {0, 0, false};
	WaveOut O = {0, 0};
	WaveCtrT<COUNT> W;
	for (;;) {
		uint64_t t_id = 0;
		bool got = pool_take<256>(P, !active, A.task_ctr, A.n_f, t_id);
		if (!active && got) {
			const uint64_t t = fq[t_id];
			r = (uint32_t)t;
			if ((uint32_t)(t >> 62) == TK_ROUND1 && ((t >> 32) & 0xffffu) == 0 && (int64_t)r < A.n_reads) {
				const uint64_t rb = A.off[r]; len = (int)(A.off[r + 1] - rb);
				if (len >= jk) {
					rd.start(A.seqp, rb, r, 0);
					uint32_t bad; const uint32_t code = rd.kmer(0, jk, bad);
					uint32_t e; unpack_lep(A.jump[code], ik, e); wc_add(W, EV_JUMP);
					if (bad <= 3 && ik.x2 > 0) {
						fq[t_id] = FTASK_NONE;                       // ours
						i = jk; textm = false; active = true;
						my_q += (unsigned)(jk - 1); my_hits += (unsigned)(jk - 1);
					}
				}
			}
		}
		if (P.exhausted && __ballot(active) == 0) break; // wave-uniform exit
		uint64_t push0 = FTASK_NONE, push1 = FTASK_NONE, aux0 = AUX_NONE;
		if (active) {
			bool fin = false;
			uint64_t fpos = POS_NONE; // text position of the read's first base, where the match ended in text mode
			if (textm) { // as in fwd_kernel: the unique match against the 2-bit text, cursor in ik.x1
				uint64_t tpos = ik.x1;
				fin = text_step(ix, rd, i, tpos, my_q, my_hits, W);
				ik.x1 = tpos;
				if (fin) { fpos = tpos - (uint64_t)i; ik.x1 = isa_direct(ix, ix.seq_len - tpos); wc_add(W, EV_ISA); }
			} else {
				uint32_t b = i < len ? rd.at(i) : 4u;
				if (b > 3) fin = true;
				else {
					++my_q;
					const Intv y = extend1<false>(ix, ik, 3 - (int)b, W);
					if (y.x2 == 0) fin = true;                       // bwt.c:313-315 with min_intv = 1
					else {
						ik = y; ++i;
						if (ik.x2 == 1) {
							const uint64_t tp = sa_direct(ix, ik.x0) + (uint64_t)i;
							wc_add(W, EV_SA);
							if (tp <= ix.seq_len) { textm = true; ik.x1 = tp; } // (always: the match lies inside the text)
						}
					}
				}
			}
			if (fin) { // the call's only SMEM is its longest forward match (bwt.c:325 starts the sweep at -1)
				push0 = emit_smem(A, r, TK_ROUND1, ik, 0, (uint32_t)i, aux0);
				if (push0 != FTASK_NONE && aux0 != AUX_NONE && fpos != POS_NONE && ix.rep) { // its re-seeding call, if rep[] settles it right here
					uint32_t nw = 0; const int pv = i >> 1;                                    // (emit_smem: pivot = (beg + end) / 2, beg = 0)
					if (A.min_seed_len >= 2 && fpos + (uint64_t)i <= ix.seq_len && pv <= 4096 && r2_quick_rep(ix, fpos, i, pv, A.min_seed_len, nw) == 0) { push0 = FTASK_NONE; ++my_r2; }
					wc_add(W, EV_REP, nw);
				}
				push1 = chain_round1(rd, r, len, i, 0);
				active = false;
			}
		}
		if (__ballot((push0 & push1) != FTASK_NONE)) {
			wave_push<64>(O, push0 != FTASK_NONE, push0, A, aux0);
			wave_push<64>(O, push1 != FTASK_NONE, push1, A);
		}
	}
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, (unsigned long long)my_q);
	if (my_hits) atomicAdd(A.n_sst_hits, (unsigned long long)my_hits);
	if (my_r2) atomicAdd(A.n_r2_quick, (unsigned long long)my_r2);
	wc_flush(W, A.evc, KID_FWD0);
}

// ------------------------------------------------------------------------------------------------------------------
template <int G, class WC>
__device__ __forceinline__ void bwd_groups_run(const SplitArgs &A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctr, WaveOut &O,
                                               unsigned long long &my_q, unsigned long long &my_hits, uint4 *sst, WC &W)
{
	const bool use_sst = A.sst != 0;
	int slen = SST2_K; uint32_t scode = 0; // this lane's match as a string, while it is short enough for the SST
	constexpr uint32_t MYCLS = G == 16 ? 0u : G == 32 ? 1u : 2u;
	const DevIndex &ix = A.ix;
	const uint32_t lane = threadIdx.x & 63u, gl = lane % G, gbase = lane - gl; // group = G consecutive lanes of a wave
	const uint64_t gmask = (G >= 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << gbase);
	bool active = false, live = false;
	uint32_t r = 0, kind = 0, min_intv = 1, pend = 0;
	int i = 0, ret = 0, nm = 0, last_start = 0, xp = 0;
	Intv e = {0, 0, 0};
	PackedReader rd;

	// Every lane of the wave stays in the loop until the whole wave is done, and every lane executes the dispenser code at
	// the top and the bottom of each iteration, so its wave-uniform state stays identical in all lanes.
	//
	// Task acquisition: the backward tasks sit in the forward tasks' slots, each tagged with its size class.  A wave takes
	// 64 consecutive slots with one atomic, its 64 lanes read the 64 tags in one coalesced load, and a ballot gives the mask
	// of slots that belong to this kernel's class; idle groups then pop slots off that mask.  Skipping foreign slots costs
	// nothing per slot, so every size class can scan the whole queue.
	uint64_t batch_base = 0, avail_m = 0; bool exhausted = false;
	for (;;) {
		uint64_t idle_m = __ballot(!active && gl == 0);
		if (idle_m != 0 && avail_m == 0 && !exhausted) {
			int src = __ffsll((long long)idle_m) - 1;
			unsigned long long base = 0;
			if ((int)lane == src) base = atomicAdd(ctr, 64ull);
			base = __shfl(base, src);
			if (base >= n_tasks) exhausted = true;
			else {
				uint64_t slot = base + lane;
				uint32_t cls = slot < n_tasks ? bq[slot].cls : 0xffffffffu;
				avail_m = __ballot(cls == MYCLS);
				batch_base = base;
			}
		}
		if (idle_m != 0 && avail_m != 0) {
			// the k-th idle group (in lane order) takes the k-th set bit of avail_m
			int k = __popcll(idle_m & ((1ull << gbase) - 1ull));
			uint64_t m = avail_m;
			for (int q = 0; q < k; ++q) m &= m - 1;
			bool mine = !active && m != 0;
			uint64_t t = batch_base + (uint64_t)(__ffsll((long long)m) - 1);
			int taken = __popcll(idle_m), have = __popcll(avail_m);
			if (taken > have) taken = have;
			for (int q = 0; q < taken; ++q) avail_m &= avail_m - 1;
			if (mine) {
				BTask bt = bq[t];
				r = bt.r; kind = bt.mi_kind >> 14; min_intv = bt.mi_kind & 0x3fffu; ret = bt.ret;
				int x = bt.x, n = bt.n; xp = x;
				live = (int)gl < n;
				if (live) { unpack_lep(A.lep[(size_t)t * A.lep_stride + (n - 1 - (int)gl)], e, pend); wc_add(W, EV_LEP); }
				uint64_t rb = A.off[r];
				rd.start(A.seqp, rb, r, x - 1);
				i = x - 1; nm = 0; last_start = 0;
				slen = SST2_K; scode = 0;
				if (use_sst && live && (int)pend - x < SST2_K) { // a short LEP: spell it, the SST is keyed by the string
					slen = (int)pend - x;
					for (int q = 0; q < slen; ++q) scode = scode << 2 | rd.at(x + q);
				}
				active = true;
			}
		}
		if (exhausted && avail_m == 0 && __ballot(active) == 0) break; // wave-uniform exit
		uint64_t push0 = FTASK_NONE, push1 = FTASK_NONE, aux0 = AUX_NONE; // forward tasks this lane spawns in this step
		if (active) {
			uint32_t b = i < 0 ? 4u : rd.at(i);
			uint64_t live_m = __ballot(live) & gmask;
			int first = __ffsll((long long)live_m) - 1; // the longest live match of the group
			bool end_call = false;
			if (b > 3) { // read start or ambiguous base (bwt.c:326): every live match stops; only the longest can be new
				if ((int)lane == first && (nm == 0 || i + 1 < last_start)) push0 = emit_smem(A, r, kind, e, i + 1, pend, aux0);
				end_call = true;
			} else {
				Intv y = e;
				bool cacheable = use_sst && live && slen < SST2_K, cached = false;
				uint32_t ccode = b << (2 * slen) | scode;       // read base b in front of the string
				if (cacheable) cached = sst_get(sst, A.sst2, slen + 1, ccode, y);
				if (live) {
					++my_q;
					if (cached) ++my_hits;
					else { y = extend1<true>(ix, e, (int)b, W); if (cacheable) sst_put(sst, A.sst2, slen + 1, ccode, y); }
					if (slen < SST2_K) { scode = ccode; ++slen; }
				}
				bool stop = live && y.x2 < min_intv, cand = live && !stop;
				uint64_t cand_m = __ballot(cand) & gmask;
				// bwt.c:328-336: the first live match is an SMEM if it stops here (nothing longer survived) and is not contained
				bool first_stops = !((cand_m >> first) & 1ull);
				if (first_stops && (nm == 0 || i + 1 < last_start)) {
					if ((int)lane == first) push0 = emit_smem(A, r, kind, e, i + 1, pend, aux0);
					++nm; last_start = i + 1;
				}
				// bwt.c:337-340: keep a surviving match unless its size equals that of the previous surviving one
				uint64_t before = cand_m & ((1ull << lane) - 1ull);
				int prev = before ? 63 - __clzll((long long)before) : (int)lane;
				uint64_t prev_x2 = __shfl(y.x2, prev);
				bool keep = cand && (before == 0 || y.x2 != prev_x2);
				live = keep; e = y;
				if (cand_m == 0) end_call = true; else --i; // the first surviving match is always kept
			}
			if (end_call) {
				if (kind == TK_ROUND1 && gl == 0) push1 = chain_round1(rd, r, (int)(A.off[r + 1] - A.off[r]), ret, xp);
				active = false;
			}
		}
		wave_push<32>(O, push0 != FTASK_NONE, push0, A, aux0);
		wave_push<32>(O, push1 != FTASK_NONE, push1, A);
	}
}

// ------------------------------------------------------------------------------------------------------------------
// The backward sweep without the triangle ("window scheme").
//
// bwt.c:325-345 carries every LEP of the forward pass backward in lockstep: with ~16 LEPs that die after ~16 steps that
// is ~136 extensions per call, almost all of them spent on matches that never reach min_seed_len and are thrown away by
// the length filter (bwamem.c:232,246).  What the sweep reports can be stated per match END t (x < t <= ret):
//   let f(t) = the first position, going left from the pivot, at which [f, t) no longer has min_intv occurrences (or the
//   read start / an ambiguous base); f is monotone in t (a longer end dies no later);
//   the LEP ending at t is reported, as [f(t)+1, t), iff every longer LEP died strictly earlier, i.e. f(t) < f(t') for
//   the nearest longer end t' (the first-survivor rule of bwt.c:328-336; LEPs dropped by the equal-size rule of
//   bwt.c:337-340 have the same occurrences as a longer one, hence the same f, and are never reported either way).
// Only reports of at least min_seed_len bases are kept, and a bi-interval is a function of the string alone.  So an end
// t < x + min_seed_len matters only if the min_seed_len-mer [t - min_seed_len, t) occurs at all, and that is looked up
// directly: the jump table gives the bi-interval of its last jump_k bases, min_seed_len - jump_k backward extensions
// decide (a random 19-mer occurs in a 6 Gbp text with probability 2 %).  A lane that survives walks on alone to its f(t).
// Ends that are not looked at die inside their window, i.e. later than any reported shorter end, so the rule above can
// be evaluated over the lanes that did survive.  Ends t >= x + min_seed_len are the LEPs the forward pass stored (it
// stores no others under this scheme); each walks alone from the pivot.  ~16 table reads + ~40 extensions, 5 deep,
// replace 136 extensions, 17 deep; results identical.
//
// A group of G lanes per call: lanes 0..17 take the ends x+1 .. x+18, lanes 18.. take the stored LEPs in ascending order.
// in two halves: the filter (does the end's min_seed_len-mer occur at all? -> its jump-table code), and the table lookup
template <class WC>
__device__ __forceinline__ bool win_lane_filter(const SplitArgs &A, PackedReader &rd, uint32_t gl, int x, int ret, uint32_t &code, WC &W)
{
	const int k = A.min_seed_len, jk = A.jump_k; // jk <= k <= 24 (the window scheme's range)
	const int te = x + 1 + (int)gl;
	if ((int)gl >= k - 1 || te > ret || te - k < 0) return false;
	uint32_t badw;
	const uint64_t w = rd.window(te - k, k, badw); // the k-mer [te - k, te): its last jk bases are the jump-table code
	if (badw & ((1u << k) - 1u)) return false;     // an ambiguous base inside the window: this end cannot reach min_seed_len
	if (A.bloom) { // does the min_seed_len-mer [te - k, te) occur at all?
		wc_add(W, EV_BLOOM);
		if (!kmer_filter_has(A.bloom, A.bloom_bits, w & ((1ull << (2 * k)) - 1ull))) return false;
	}
	code = __brev((uint32_t)(w >> (2 * (k - jk))));                // group q at 2(15-q), its two bits swapped
	code = (((code & 0xAAAAAAAAu) >> 1) | ((code & 0x55555555u) << 1)) >> (32 - 2 * jk);
	return true;
}
template <class WC>
__device__ __forceinline__ bool win_lane_jump(const SplitArgs &A, uint32_t code, int te, uint32_t min_intv,
                                              Intv &e, uint32_t &pend, int &s, unsigned long long &my_q, unsigned long long &my_hits, WC &W)
{
	const int jk = A.jump_k;
	uint32_t dummy; unpack_lep(A.jump[code], e, dummy); wc_add(W, EV_JUMP);
	my_q += (unsigned)(jk - 1); my_hits += (unsigned)(jk - 1);
	if (e.x2 < min_intv) return false;
	pend = (uint32_t)te; s = te - jk - 1;
	return true;
}
template <class WC>
__device__ __forceinline__ bool win_lane_init(const SplitArgs &A, PackedReader &rd, uint32_t gl, int x, int ret, uint32_t min_intv,
                                              Intv &e, uint32_t &pend, int &s, unsigned long long &my_q, unsigned long long &my_hits, WC &W)
{
	uint32_t code;
	return win_lane_filter(A, rd, gl, x, ret, code, W) && win_lane_jump(A, code, x + 1 + (int)gl, min_intv, e, pend, s, my_q, my_hits, W);
}

template <int G, class WC>
__device__ __forceinline__ void bwd_win_run(const SplitArgs &A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctr, WaveOut &O,
                                            unsigned long long &my_q, unsigned long long &my_hits, WC &W)
{
	constexpr uint32_t MYCLS = G == 32 ? 4u : 5u;
	const DevIndex &ix = A.ix;
	const uint32_t lane = threadIdx.x & 63u, gl = lane % G, gbase = lane - gl;
	const uint64_t gmask = (G >= 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << gbase);
	bool active = false, walking = false, valid = false;
	uint32_t r = 0, kind = 0, min_intv = 1, pend = 0;
	int s = 0, f = 0, ret = 0, xp = 0, clk = 0;
	Intv e = {0, 0, 0};
	PackedReader rd;
	uint64_t batch_base = 0, avail_m = 0; bool exhausted = false;
	for (;;) { // task acquisition exactly as in bwd_groups_run
		uint64_t idle_m = __ballot(!active && gl == 0);
		if (idle_m != 0 && avail_m == 0 && !exhausted) {
			int src = __ffsll((long long)idle_m) - 1;
			unsigned long long base = 0;
			if ((int)lane == src) base = atomicAdd(ctr, 64ull);
			base = __shfl(base, src);
			if (base >= n_tasks) exhausted = true;
			else {
				uint64_t slot = base + lane;
				uint32_t cls = slot < n_tasks ? bq[slot].cls : 0xffffffffu;
				avail_m = __ballot(cls == MYCLS);
				batch_base = base;
			}
		}
		if (idle_m != 0 && avail_m != 0) {
			int kth = __popcll(idle_m & ((1ull << gbase) - 1ull));
			uint64_t m = avail_m;
			for (int q = 0; q < kth; ++q) m &= m - 1;
			bool mine = !active && m != 0;
			uint64_t t = batch_base + (uint64_t)(__ffsll((long long)m) - 1);
			int taken = __popcll(idle_m), have = __popcll(avail_m);
			if (taken > have) taken = have;
			for (int q = 0; q < taken; ++q) avail_m &= avail_m - 1;
			if (mine) {
				BTask bt = bq[t];
				r = bt.r; kind = bt.mi_kind >> 14; min_intv = bt.mi_kind & 0x3fffu; ret = bt.ret; xp = bt.x;
				const int n = bt.n;
				const uint64_t rb = A.off[r];
				f = 0x7fffffff;
				rd.start(A.seqp, rb, r, xp);
				if (gl < (uint32_t)WIN_LANES) valid = win_lane_init(A, rd, gl, xp, ret, min_intv, e, pend, s, my_q, my_hits, W);
				else {
					int j = (int)gl - WIN_LANES;
					valid = j < n;
					if (valid) { unpack_lep(A.lep[(size_t)t * A.lep_stride + j], e, pend); s = xp - 1; wc_add(W, EV_LEP); }
				}
				walking = valid;
				clk = xp + WIN_LANES - A.jump_k - 1; // the base in front of the last window lane, the first to join
				if (clk < xp - 1) clk = xp - 1;        // (the LEP lanes join at the pivot)
				active = true;
			}
		}
		if (exhausted && avail_m == 0 && __ballot(active) == 0) break; // wave-uniform exit
		uint64_t push0 = FTASK_NONE, push1 = FTASK_NONE, aux0 = AUX_NONE;
#ifdef CS_STEP_HIST
		W.steps((uint32_t)__popcll(__ballot(active && walking && s == clk)));
#endif
		if (active) {
			// One clock per group: position clk is the read base every walking lane prepends in this iteration.  A lane joins
			// when the clock reaches the base in front of its match (the window lanes start staggered, the LEPs at the pivot),
			// so all lanes that walk hold matches with the SAME start, and the equal-size rule of bwt.c:337-340 applies to
			// them as it stands: a match with as many occurrences as the next longer walking one has the same occurrences,
			// shares its fate from here on and is never reported -- it stops.  In repeats that is most lanes.
			const bool step = walking && s == clk;
			if (step) {
				uint32_t b = s < 0 ? 4u : rd.at(s);
				if (b > 3) { f = s; walking = false; }
				else {
					Intv y = extend1<true>(ix, e, (int)b, W); ++my_q;
					if (y.x2 < min_intv) { f = s; walking = false; } else { e = y; --s; }
				}
			}
			if ((clk & 3) == 0) { // (every fourth step is enough: a lane that could have stopped earlier only repeats a few extensions)
				const bool surv = step && walking;
				const uint64_t lm = __ballot(surv) & gmask;
				const uint64_t above = lane == 63 ? 0ull : lm & ~((2ull << lane) - 1ull);
				const int asrc = above ? __ffsll((long long)above) - 1 : (int)lane;
				const uint64_t ax2 = __shfl(e.x2, asrc);
				if (surv && above && ax2 == e.x2) { walking = false; valid = false; }
			}
			--clk;
			if ((__ballot(walking) & gmask) == 0) { // all ends of this call are settled: apply the first-survivor rule
				uint64_t vm = __ballot(valid) & gmask;
				uint64_t higher = lane == 63 ? 0ull : vm & ~((2ull << lane) - 1ull);
				int src = higher ? __ffsll((long long)higher) - 1 : (int)lane;
				int fn = __shfl(f, src);
				if (valid && (higher == 0 || f < fn)) push0 = emit_smem(A, r, kind, e, f + 1, pend, aux0);
				if (kind == TK_ROUND1 && gl == 0) push1 = chain_round1(rd, r, (int)(A.off[r + 1] - A.off[r]), ret, xp);
				active = false; valid = false;
			}
		}
		if (__ballot((push0 & push1) != FTASK_NONE)) { // (a call ends once in ~20 iterations: keep the dispenser code off the common path)
			wave_push<32>(O, push0 != FTASK_NONE, push0, A, aux0);
			wave_push<32>(O, push1 != FTASK_NONE, push1, A);
		}
	}
}

// Calls with more than 64 LEPs (tandem arrays, very long reads): one WAVE per call, the list stays in HBM and every step
// of the sweep streams the live part through the wave 64 entries at a time, longest first, compacting it in place (the
// write index never passes below the chunk being processed).  Same rules as above; the "previous surviving size" is
// carried from chunk to chunk.
template <class WC>
__device__ __forceinline__ void bwd_wide_run(const SplitArgs &A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctr, WaveOut &O,
                                             unsigned long long &my_q, WC &W)
{
	const DevIndex &ix = A.ix;
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	uint64_t batch_base = 0, avail_m = 0;
	for (;;) {
		if (avail_m == 0) { // wave-uniform acquisition, as in bwd_groups_run
			unsigned long long base = 0;
			if (lane == 0) base = atomicAdd(ctr, 64ull);
			base = __shfl(base, 0);
			if (base >= n_tasks) break;
			uint64_t slot = base + lane;
			uint32_t cls = slot < n_tasks ? bq[slot].cls : 0xffffffffu;
			avail_m = __ballot(cls == 3u);
			batch_base = base;
			if (avail_m == 0) continue;
		}
		uint64_t t = batch_base + (uint64_t)(__ffsll((long long)avail_m) - 1);
		avail_m &= avail_m - 1;
		BTask bt = bq[t];
		uint32_t r = bt.r, kind = bt.mi_kind >> 14, min_intv = bt.mi_kind & 0x3fffu;
		uint4 *lep = A.lep + (size_t)t * A.lep_stride;
		PackedReader rd; rd.start(A.seqp, A.off[r], r, (int)bt.x - 1);
		int n = bt.n, lo = 0, nm = 0, last_start = 0, f_long = 0x7fffffff;
		if (n <= 64) { // the whole list fits the wave: lane g holds LEP n-1-g in registers, nothing is streamed (same rules; no compaction:
			// the longest live match is the lowest live lane, the previous survivor the nearest surviving lane below)
			bool live = (int)lane < n;
			Intv p = {0, 0, 0}; uint32_t pend = 0;
			if (live) { unpack_lep(lep[n - 1 - (int)lane], p, pend); wc_add(W, EV_LEP); }
			for (int i = (int)bt.x - 1; i >= -1; --i) {
				const uint32_t b = i < 0 ? 4u : rd.at(i);
				Intv y = p;
				if (live && b <= 3) { y = extend1<true>(ix, p, (int)b, W); ++my_q; }
				const bool cand = live && b <= 3 && y.x2 >= min_intv;
				const uint64_t live_m = __ballot(live), cand_m = __ballot(cand);
				const int first = __ffsll((long long)live_m) - 1;
				uint64_t push0 = FTASK_NONE, aux0 = AUX_NONE;
				if (!((cand_m >> first) & 1ull) && (nm == 0 || i + 1 < last_start)) { // the longest live match stops here (bwt.c:328-336)
					if ((int)lane == first) push0 = emit_smem(A, r, kind, p, i + 1, pend, aux0);
					++nm; last_start = i + 1;
				}
				const uint64_t before = cand_m & lt_mask;
				const int prev = before ? 63 - __clzll((long long)before) : (int)lane;
				const uint64_t px2 = __shfl(y.x2, prev);
				live = cand && (!before || y.x2 != px2); // bwt.c:337-340
				if (live) p = y;
				wave_push<32>(O, push0 != FTASK_NONE, push0, A, aux0);
				f_long = i;
				if (__ballot(live) == 0) break;
			}
			n = 0; // (skips the streamed form below)
		}
		for (int i = (int)bt.x - 1; n > 0 && i >= -1; --i) {
			uint32_t b = i < 0 ? 4u : rd.at(i);
			int w = n; bool first_done = false, have_prev = false; uint64_t prev_carry = 0, push0 = FTASK_NONE, aux0 = AUX_NONE;
			for (int top = n; top > lo; top -= 64) {
				int j = top - 1 - (int)lane; bool valid = j >= lo;
				Intv p = {0, 0, 0}; uint32_t pend = 0;
				if (valid) { unpack_lep(lep[j], p, pend); wc_add(W, EV_LEP); }
				Intv y = p;
				if (valid && b <= 3) { y = extend1<true>(ix, p, (int)b, W); ++my_q; }
				bool cand = valid && b <= 3 && y.x2 >= min_intv;
				uint64_t cand_m = __ballot(cand);
				if (!first_done) { // lane 0 of the first chunk holds the longest live match (bwt.c:328-336)
					first_done = true;
					if (!(cand_m & 1ull) && (nm == 0 || i + 1 < last_start)) {
						if (lane == 0) push0 = emit_smem(A, r, kind, p, i + 1, pend, aux0);
						++nm; last_start = i + 1;
					}
				}
				uint64_t before = cand_m & lt_mask;
				int prev = before ? 63 - __clzll((long long)before) : (int)lane;
				uint64_t px2 = __shfl(y.x2, prev);
				if (!before) px2 = prev_carry;
				bool keep = cand && ((!before && !have_prev) || y.x2 != px2); // bwt.c:337-340
				uint64_t keep_m = __ballot(keep);
				if (keep) { lep[w - 1 - __popcll(keep_m & lt_mask)] = pack_lep(y, pend); wc_add(W, EV_LEP); }
				w -= __popcll(keep_m);
				if (cand_m) { have_prev = true; prev_carry = __shfl(y.x2, 63 - __clzll((long long)cand_m)); }
			}
			__threadfence_block(); // the compacted list is read back by other lanes of this wave in the next step
			wave_push<32>(O, push0 != FTASK_NONE, push0, A, aux0);
			f_long = i;            // the step at which the last stored LEP died, if this is the last step
			if (w == n) break;
			lo = w;
		}
		if (A.win) { // window scheme: the forward pass stored only the LEPs of min_seed_len bases or more; the short ends are
			// settled here by lanes 0..17 (bwd_win_run), the nearest longer end of the longest of them being the list above
			Intv e = {0, 0, 0}; uint32_t pend = 0; int s = 0, f = 0x7fffffff;
			unsigned long long hits = 0;
			bool valid = lane < (uint32_t)WIN_LANES && win_lane_init(A, rd, lane, (int)bt.x, (int)bt.ret, min_intv, e, pend, s, my_q, hits, W);
			bool walking = valid;
			while (__ballot(walking)) {
				if (walking) {
					uint32_t b = s < 0 ? 4u : rd.at(s);
					if (b > 3) { f = s; walking = false; }
					else {
						Intv y = extend1<true>(ix, e, (int)b, W); ++my_q;
						if (y.x2 < min_intv) { f = s; walking = false; } else { e = y; --s; }
					}
				}
			}
			uint64_t vm = __ballot(valid);
			uint64_t higher = vm & ~((2ull << lane) - 1ull);
			int src = higher ? __ffsll((long long)higher) - 1 : (int)lane;
			int fn = __shfl(f, src);
			if (!higher) fn = f_long;
			uint64_t pushw = FTASK_NONE, auxw = AUX_NONE;
			if (valid && f < fn) pushw = emit_smem(A, r, kind, e, f + 1, pend, auxw);
			wave_push<32>(O, pushw != FTASK_NONE, pushw, A, auxw);
		}
		uint64_t push1 = (kind == TK_ROUND1 && lane == 0) ? chain_round1(rd, r, (int)(A.off[r + 1] - A.off[r]), bt.ret, bt.x) : FTASK_NONE;
		wave_push<32>(O, push1 != FTASK_NONE, push1, A);
	}
}

// All cooperative backward work of one forward launch in ONE kernel: every wave works through the three size classes,
// starting with a different one depending on its workgroup, so all classes progress at once and a wave whose class runs
// dry moves on to the next instead of idling through that class's tail.  The three orders are written out (a loop over a
// class index costs 35 more VGPRs and one wave per SIMD).  ctrs[c] is the slot counter of class c.
template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, 5) void bwd_all_kernel(const SplitArgs A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctrs)
{
	if (*A.n_btasks == 0) return; // e.g. the first launch of a batch: every call sits at pivot 0 and needs no sweep
	WaveOut O = {0, 0};
	unsigned long long my_q = 0, my_hits = 0;
	__shared__ uint4 sst[SST_ENTRIES];
	sst_clear(sst);
	WaveCtrT<COUNT> W;
	const uint32_t role = blockIdx.x & 7u; // 5/8 of the workgroups start on the <=16 class, 2/8 on <=32, 1/8 on <=64
	if (role < 5) {
		bwd_groups_run<16>(A, bq, n_tasks, ctrs + 0, O, my_q, my_hits, sst, W);
		bwd_groups_run<32>(A, bq, n_tasks, ctrs + 1, O, my_q, my_hits, sst, W);
		bwd_groups_run<64>(A, bq, n_tasks, ctrs + 2, O, my_q, my_hits, sst, W);
	} else if (role < 7) {
		bwd_groups_run<32>(A, bq, n_tasks, ctrs + 1, O, my_q, my_hits, sst, W);
		bwd_groups_run<64>(A, bq, n_tasks, ctrs + 2, O, my_q, my_hits, sst, W);
		bwd_groups_run<16>(A, bq, n_tasks, ctrs + 0, O, my_q, my_hits, sst, W);
	} else {
		bwd_groups_run<64>(A, bq, n_tasks, ctrs + 2, O, my_q, my_hits, sst, W);
		bwd_groups_run<16>(A, bq, n_tasks, ctrs + 0, O, my_q, my_hits, sst, W);
		bwd_groups_run<32>(A, bq, n_tasks, ctrs + 1, O, my_q, my_hits, sst, W);
	}
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, my_q);
	if (my_hits) atomicAdd(A.n_sst_hits, my_hits);
	wc_flush(W, A.evc, KID_BWD_ALL);
}

// Calls without any stored LEP (the forward match is shorter than min_seed_len: typically the call at a mismatch, whose
// matches are all chance matches) are the bulk, and all their work is the 18 window lookups, of which 97 % end at the
// filter.  They get a kernel of their own that packs three calls into a wave (54 of 64 lanes busy) instead of one call per
// 32-lane group: a wave owns 64 consecutive slots, finds this class by ballot and works through it three at a time.  A round
// is the filter alone; the ends that pass it are PARKED in LDS (16 bytes: read, end, jump-table code, call key), and only
// when 64 of them have gathered does the wave look them up in the jump table, extend them to min_seed_len in lockstep,
// walk the survivors on to their ends -- all of that on full waves instead of one or two lanes out of 64 -- and apply the
// first-survivor rule per call (the lanes of a call found by their key).  The kernel is VALU-bound: that is the point.
// No dispenser and no atomics on the task side.
struct WinPark { uint32_t r, code; int32_t te; uint16_t mk, key; };             // an end that passed the filter; key: 64-slot batch (10 bits) | slot (6)
#ifndef CS_WIN_WAVES
#define CS_WIN_WAVES 6
#endif
#ifndef CS_WIN0_WAVES
#define CS_WIN0_WAVES 6
#endif
template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, CS_WIN0_WAVES) void bwd_win0_kernel(const SplitArgs A, const BTask *bq, uint64_t n_tasks)
{
	if (*A.n_btasks == 0) return;
	constexpr int PARK = 64;
	constexpr int32_t F_DEAD = (int32_t)0x80000000;
	__shared__ uint8_t rank2lane[BLOCK / 64][64];
	__shared__ WinPark park[BLOCK / 64][PARK];
	__shared__ int32_t park_f[BLOCK / 64][PARK];
	const DevIndex &ix = A.ix;
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const uint32_t seg = lane / WIN_LANES, gl = lane - seg * WIN_LANES;      // three segments of 18 lanes; lanes 54..63 idle
	const int kx = A.min_seed_len - A.jump_k;                                // extensions from the jump_k-mer to min_seed_len
	WaveOut O = {0, 0};
	WaveCtrT<COUNT> W;
	unsigned long long my_q = 0, my_hits = 0;
	int npark = 0; uint32_t park_seq0 = 0; // wave-uniform: parked ends, batch number of the oldest of them
	auto flush = [&]() {
		__builtin_amdgcn_wave_barrier();
		const bool mine = (int)lane < npark;
		Intv e = {0, 0, 0}; uint32_t pend = 0, r = 0, mk = 0, key = 0; int s = 0, f = 0x7fffffff, te = 0;
		PackedReader rd;
		bool alive = false;
		if (mine) {
			const WinPark p = park[wv][lane];
			r = p.r; mk = p.mk; key = p.key; te = p.te;
			alive = win_lane_jump(A, p.code, te, mk & 0x3fffu, e, pend, s, my_q, my_hits, W);
			if (alive) rd.start(A.seqp, A.off[r], r, s);
		}
		const uint32_t kind = mk >> 14, min_intv = mk & 0x3fffu;
		for (int st = 0; st < kx; ++st) { // wave-uniform: the jump_k-mer grows to min_seed_len bases, or the lane drops out
			if (alive) {
				const uint32_t b = rd.at(s);                                   // s >= 0: the window starts inside the read
				if (b > 3) alive = false;
				else {
					Intv y = extend1<true>(ix, e, (int)b, W); ++my_q;
					if (y.x2 < min_intv) alive = false; else { e = y; --s; }
				}
			}
		}
		bool walking = alive;
		while (__ballot(walking)) {
			if (walking) {
				uint32_t b = s < 0 ? 4u : rd.at(s);
				if (b > 3) { f = s; walking = false; }
				else {
					Intv y = extend1<true>(ix, e, (int)b, W); ++my_q;
					if (y.x2 < min_intv) { f = s; walking = false; } else { e = y; --s; }
				}
			}
		}
		if (mine) park_f[wv][lane] = alive ? f : F_DEAD;
		__builtin_amdgcn_wave_barrier();
		bool emit = alive;
		if (alive) { // the nearest longer match of the same call among the parked ones that reached min_seed_len
			int best_te = 0x7fffffff, best_f = 0;
			for (int q = 0; q < npark; ++q) {
				const int tq = park[wv][q].te, fq2 = park_f[wv][q];
				if (park[wv][q].key == key && fq2 != F_DEAD && tq > te && tq < best_te) { best_te = tq; best_f = fq2; }
			}
			emit = best_te == 0x7fffffff || f < best_f;
		}
		uint64_t push0 = FTASK_NONE, aux0 = AUX_NONE;
		if (emit) push0 = emit_smem(A, r, kind, e, f + 1, pend, aux0);
		wave_push<64>(O, push0 != FTASK_NONE, push0, A, aux0);
		__builtin_amdgcn_wave_barrier();
		npark = 0;
	};
	const uint64_t n_batches = (n_tasks + 63) / 64, wstride = (uint64_t)gridDim.x * (BLOCK / 64);
	const int k = A.min_seed_len, jk = A.jump_k;
	uint32_t bseq = 0;
	for (uint64_t bch = (uint64_t)blockIdx.x * (BLOCK / 64) + wv; bch < n_batches; bch += wstride, ++bseq) { // wave-uniform
		const uint64_t slot = bch * 64 + lane;
		BTask bt = {0, 0, 0, 0, 0, 0xffffffffu};
		if (slot < n_tasks) bt = bq[slot];
		const bool is = bt.cls == 6u;
		const uint64_t m = __ballot(is);
		const int cnt = __popcll(m);
		if (cnt == 0) continue;
		if (npark && bseq - park_seq0 >= 1000u) flush(); // (keys carry ten bits of the batch number)
		// Per slot, all calls of the batch at once: the bases every window of the call can touch, [x + 1 - k, x + 18), as one
		// 96-bit string + ambiguity bits (three records at most), and the call's successor.  The rounds below then need no
		// memory access but the filter word itself.
		uint64_t cw0 = 0, cbad = ~0ull; uint32_t cw1 = 0; uint64_t push1 = FTASK_NONE;
		if (is) {
			const uint64_t rb = A.off[bt.r]; const int len = (int)(A.off[bt.r + 1] - rb);
			PackedReader rd; rd.rec = A.seqp + (rb >> 5) + bt.r;
			const int c0 = (int)bt.x + 1 - k, start = c0 < 0 ? 0 : c0;
			const uint32_t sh = (uint32_t)(start - c0), j0 = (uint32_t)start & 31u;
			const int i0 = start >> 5, imax = len >> 5;           // (record imax exists and ends the read)
			const uint4 none = {0u, 0u, ~0u, 0u};
			const uint4 q0 = rd.rec[i0], q1 = i0 + 1 <= imax ? rd.rec[i0 + 1] : none, q2 = i0 + 2 <= imax ? rd.rec[i0 + 2] : none;
			const uint64_t b0 = (uint64_t)q0.x | (uint64_t)q0.y << 32, b1 = (uint64_t)q1.x | (uint64_t)q1.y << 32, b2 = (uint64_t)q2.x | (uint64_t)q2.y << 32;
			uint64_t wa = b0 >> (j0 << 1), wb = b1 >> (j0 << 1);      // bases start .. start+31 and start+32 .. start+63
			uint64_t bad = ((uint64_t)q1.z << 32 | q0.z) >> j0;         // their ambiguity bits (64 - j0 of them; the rest from q2)
			if (j0) { wa |= b1 << ((32u - j0) << 1); wb |= b2 << ((32u - j0) << 1); bad |= (uint64_t)q2.z << (64u - j0); }
			if (sh) { wb = wb << (sh << 1) | wa >> (64u - (sh << 1)); wa <<= sh << 1; bad = bad << sh | ((1ull << sh) - 1ull); } // the read starts inside the range
			cw0 = wa; cw1 = (uint32_t)wb; cbad = bad;
			if ((bt.mi_kind >> 14) == TK_ROUND1) { rd.wk = i0; rd.bases = b0; rd.bad = q0.z; push1 = chain_round1(rd, bt.r, len, (int)bt.ret, (int)bt.x); }
			rank2lane[wv][__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
		}
		wave_push<64>(O, push1 != FTASK_NONE, push1, A);
		__builtin_amdgcn_wave_barrier();
		for (int r0 = 0; r0 < cnt; r0 += 3) { // wave-uniform
			const int rank = r0 + (int)seg;
			const bool job = seg < 3 && rank < cnt;
			const int src = job ? (int)rank2lane[wv][rank] : (int)lane;
			const uint32_t r = __shfl(bt.r, src), mk = __shfl((uint32_t)bt.mi_kind, src);
			const int x = __shfl((int)bt.x, src), ret = __shfl((int)bt.ret, src);
			const uint64_t c0w = __shfl(cw0, src), cb = __shfl(cbad, src); const uint32_t c1w = __shfl(cw1, src);
			const int te = x + 1 + (int)gl;
			bool pass = job && (int)gl < k - 1 && te <= ret && te - k >= 0;
			uint32_t code = 0;
			if (pass) {
				const uint32_t g2 = gl << 1;
				const uint64_t w = gl ? c0w >> g2 | (uint64_t)c1w << (64u - g2) : c0w; // the k-mer [te - k, te)
				pass = ((uint32_t)(cb >> gl) & ((1u << k) - 1u)) == 0;              // no ambiguous base in it
				if (pass && A.bloom) { wc_add(W, EV_BLOOM); pass = kmer_filter_has(A.bloom, A.bloom_bits, w & ((1ull << (2 * k)) - 1ull)); }
				code = __brev((uint32_t)(w >> (2 * (k - jk))));
				code = (((code & 0xAAAAAAAAu) >> 1) | ((code & 0x55555555u) << 1)) >> (32 - 2 * jk);
			}
			const uint64_t am = __ballot(pass);
			if (npark + __popcll(am) > PARK) flush();
			if (npark == 0) park_seq0 = bseq;
			if (pass) {
				WinPark p; p.r = r; p.code = code; p.te = te; p.mk = (uint16_t)mk; p.key = (uint16_t)((bseq & 0x3ffu) << 6 | (uint32_t)src);
				park[wv][npark + __popcll(am & ((1ull << lane) - 1ull))] = p;
			}
			npark += __popcll(am);
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (npark) flush();
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, my_q);
	if (my_hits) atomicAdd(A.n_sst_hits, my_hits);
	wc_flush(W, A.evc, KID_BWD_WIN0);
}

// window scheme: ctrs[0] / ctrs[1] are the slot counters of the classes with up to 14 / 46 stored LEPs
template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, CS_WIN_WAVES) void bwd_win_kernel(const SplitArgs A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctrs)
{
	if (*A.n_btasks == 0) return;
	WaveOut O = {0, 0};
	WaveCtrT<COUNT> W;
	unsigned long long my_q = 0, my_hits = 0;
	if ((blockIdx.x & 7u) != 7u) {
		bwd_win_run<32>(A, bq, n_tasks, ctrs + 0, O, my_q, my_hits, W);
		bwd_win_run<64>(A, bq, n_tasks, ctrs + 1, O, my_q, my_hits, W);
	} else {
		bwd_win_run<64>(A, bq, n_tasks, ctrs + 1, O, my_q, my_hits, W);
		bwd_win_run<32>(A, bq, n_tasks, ctrs + 0, O, my_q, my_hits, W);
	}
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, my_q);
	if (my_hits) atomicAdd(A.n_sst_hits, my_hits);
	wc_flush(W, A.evc, KID_BWD_WIN);
}

// the calls with more than 64 LEPs, one wave each; rare, so it runs beside bwd_all_kernel on its own stream
#ifndef CS_WIDE_BLOCKS
#define CS_WIDE_BLOCKS 6
#endif
template <bool COUNT>
__global__ __launch_bounds__(256, CS_WIDE_BLOCKS) void bwd_wide_kernel(const SplitArgs A, const BTask *bq, uint64_t n_tasks, unsigned long long *ctr)
{
	if (*A.n_btasks == 0) return;
	WaveOut O = {0, 0};
	WaveCtrT<COUNT> W;
	unsigned long long my_q = 0;
	bwd_wide_run(A, bq, n_tasks, ctr, O, my_q, W);
	wave_push_finish(O, A);
	atomicAdd(A.n_queries, my_q);
	wc_flush(W, A.evc, KID_BWD_WIDE);
}

// ------------------------------------------------------------------------------------------------------------------
// Re-seeding from the text.
//
// A round-1 SMEM [beg, end) with a single occurrence triggers bwt_smem1a(pivot = (beg+end)/2, min_intv = 2)
// (bwamem.c:241-249): all maximal substrings through the pivot that occur at least twice.  On the FM index that is a
// forward pass plus a triangular backward sweep, ~150 extensions, and for most reads it finds nothing of min_seed_len.
// But inside [beg, end) the read IS the text at the SMEM's position P = SA[x0], and "occurs at least twice" is a property
// of the text alone: the substring of length l at text position p is repeated iff l <= rep[p] (fm_device.hpp).  So with
// e(q) = q + rep[q] the sweep of bwt.c:303-345 reads off directly:
//   * forward pass from the pivot p: longest match with >= 2 occurrences ends at e(p);
//   * backward step to start q: the longest surviving end is e(q) (never larger than e(q+1));
//   * [q, e(q)) is reported when it does not survive the next step, e(q-1) < e(q), i.e. rep[q-1] <= rep[q];
//   * the sweep is over when e(q) <= p.
// The bi-interval of a reported substring comes from the inverse suffix array and a short walk over lcp[] to the ends of
// its suffix-array interval (forward strand and reverse-complement strand).
// This only holds while the substrings stay inside [beg, end), where read and text agree: if a candidate reaches either
// end of the SMEM, a capped value (255) turns up, or an interval walk gets long, nothing is emitted and the call stays
// in the queue for fwd_kernel / bwd_all_kernel.  So the result is the reference's either way; only the cost differs.
// per-lane event counts of the text-side kernels, handed to the kernel's WaveCtr at the end
struct LaneCtr { uint32_t sa, isa, rep, lcp, mem; };
__device__ __forceinline__ void lc_flush(LaneCtr c, WaveCtr &W)
{
	W.addn(EV_SA, c.sa); W.addn(EV_ISA, c.isa); W.addn(EV_REP, c.rep); W.addn(EV_LCP, c.lcp); W.addn(EV_MEM, c.mem);
}
struct RepReader { // rep[] / lcp[] bytes around a moving position, one aligned 8-byte load per 8 positions (the arrays are padded)
	const uint8_t *base; uint64_t wk, w; uint32_t loads;
	__device__ __forceinline__ uint32_t at(uint64_t pos)
	{
		uint64_t k = pos >> 3;
		if (k != wk) { wk = k; w = *reinterpret_cast<const uint64_t *>(base + (k << 3)); ++loads; }
		return (uint32_t)(w >> ((pos & 7) << 3)) & 0xffu;
	}
};
#ifndef CS_LCP_BYTES
#define CS_LCP_BYTES 0
#endif
#if CS_LCP_BYTES
struct LcpReader { // (A/B variant: one byte per load, as before)
	static constexpr uint32_t BYTES = 1;
	const uint8_t *base; uint64_t wk, w; uint32_t loads;
	__device__ __forceinline__ uint32_t at(uint64_t pos) { ++loads; return base[pos]; }
};
#else
struct LcpReader : RepReader { static constexpr uint32_t BYTES = 8; };
#endif
// bi-interval of the repeated substring of length v at text position pos (v <= 254, so the capped lcp[] decides exactly)
__device__ __forceinline__ bool text_interval(const DevIndex &ix, uint64_t pos, uint32_t v, Intv &out, LaneCtr &C, int MAX_WALK = 48)
{
	if (v == 0 || pos + v > ix.seq_len) return false; // (cannot happen for a substring of a mem; a walk must never leave the arrays)
	uint64_t lo = isa_direct(ix, pos), hi = lo, lo2 = isa_direct(ix, ix.seq_len - (pos + v));
	int steps = 0;
	C.isa += 2;
	LcpReader Lr = {ix.lcp, ~0ull, 0, 0}; // a walk is a chain of dependent loads: eight rows per load instead of one
	struct Tally { LcpReader &R; LaneCtr &C; __device__ ~Tally() { C.lcp += LcpReader::BYTES * R.loads; } } tally = {Lr, C};
	while (lo > 0 && Lr.at(lo) >= v) { --lo; if (++steps > MAX_WALK) return false; }
	while (hi < ix.seq_len && Lr.at(hi + 1) >= v) { ++hi; if (++steps > MAX_WALK) return false; }
	while (lo2 > 0 && Lr.at(lo2) >= v) { --lo2; if (++steps > 2 * MAX_WALK) return false; }
	out.x0 = lo; out.x1 = lo2; out.x2 = hi - lo + 1;
	return true;
}
// (r2_quick_rep, further up, is the test on rep[] itself: fwd0_kernel asks it too.)  0: answered; 1: needs the sweep; 2: the text cannot tell.
__device__ __forceinline__ int r2_quick(const SplitArgs &A, uint64_t x0, int beg, int end, int pivot, uint64_t &P, LaneCtr &C)
{
	const DevIndex &ix = A.ix;
	const int len = end - beg, po = pivot - beg, k = A.min_seed_len;
	if (k < 2 || po > 4096) return 2;
	P = sa_direct(ix, x0);
	++C.sa;
	if (P >= ix.seq_len || P + (uint64_t)len > ix.seq_len) return 2; // (an SMEM lies inside the text)
	uint32_t nw = 0;
	const int q = r2_quick_rep(ix, P, len, po, k, nw);
	C.rep += nw;
	return q;
}
__device__ __forceinline__ bool r2_by_text(const SplitArgs &A, uint32_t r, uint64_t P, int beg, int end, int pivot, LaneCtr &C)
{
	const DevIndex &ix = A.ix;
	const int len = end - beg, po = pivot - beg, k = A.min_seed_len;
	// Where the SMEM touches an end of the READ the sweep cannot run past it either (bwt.c:303 stops the forward pass at
	// the last base, bwt.c:326 the backward sweep in front of the first), so there the text still tells everything: ends
	// are clipped to the read end, and a match that is still alive at the first base is reported there.
	const bool at_start = beg == 0, at_end = (uint64_t)end == A.off[r + 1] - A.off[r];
	// rep[] of the SMEM's bases, eight per aligned load; the window covers the offsets [wo, wo + 8) of the SMEM (32-bit arithmetic:
	// the walk is this kernel's inner loop, and a wave runs as long as its longest walk)
	const uint8_t *rp = ix.rep + P;
	int wo = po - (int)((P + (uint64_t)po) & 7ull);
	uint64_t w = *reinterpret_cast<const uint64_t *>(rp + wo);
	uint32_t loads = 1;
	struct Tally { uint32_t &n; LaneCtr &C; __device__ ~Tally() { C.rep += n; } } tally = {loads, C};
	constexpr int MAXC = 8;               // reported substrings per call; more (tandem arrays): leave it to the index
	int co[MAXC], cv[MAXC], ne = 0;
	auto eff = [&](int o, int &v) -> bool { // repeat length at offset o (inside the window) as far as it matters; false: the text cannot tell
		v = (int)((uint32_t)(w >> ((uint32_t)(o - wo) << 3)) & 0xffu);
		if (v == 0) return false;
		if (o + v >= len) { if (!at_end) return false; v = len - o; return true; } // (a capped 255 that reaches the end is as good as the true value)
		return v != 255;
	};
	int o = po, v = 0;
	if (!eff(o, v)) return false;
	for (;;) {
		if (o + v <= po) break;                               // no longer through the pivot: the sweep is over
		if (o == 0) {                                         // alive at the SMEM's first base
			if (!at_start) return false;                      // ... which is not the read's: the match may extend beyond it
			if (v >= k) { if (ne == MAXC) return false; co[ne] = 0; cv[ne] = v; ++ne; }
			break;
		}
		if (o == wo) { wo -= 8; w = *reinterpret_cast<const uint64_t *>(rp + wo); ++loads; } // (P + wo >= 0: an aligned address below P + o)
		int vp = 0;
		if (!eff(o - 1, vp)) return false;
		if (vp <= v && v >= k) { if (ne == MAXC) return false; co[ne] = o; cv[ne] = v; ++ne; }
		--o; v = vp;
	}
	Intv ci[MAXC];
	for (int j = 0; j < ne; ++j) if (!text_interval(ix, P + (uint64_t)co[j], (uint32_t)cv[j], ci[j], C, 192)) return false;
	for (int j = 0; j < ne; ++j) emit_mem(A, r, ci[j], (uint32_t)(beg + co[j]), (uint32_t)(beg + co[j] + cv[j]));
	return true;
}
// One lane per slot of the next forward queue.  Calls that are answered drop out; everything else is copied, without the
// no-op padding, to `fq_out` (the queue the finished iteration has consumed), so the next launches see a dense queue.
#ifndef CS_R2_WAVES
#define CS_R2_WAVES 6
#endif
__global__ __launch_bounds__(256, CS_R2_WAVES) void r2text_kernel(const SplitArgs A, const uint64_t *fq, const uint64_t *aux, const unsigned long long *n_ptr,
                                                     unsigned long long *n_done, unsigned long long *n_left, uint64_t *fq_out, unsigned long long *n_out)
{
	uint64_t n = *n_ptr; if (n > A.fq_cap) n = A.fq_cap;
	const uint32_t lane = threadIdx.x & 63u;
	unsigned long long done = 0, left = 0;
	LaneCtr C = {0, 0, 0, 0, 0};
	// a wave takes 256 consecutive slots at a time (one atomic on the output counter per 256 slots).  The quick test settles
	// most candidates; the others are gathered (LDS) and swept together on full waves afterwards -- the sweep is a loop of up
	// to a few dozen steps, and a wave runs as long as its longest.
	__shared__ uint64_t slow_P[256 / 64][256];
	__shared__ uint8_t slow_src[256 / 64][256], slow_ok[256 / 64][256];
	const uint32_t wv = threadIdx.x >> 6;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t t0 = wave * 256; t0 < n; t0 += n_waves * 256) { // wave-uniform
		uint64_t task[4]; uint64_t km[4]; uint32_t total = 0, nslow = 0, slowm = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint64_t t = t0 + (uint64_t)(64 * j) + lane;
			task[j] = t < n ? fq[t] : FTASK_NONE;
			bool slow = false; uint64_t P = 0;
			if ((uint32_t)(task[j] >> 62) == TK_ROUND2 && ((task[j] >> 48) & 0x3fffu) == 2u) {
				const uint64_t a = aux[t];
				const int pivot = (int)((task[j] >> 32) & 0xffffu);
				const int beg = (int)((a >> 37) & 0xffffu), end = 2 * pivot + (int)((a >> 53) & 1u) - beg;
				const int q = r2_quick(A, a & ((1ull << 37) - 1ull), beg, end, pivot, P, C);
				if (q == 0) { task[j] = FTASK_NONE; ++done; } else if (q == 1) slow = true; else ++left;
			}
			const uint64_t sm = __ballot(slow);
			if (slow) { const uint32_t i = nslow + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull)); slow_P[wv][i] = P; slow_src[wv][i] = (uint8_t)(64 * j + (int)lane); slowm |= 1u << j; }
			nslow += (uint32_t)__popcll(sm);
		}
		__builtin_amdgcn_wave_barrier();
		for (uint32_t c = 0; c < nslow; c += 64) { // wave-uniform
			const uint32_t i = c + lane;
			if (i < nslow) {
				const uint32_t src = slow_src[wv][i];
				const uint64_t t = t0 + src, tk = fq[t], a = aux[t];
				const int pivot = (int)((tk >> 32) & 0xffffu);
				const int beg = (int)((a >> 37) & 0xffffu), end = 2 * pivot + (int)((a >> 53) & 1u) - beg;
				slow_ok[wv][src] = r2_by_text(A, (uint32_t)tk, slow_P[wv][i], beg, end, pivot, C) ? 1 : 0;
			}
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (slowm & (1u << j)) { if (slow_ok[wv][64 * j + (int)lane]) { task[j] = FTASK_NONE; ++done; } else ++left; }
			km[j] = __ballot(task[j] != FTASK_NONE);
			total += (uint32_t)__popcll(km[j]);
		}
		__builtin_amdgcn_wave_barrier();
		if (total) {
			unsigned long long base = 0;
			if (lane == 0) base = atomicAdd(n_out, (unsigned long long)total);
			base = __shfl(base, 0);
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (task[j] != FTASK_NONE) fq_out[base + (uint64_t)__popcll(km[j] & ((1ull << lane) - 1ull))] = task[j];
				base += (uint64_t)__popcll(km[j]);
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) { done += __shfl_xor(done, o); left += __shfl_xor(left, o); }
	if (lane == 0) { if (done) atomicAdd(n_done, done); if (left) atomicAdd(n_left, left); }
	WaveCtr W; // (these two kernels have registers to spare: they always count)
	lc_flush(C, W);
	wc_flush(W, A.evc, KID_R2TEXT);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3 (bwt_seed_strategy1, bwt.c:357-381; the loop of bwamem.c:253-262) after rounds 1 and 2, one lane per read.
//
// A round-3 seed starting at x is the shortest prefix [x, x+L), L >= min_seed_len + 1, with fewer than max_mem_intv
// occurrences.  Where the read lies inside one of its own mems of rounds 1 and 2, [beg, end) -- an exact match of the text at
// P = SA[x0], whichever of its occurrences that row is -- the read IS the text, and for L = min_seed_len + 1 the answer is in the text arrays: rep[p] < L means the L-mer at p is
// unique (bi-interval = two inverse-suffix-array reads), otherwise a short walk over lcp[] counts its occurrences.  Only
// where that does not apply (the seed would leave the SMEM, 20 or more occurrences, repeats without a unique SMEM) the
// seed is computed on the FM index as before (jump table + extensions).  Same seeds, a fraction of the index reads:
// on the bench workload round 3 was the largest single consumer of HBM traffic.
// Length of the round-3 seed that starts at text position p: the smallest L >= k1 for which the L-mer at p has fewer than
// max_intv occurrences (bwt.c:370).  The suffixes that share a prefix with suffix p sit around row ISA[p]; going outwards,
// the running minimum of lcp[] on each side is the length shared with the j-th neighbour, non-increasing.  The L-mer has
// 1 + #{neighbours sharing >= L} occurrences, so L = 1 + the (max_intv - 1)-th largest shared length (or k1 if fewer than
// that many neighbours share k1 bases).  At most max_intv - 1 bytes of lcp[] on either side: two cache lines instead of the
// dozens of bwt_extend calls such a seed costs in a repeat.  False when a capped value (255) would decide.
// 0: the arrays cannot tell; 1: L; 2: max_intv - 1 neighbours share 255 bases or more (every prefix of up to 254 bases has max_intv occurrences)
__device__ __forceinline__ int r3_text_len(const DevIndex &ix, uint64_t p, int k1, uint32_t max_intv, int &L, LaneCtr &C)
{
	if (max_intv < 2 || max_intv > 41) return 0;
	const uint32_t m = max_intv - 1;
	if (p >= ix.seq_len) return 0;
	uint64_t up = isa_direct(ix, p), dn = up + 1;
	LcpReader Lu = {ix.lcp, ~0ull, 0, 0}, Ld = {ix.lcp, ~0ull, 0, 0}; // one window per side: ~3 dependent loads instead of up to 19
	struct Tally { LcpReader &A, &B; LaneCtr &C; __device__ ~Tally() { C.lcp += LcpReader::BYTES * (A.loads + B.loads); } } tally = {Lu, Ld, C};
	uint32_t mu = Lu.at(up), md = Ld.at(dn), val = 0;
	++C.isa;
	for (uint32_t t = 0; t < m; ++t) {
		val = mu > md ? mu : md;
		if (val < (uint32_t)k1) { L = k1; return 1; }      // fewer than max_intv occurrences already at k1 bases
		if (mu >= md) { if (up == 0) return 0; --up; const uint32_t c = Lu.at(up); mu = c < mu ? c : mu; }
		else { if (dn > ix.seq_len) return 0; ++dn; const uint32_t c = Ld.at(dn); md = c < md ? c : md; }
	}
	if (val >= 255u) return 2;                              // the true shared length is not known
	L = (int)val + 1;
	return 1;
}

// cnt_snap: the per-read mem counts at a moment when every entry below them was complete (a copy taken between launches):
// the kernel may run beside the last, thin iterations of rounds 1/2, which keep appending to the same lists.  A read that is
// still being worked on simply finds fewer covering mems and takes more of its seeds from the index.
#ifndef CS_R3_WAVES
#define CS_R3_WAVES 5
#endif
// reads that still have calls of rounds 1/2 in the queue when r3text_kernel starts: their mem lists are not final
__global__ void mark_pending_kernel(const uint64_t *fq, const unsigned long long *n_ptr, uint64_t cap, int64_t n_reads, uint8_t *pending)
{
	uint64_t n = *n_ptr; if (n > cap) n = cap;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t task = fq[t];
		if ((uint32_t)(task >> 62) != TK_NOP && (int64_t)(uint32_t)task < n_reads) pending[(uint32_t)task] = 1;
	}
}
// `pending[r] == 0` and no more than `cap` mems: every SMEM of the read is in its list.  Then the text answers everything: the mem that
// covers [x, x + k1) and reaches furthest to the right ends where the longest match from x ends (a longer one would sit in an SMEM of
// its own, which would be in the list), and if no mem covers it the k1-mer does not occur at all.
__global__ __launch_bounds__(256, CS_R3_WAVES) void r3text_kernel(const SplitArgs A, const uint32_t *cnt_snap, unsigned long long *n_text_seeds, const uint8_t *pending)
{
	const DevIndex &ix = A.ix;
	const int k1 = A.min_seed_len + 1;
	const int jk = (A.jump && A.jump_k <= A.min_seed_len) ? A.jump_k : 0;
	unsigned long long my_q = 0, my_hits = 0, my_text = 0;
	LaneCtr C = {0, 0, 0, 0, 0};
	WaveCtr W;
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t rb = A.off[r]; const int len = (int)(A.off[r + 1] - rb);
		const uint32_t cs = cnt_snap[r], nm0 = cs < A.cap ? cs : A.cap; // the mems of rounds 1 and 2 known to be complete
		const bool complete = pending[r] == 0 && cs <= A.cap;
		const OutMem *mine = A.out + (size_t)r * A.cap;
		int cb = 0, ce = 0; uint64_t cp = 0; // the mem the cursor is in: [cb, ce) at text position cp
		bool covered = false;
		int x = 0;
		PackedReader rd; rd.start(A.seqp, rb, (uint32_t)r, 0);
		while (x < len) {
			if (rd.at(x) > 3) { ++x; continue; }
			if (!(x >= cb && x + k1 <= ce)) { // look for a mem that covers [x, x + k1)
				cb = ce = 0;
				int best = -1;
				for (uint32_t a = 0; a < nm0; ++a) { // the one that reaches furthest (any occurrence of it will do)
					const uint64_t info = mine[a].info; const int mb = (int)(info >> 32), me = (int)(uint32_t)info;
					++C.mem;
					if (x >= mb && x + k1 <= me && me > ce) { best = (int)a; cb = mb; ce = me; }
				}
				covered = best >= 0;
				if (covered) {
					cp = sa_direct(ix, mine[best].x0); ++C.sa;
					if (cp >= ix.seq_len || cp + (uint64_t)(ce - cb) > ix.seq_len) cb = ce = 0; // (a mem lies inside the text)
				}
			}
			if (complete && !covered) { // [x, x + k1) does not occur (or is cut short by an ambiguous base or the read's end): bwt.c:366-377
				uint32_t badw; const int nb = len - x < k1 ? len - x : k1;   // walks on to x + k1, the ambiguous base or the end, reports nothing
				(void)rd.window(x, nb, badw);
				const int fb = badw ? __ffs((int)badw) - 1 : 32;
				int nx = fb < k1 ? x + fb + 1 : x + k1;
				if (nx > len) nx = len;
				my_q += (unsigned)(nx - x - 1); my_hits += (unsigned)(nx - x - 1);
				x = nx;
				continue;
			}
			if (x >= cb && x + k1 <= ce && ix.rep) {
				const uint64_t p = cp + (uint64_t)(x - cb);
				// Inside a mem the next seeds start k1 apart as long as each k1-mer is unique, so up to four of them are resolved
				// at once: four rep[] bytes, then eight independent inverse-SA reads, one counter update for the four mems.
				{
					constexpr int SPEC = 4;
					int ns = (ce - x) / k1; if (ns > SPEC) ns = SPEC;
					uint32_t vj[SPEC];
#pragma unroll
					for (int j = 0; j < SPEC; ++j) vj[j] = j < ns ? (uint32_t)ix.rep[p + (uint64_t)(j * k1)] : 255u;
					C.rep += (uint32_t)ns; // (single bytes, one line apiece: counted like the 8-byte loads of r2text_kernel)
					int nu = 0; // leading unique k1-mers
#pragma unroll
					for (int j = 0; j < SPEC; ++j) if (nu == j && vj[j] < (uint32_t)k1) nu = j + 1;
					if (nu > 0) {
						uint64_t a0[SPEC], a1[SPEC];
#pragma unroll
						for (int j = 0; j < SPEC; ++j) {
							const uint64_t pj = p + (uint64_t)(j * k1);
							a0[j] = j < nu ? isa_direct(ix, pj) : 0; a1[j] = j < nu ? isa_direct(ix, ix.seq_len - (pj + (uint64_t)k1)) : 0;
						}
						C.isa += 2u * (uint32_t)nu;
						const uint32_t k0 = atomicAdd(&A.out_cnt[r], (uint32_t)nu);
#pragma unroll
						for (int j = 0; j < SPEC; ++j) {
							if (j < nu) {
								OutMem m = {a0[j], a1[j], 1, (uint64_t)(uint32_t)(x + j * k1) << 32 | (uint32_t)(x + (j + 1) * k1)};
								const uint32_t kk = k0 + (uint32_t)j;
								if (kk < A.cap) A.out[(size_t)r * A.cap + kk] = m;
								else {
									unsigned long long sl = atomicAdd(A.ovf_cnt, 1ull);
									if (sl < A.ovf_cap) { OvfRec o = {m, (uint32_t)r, 0}; A.ovf[sl] = o; } else atomicMax(A.err, 1ull);
								}
							}
						}
						my_q += (unsigned)(nu * (k1 - 1)); my_hits += (unsigned)(nu * (k1 - 1)); my_text += (unsigned)nu;
						x += nu * k1;
						continue;
					}
				}
				const uint32_t v = ix.rep[p];
				Intv iv = {0, 0, 0}; bool ok = false;
				++C.rep;
				if (v < (uint32_t)k1) { iv.x0 = isa_direct(ix, p); iv.x1 = isa_direct(ix, ix.seq_len - (p + (uint64_t)k1)); iv.x2 = 1; ok = true; C.isa += 2; }
				int L = k1;
				if (!ok && k1 < 255) {
					const int st = r3_text_len(ix, p, k1, (uint32_t)(A.max_mem_intv > 0xffffffffull ? 0xffffffffull : A.max_mem_intv), L, C);
					if (st == 1 && x + L <= ce) ok = text_interval(ix, p, (uint32_t)L, iv, C) && iv.x2 < A.max_mem_intv;
					else if (ce == len && ((st == 1 && x + L > ce) || (st == 2 && ce - x <= 254))) {
						// Every prefix of [x, len) has max_mem_intv occurrences or more (the mem reaches the read's end, so the read is the
						// text all the way): bwt.c:366-377 walks to the end without reporting -- reads from tandem arrays and young
						// duplications, each such walk a chain of a hundred extensions.  Round 3 is over for this read.
						my_q += (unsigned)(len - x - 1); my_hits += (unsigned)(len - x - 1);
						x = len;
						continue;
					} else if (complete && ((st == 1 && x + L > ce) || (st == 2 && ce - x <= 254))) {
						// ... and where the mem ends inside the read, the longest match from x ends with it -- if no other mem covers x
						// and reaches further (this one was picked for an earlier x): one base more and nothing is left, which is below
						// max_mem_intv but reports nothing either (bwt.c:370-371)
						int best = -1, bmb = cb, bme = ce;
						for (uint32_t a = 0; a < nm0; ++a) {
							const uint64_t info = mine[a].info; const int mb = (int)(info >> 32), me = (int)(uint32_t)info;
							++C.mem;
							if (x >= mb && x + k1 <= me && me > bme) { best = (int)a; bmb = mb; bme = me; }
						}
						if (best < 0) {
							my_q += (unsigned)(ce - x); my_hits += (unsigned)(ce - x);
							x = ce + 1;
							continue;
						}
						cb = bmb; ce = bme; cp = sa_direct(ix, mine[best].x0); ++C.sa;
						if (!(cp >= ix.seq_len || cp + (uint64_t)(ce - cb) > ix.seq_len)) continue; // the same question again, inside that mem
						cb = ce = 0;                                                                // (cannot happen; then the index answers)
					}
				}
				if (ok) {
					emit_mem(A, (uint32_t)r, iv, (uint32_t)x, (uint32_t)(x + L));
					my_q += (unsigned)(L - 1); my_hits += (unsigned)(L - 1); ++my_text;
					x += L;
					continue;
				}
			}
			// bwt_seed_strategy1 on the index -- unless the min_seed_len-mer at x does not occur at all (the filter of the window
			// scheme; typically a seed across a mismatch): then the interval runs empty before the seed may end, the reference walks on
			// to min_seed_len + 1 bases reporting nothing (bwt.c:369-371), and the next seed starts there
			if (A.bloom && x + A.min_seed_len <= len) {
				uint32_t badw; const uint64_t w = rd.window(x, A.min_seed_len, badw); wc_add(W, EV_BLOOM);
				if (!(badw & ((1u << A.min_seed_len) - 1u)) && !kmer_filter_has(A.bloom, A.bloom_bits, w & ((1ull << (2 * A.min_seed_len)) - 1ull))) {
					my_q += (unsigned)(A.min_seed_len - 1); my_hits += (unsigned)(A.min_seed_len - 1);
					x = x + k1 < len ? x + k1 : len;
					continue;
				}
			}
			Intv ik; int i; bool jumped = false;
			if (jk && x + jk <= len) {
				uint32_t bad; const uint32_t code = rd.kmer(x, jk, bad);
				if (bad <= 3) { uint32_t dummy; unpack_lep(A.jump[code], ik, dummy); wc_add(W, EV_JUMP); i = x + jk; jumped = true; my_q += (unsigned)(jk - 1); my_hits += (unsigned)(jk - 1); }
			}
			if (!jumped) { ik = set_intv(ix, (int)rd.at(x)); i = x + 1; }
			int nx = len;
			bool dead = ik.x2 == 0; // an empty interval stays empty (bwt.c:369 keeps extending it): no more index reads, the
			                        // reference still walks on to the first ambiguous base or to min_seed_len bases and reports nothing
			for (; i < len; ++i) {
				const uint32_t b = rd.at(i);
				if (b > 3) { nx = i + 1; break; }
				if (dead) { if (i - x >= A.min_seed_len) { nx = i + 1; break; } continue; }
				const Intv y = extend1<false>(ix, ik, 3 - (int)b, W); ++my_q;
				if (y.x2 < A.max_mem_intv && i - x >= A.min_seed_len) { if (y.x2 > 0) emit_mem(A, (uint32_t)r, y, (uint32_t)x, (uint32_t)(i + 1)); nx = i + 1; break; }
				ik = y; dead = y.x2 == 0;
			}
			x = nx;
		}
	}
	for (int o = 32; o > 0; o >>= 1) { my_q += __shfl_xor(my_q, o); my_hits += __shfl_xor(my_hits, o); my_text += __shfl_xor(my_text, o); }
	if ((threadIdx.x & 63u) == 0) { atomicAdd(A.n_queries, my_q); if (my_hits) atomicAdd(A.n_sst_hits, my_hits); if (my_text) atomicAdd(n_text_seeds, my_text); }
	lc_flush(C, W);
	wc_flush(W, A.evc, KID_R3TEXT);
}

// ------------------------------------------------------------------------------------------------------------------
// per-read sort by info (comp_seed.cpp:2301) + CSR compaction; a read's mems beyond `cap` come from the overflow
// records, which have been sorted by read id
__device__ __forceinline__ const OutMem &mem_at(const OutMem *src, uint32_t cap, const OvfRec *ovf, const uint32_t *ovf_idx, uint64_t olo, uint32_t a)
{
	return a < cap ? src[a] : ovf[ovf_idx[olo + (a - cap)]].m;
}
__global__ void sort_compact2_kernel(const OutMem *raw, const uint32_t *cnt, uint32_t cap, const OvfRec *ovf, const uint32_t *ovf_key,
                                     const uint32_t *ovf_idx, uint64_t n_ovf, const uint64_t *mem_off, int64_t n_reads, OutMem *mems, int skip_small,
                                     uint64_t *salcnt, uint32_t max_occ)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	uint32_t n = cnt[r];
	if (skip_small && n <= 64 && n <= cap) return; // done by sort_compact16_kernel
	const OutMem *src = raw + (size_t)r * cap;
	OutMem *dst = mems + mem_off[r]; uint64_t *dsc = salcnt + mem_off[r]; // (the SA slots each mem will ask for: comp_seed.cpp:2313-2325)
	if (n <= 16 && n <= cap) { // keys in registers, ranks by 16 x 16 compares, no re-reads
		uint64_t k[16];
#pragma unroll
		for (int a = 0; a < 16; ++a) k[a] = (uint32_t)a < n ? src[a].info : ~0ull;
#pragma unroll
		for (int a = 0; a < 16; ++a) {
			if ((uint32_t)a < n) {
				uint32_t rank = 0;
#pragma unroll
				for (int b = 0; b < 16; ++b) rank += (k[b] < k[a]) || (k[b] == k[a] && b < a); // padding keys are never smaller
				dst[rank] = src[a]; dsc[rank] = src[a].x2 < max_occ ? src[a].x2 : max_occ;
			}
		}
		return;
	}
	uint64_t olo = 0;
	if (n > cap) { // lower bound of r among the sorted overflow keys
		uint64_t lo = 0, hi = n_ovf;
		while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ovf_key[mid] < (uint32_t)r) lo = mid + 1; else hi = mid; }
		olo = lo;
	}
	for (uint32_t a = 0; a < n; ++a) {
		OutMem ma = mem_at(src, cap, ovf, ovf_idx, olo, a);
		uint32_t rank = 0;
		for (uint32_t b = 0; b < n; ++b) { uint64_t kb = mem_at(src, cap, ovf, ovf_idx, olo, b).info; rank += (kb < ma.info) || (kb == ma.info && b < a); }
		dst[rank] = ma; dsc[rank] = ma.x2 < max_occ ? ma.x2 : max_occ;
	}
}
// The reads sort_compact16_kernel leaves out (more than 64 mems, or mems beyond `cap`: tandem arrays, repeats): one WAVE per
// read.  A wave owns 64 consecutive reads, finds the heavy ones by ballot and rank-sorts each with all 64 lanes: lane j
// owns mems j, j+64, ...; the keys of 64 mems at a time sit in registers and travel by shuffle.  (One lane per read made
// this kernel as slow as its slowest read: 5 ms for a handful of reads with hundreds of mems.)
__global__ __launch_bounds__(256) void sort_compact_wave_kernel(const OutMem *raw, const uint32_t *cnt, uint32_t cap, const OvfRec *ovf, const uint32_t *ovf_key,
                                                                const uint32_t *ovf_idx, uint64_t n_ovf, const uint64_t *mem_off, int64_t n_reads, OutMem *mems,
                                                                uint64_t *salcnt, uint32_t max_occ)
{
	const uint32_t lane = threadIdx.x & 63u;
	const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int64_t r0 = w * 64 + lane;
	const uint32_t n_mine = r0 < n_reads ? cnt[r0] : 0;
	uint64_t heavy = __ballot(n_mine > 64 || n_mine > cap);
	while (heavy) {
		const int hs = __ffsll((long long)heavy) - 1; heavy &= heavy - 1;
		const int64_t r = w * 64 + hs; const uint32_t n = __shfl(n_mine, hs);
		const OutMem *src = raw + (size_t)r * cap;
		OutMem *dst = mems + mem_off[r];
		uint64_t olo = 0;
		if (n > cap) { // lower bound of r among the sorted overflow keys
			uint64_t lo = 0, hi = n_ovf;
			while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (ovf_key[mid] < (uint32_t)r) lo = mid + 1; else hi = mid; }
			olo = lo;
		}
		for (uint32_t a0 = 0; a0 < n; a0 += 64) {
			const uint32_t a = a0 + lane;
			OutMem ma = {0, 0, 0, ~0ull};
			if (a < n) ma = mem_at(src, cap, ovf, ovf_idx, olo, a);
			uint32_t rank = 0;
			for (uint32_t b0 = 0; b0 < n; b0 += 64) {
				const uint32_t b = b0 + lane;
				const uint64_t kb = b < n ? mem_at(src, cap, ovf, ovf_idx, olo, b).info : ~0ull; // padding keys are never smaller
				for (int j = 0; j < 64; ++j) {
					const uint64_t kj = __shfl(kb, j);
					rank += (kj < ma.info) || (kj == ma.info && b0 + (uint32_t)j < a);
				}
			}
			if (a < n) { dst[rank] = ma; salcnt[mem_off[r] + rank] = ma.x2 < max_occ ? ma.x2 : max_occ; }
		}
	}
}

// Fast form of the same for the bulk: 16 lanes per read (4 reads per wave).  Up to 16 mems: lane a owns mem a; 17..64 mems
// (repeat-rich reads): lane a owns mems a, a+16, a+32, a+48.  Each mem is read once (coalesced: 16 lanes x 32 B contiguous),
// keys travel by shuffle, and every lane writes its mems at their ranks.  Reads with more than 64 mems, or whose mems
// spilled beyond `cap`, are left to sort_compact2_kernel (launched over the same range with skip_upto = 64).
__global__ __launch_bounds__(256) void sort_compact16_kernel(const OutMem *raw, const uint32_t *cnt, uint32_t cap, const uint64_t *mem_off,
                                                             int64_t n_reads, OutMem *mems, uint64_t *salcnt, uint32_t max_occ)
{
	const uint32_t lane = threadIdx.x & 63u, a = lane & 15u, gbase = lane & ~15u;
	int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	uint32_t n = r < n_reads ? cnt[r] : 0;
	if (n > 64 || n > cap) n = 0; // not ours
	const OutMem *src = raw + (size_t)(r < n_reads ? r : 0) * cap;
	OutMem m[4]; uint64_t key[4]; uint32_t rank[4] = {0, 0, 0, 0};
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		uint32_t e = a + 16u * s;
		key[s] = ~0ull;
		if (e < n) { m[s] = src[e]; key[s] = m[s].info; }
	}
	const int rounds = n > 48 ? 4 : n > 32 ? 3 : n > 16 ? 2 : 1; // group-uniform
	for (int sb = 0; sb < rounds; ++sb) {
#pragma unroll
		for (int b = 0; b < 16; ++b) {
			uint64_t kb = __shfl(sb == 0 ? key[0] : sb == 1 ? key[1] : sb == 2 ? key[2] : key[3], (int)(gbase + b));
			uint32_t eb = (uint32_t)b + 16u * sb; // index of the mem whose key this is
#pragma unroll
			for (int s = 0; s < 4; ++s) rank[s] += (kb < key[s]) || (kb == key[s] && eb < a + 16u * s);
		}
	}
	if (n) {
		OutMem *dst = mems + mem_off[r]; uint64_t *dsc = salcnt + mem_off[r];
#pragma unroll
		for (int s = 0; s < 4; ++s) if (a + 16u * s < n) { dst[rank[s]] = m[s]; dsc[rank[s]] = m[s].x2 < max_occ ? m[s].x2 : max_occ; }
	}
}

__global__ void ovf_keys_kernel(const OvfRec *ovf, uint64_t n, uint32_t *key, uint32_t *idx)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	key[t] = ovf[t].r; idx[t] = (uint32_t)t;
}

} // namespace csd
