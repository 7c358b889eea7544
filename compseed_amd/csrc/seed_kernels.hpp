// seed_kernels.hpp -- the SMEM seeding kernels (gfx950).
//
// smem_kernel replaces, for a whole batch of reads, the "Collect exact matches" block of seed_and_extend()
// (mapping/comp_seed.cpp:2255-2302), i.e. mem_collect_intv (mapping/bwamem.c:218-272) with its three rounds:
//   round 1  SMEMs through successive pivots           bwt_smem1a, FM_index/bwt.c:289-351 (max_intv = 0)
//                                                      == collect_mem_with_sst, comp_seed.cpp:67-139
//   round 2  re-seeding inside long, rare SMEMs        bwamem.c:241-249 / comp_seed.cpp:2276-2285
//   round 3  LAST-like forward seeds                   bwt_seed_strategy1, bwt.c:358-379 == tem_forward_sst, comp_seed.cpp:141-160
//
// Execution model.  The path is a dependent chain of random 64-byte Occ-block reads (one or two lines per
// bwt_extend), so throughput comes from the number of independent chains in flight, not from arithmetic.  One LANE
// owns one read and runs the whole three-round control flow as a small state machine whose loop body contains exactly
// ONE extension site: every iteration each active lane issues the loads for its next bwt_extend, whatever phase
// (forward / backward / round 3) it is in, so divergence is confined to the cheap bookkeeping and all 64 lanes keep a
// line in flight.  Lanes are persistent: a lane that finishes its read pulls the next read index from a global
// counter, so a wavefront stays full until the batch drains.
//
// The only variable-size per-read state is the list of left-extension points (LEPs): the forward intervals at which
// the occurrence count changes (bwt.c:307-318).  The backward sweep (bwt.c:325-345) walks that list longest match
// first and compacts it in place (write index never overtakes the read index), so ONE list per lane suffices.  It
// lives in LDS, packed to 16 bytes per entry (3 x 37-bit coordinates + 16-bit query end), entry-major so that lanes
// touching the same depth are bank-conflict free; entries beyond LEP_LDS spill to a per-lane global region.
#pragma once
#include "fm_device.hpp"

namespace csd {

struct OutMem { uint64_t x0, x1, x2, info; };   // == cs_intv_t / bwtintv_t

struct SeedArgs {
	DevIndex ix;
	const uint8_t  *seq;        // nt4 codes of the sub-batch, 8-byte padded at the end
	const uint64_t *off;        // n_reads + 1 offsets into seq (absolute within the sub-batch buffer)
	const uint32_t *read_ids;   // optional indirection (second pass over overflowed reads) or nullptr
	int64_t   n_tasks;          // number of reads to process in this launch
	OutMem   *out;              // [n_tasks][cap] unsorted mems of each task
	uint32_t *out_cnt;          // [n_tasks]      number of mems found (may exceed cap => task overflowed)
	uint32_t  cap;
	int32_t   min_seed_len, split_len;
	uint32_t  split_width;
	uint64_t  max_mem_intv;
	unsigned long long *task_counter;
	uint4    *spill;            // [grid threads][spill_cap] LEP entries beyond the LDS part
	uint32_t  spill_cap;
	unsigned long long *n_queries; // device counter: bwt_extend queries issued
	unsigned long long *evc;       // byte-model event counters [N_KID][N_EV] (fm_device.hpp), or null
};

enum : int {
	ST_FETCH = 0, ST_R1_NEXT, ST_CALL_START, ST_FWD_CHECK, ST_BWD_INIT, ST_BWD_STEP, ST_BWD_LOAD, ST_CALL_END,
	ST_R2_NEXT, ST_R3_START, ST_R3_CHECK, ST_FINISH, ST_EXIT,
	ST_FWD_WAIT, ST_BWD_WAIT, ST_R3_WAIT
};

__device__ __forceinline__ uint4 pack_lep(const Intv &v, uint32_t end)
{
	uint4 e;
	e.x = (uint32_t)v.x0; e.y = (uint32_t)v.x1; e.z = (uint32_t)v.x2;
	e.w = (uint32_t)(v.x0 >> 32) | (uint32_t)(v.x1 >> 32) << 5 | end << 16; // (bits 10..14 held the top of a 37-bit size; sizes are 32-bit)
	return e;
}
__device__ __forceinline__ void unpack_lep(const uint4 &e, Intv &v, uint32_t &end)
{
	v.x0 = (uint64_t)(e.w & 31u) << 32 | e.x;
	v.x1 = (uint64_t)((e.w >> 5) & 31u) << 32 | e.y;
	v.x2 = e.z;
	end = e.w >> 16;
}

template <int BLOCK, int LEP_LDS, bool COUNT>
__global__ __launch_bounds__(BLOCK, (LEP_LDS <= 10 ? 4 : LEP_LDS <= 13 ? 3 : 2)) void smem_kernel(const SeedArgs A)
{
	__shared__ uint4 lds_lep[LEP_LDS * BLOCK];
	const DevIndex &ix = A.ix;
	const uint32_t tid = threadIdx.x;
	uint4 *my_spill = A.spill + ((size_t)blockIdx.x * BLOCK + tid) * A.spill_cap;

	auto lep_put = [&](int idx, const Intv &v, uint32_t end) {
		uint4 e = pack_lep(v, end);
		if (idx < LEP_LDS) lds_lep[idx * BLOCK + tid] = e; else my_spill[idx - LEP_LDS] = e;
	};
	auto lep_get = [&](int idx, Intv &v, uint32_t &end) {
		uint4 e = (idx < LEP_LDS) ? lds_lep[idx * BLOCK + tid] : my_spill[idx - LEP_LDS];
		unpack_lep(e, v, end);
	};

	// ---- per-lane state
	int st = ST_FETCH;
	int64_t task = 0; uint64_t rbase = 0; int len = 0;
	uint32_t nout = 0; int round = 1;
	int x = 0, i = 0, ret = 0; uint32_t min_intv = 1;
	Intv ik = {0, 0, 0};                 // running interval (forward pass / round 3)
	int n = 0, lo = 0, j = 0, w = 0; bool kept = false; uint64_t last_kept = 0;
	int nm_call = 0, last_start = 0;     // SMEMs emitted by the current call (before the length filter)
	Intv pj = {0, 0, 0}; uint32_t pend = 0; // LEP being extended backward
	int c = 0; bool is_back = false;     // pending extension request
	int r2_k = 0, r2_n = 0;
	uint64_t win = 0; uint32_t win_key = 0xffffffffu; // 8-base window of the read
	unsigned long long my_queries = 0;
	WaveCtrT<COUNT> W;
	OutMem *my_out = nullptr;

	auto qbase = [&](int pos) -> uint32_t {
		uint64_t a = rbase + (uint64_t)pos;
		uint32_t key = (uint32_t)(a >> 3);
		if (key != win_key) { win = *reinterpret_cast<const uint64_t *>(A.seq + (a & ~7ull)); win_key = key; }
		return (uint32_t)(win >> ((a & 7) << 3)) & 0xffu;
	};
	auto emit = [&](const Intv &v, uint32_t beg, uint32_t end) {
		if (nout < A.cap) { OutMem m = {v.x0, v.x1, v.x2, (uint64_t)beg << 32 | end}; my_out[nout] = m; }
		++nout;
	};
	auto emit_smem = [&](const Intv &v, int beg, uint32_t end) { // bwt.c:333-336 + the length filter of bwamem.c:232
		++nm_call; last_start = beg;
		if ((int)end - beg >= A.min_seed_len) emit(v, (uint32_t)beg, end);
	};

	while (st != ST_EXIT) {
		// ---------------------------------------------------------------- advance to the next extension request
		bool need = false;
		while (!need && st != ST_EXIT) {
			switch (st) {
			case ST_FETCH: {
				task = (int64_t)atomicAdd(A.task_counter, 1ull);
				if (task >= A.n_tasks) { st = ST_EXIT; break; }
				uint32_t rid = A.read_ids ? A.read_ids[task] : (uint32_t)task;
				rbase = A.off[rid]; len = (int)(A.off[rid + 1] - rbase);
				my_out = A.out + (size_t)task * A.cap;
				nout = 0; round = 1; x = 0; min_intv = 1; win_key = 0xffffffffu;
				st = ST_R1_NEXT;
			} break;
			case ST_R1_NEXT: // bwamem.c:226-236
				if (x >= len) { round = 2; r2_k = 0; r2_n = (int)nout; st = ST_R2_NEXT; }
				else if (qbase(x) > 3) ++x;
				else st = ST_CALL_START;
				break;
			case ST_CALL_START: // bwt.c:300-301
				ik = set_intv(ix, (int)qbase(x)); i = x + 1; n = 0; st = ST_FWD_CHECK;
				break;
			case ST_FWD_CHECK: { // bwt.c:303-320, the part before bwt_extend
				if (i >= len) { lep_put(n++, ik, (uint32_t)len); ret = len; st = ST_BWD_INIT; break; }
				uint32_t b = qbase(i);
				if (b > 3) { lep_put(n++, ik, (uint32_t)i); ret = i; st = ST_BWD_INIT; break; }
				c = 3 - (int)b; is_back = false; need = true; st = ST_FWD_WAIT;
			} break;
			case ST_BWD_INIT:
				i = x - 1; lo = 0; nm_call = 0; st = ST_BWD_STEP;
				break;
			case ST_BWD_STEP: { // one position of the backward sweep, bwt.c:325-345
				uint32_t b = i < 0 ? 4u : qbase(i);
				if (b > 3) { // read start or ambiguous base: every live match stops here, only the longest can be new
					if (nm_call == 0 || i + 1 < last_start) { lep_get(n - 1, pj, pend); emit_smem(pj, i + 1, pend); }
					st = ST_CALL_END; break;
				}
				c = (int)b; j = n - 1; w = n; kept = false; st = ST_BWD_LOAD;
			} break;
			case ST_BWD_LOAD:
				lep_get(j, pj, pend); is_back = true; need = true; st = ST_BWD_WAIT;
				break;
			case ST_CALL_END:
				if (round == 1) { x = ret; st = ST_R1_NEXT; } else st = ST_R2_NEXT;
				break;
			case ST_R2_NEXT: { // bwamem.c:241-249; threshold in CompSeed's form (comp_seed.cpp:2279), computed on the host
				if (r2_k >= r2_n || r2_k >= (int)A.cap) { st = (A.max_mem_intv > 0) ? ST_R3_START : ST_FINISH; x = 0; round = 3; break; }
				const OutMem *p = my_out + r2_k++;
				uint64_t info = p->info, x2 = p->x2;
				int beg = (int)(info >> 32), end = (int)(uint32_t)info;
				if (end - beg < A.split_len || x2 > A.split_width) break;
				x = (beg + end) >> 1; min_intv = (uint32_t)x2 + 1; st = ST_CALL_START;
			} break;
			case ST_R3_START: // bwamem.c:253-268
				if (x >= len) st = ST_FINISH;
				else if (qbase(x) > 3) ++x;
				else { ik = set_intv(ix, (int)qbase(x)); i = x + 1; st = ST_R3_CHECK; }
				break;
			case ST_R3_CHECK: { // bwt.c:366-377
				if (i >= len) { st = ST_FINISH; break; }
				uint32_t b = qbase(i);
				if (b > 3) { x = i + 1; st = ST_R3_START; break; }
				c = 3 - (int)b; is_back = false; need = true; st = ST_R3_WAIT;
			} break;
			case ST_FINISH:
				A.out_cnt[task] = nout; st = ST_FETCH;
				break;
			default: break;
			}
		}
		if (!need) break; // this lane is done; the wave keeps looping for the others

		// ---------------------------------------------------------------- the one extension site
		Intv src = is_back ? pj : ik;
		Intv y = extend1_rt(ix, src, is_back, c, W);
		++my_queries;

		// ---------------------------------------------------------------- consume
		if (st == ST_FWD_WAIT) { // bwt.c:309-315
			st = ST_FWD_CHECK;
			if (y.x2 != ik.x2) {
				lep_put(n++, ik, (uint32_t)i);
				if (y.x2 < min_intv) { ret = i; st = ST_BWD_INIT; }
			}
			if (st == ST_FWD_CHECK) { ik = y; ++i; }
		} else if (st == ST_BWD_WAIT) { // bwt.c:328-340
			if (y.x2 < min_intv) {
				if (!kept && (nm_call == 0 || i + 1 < last_start)) emit_smem(pj, i + 1, pend);
			} else if (!kept || y.x2 != last_kept) {
				lep_put(--w, y, pend); kept = true; last_kept = y.x2;
			}
			if (--j < lo) {
				if (!kept) st = ST_CALL_END; else { lo = w; --i; st = ST_BWD_STEP; }
			} else st = ST_BWD_LOAD;
		} else { // ST_R3_WAIT, bwt.c:370-375
			if (y.x2 < A.max_mem_intv && i - x >= A.min_seed_len) {
				if (y.x2 > 0) emit(y, (uint32_t)x, (uint32_t)(i + 1));
				x = i + 1; st = ST_R3_START;
			} else { ik = y; ++i; st = ST_R3_CHECK; }
		}
	}
	if (A.n_queries) atomicAdd(A.n_queries, my_queries);
	wc_flush(W, A.evc, KID_FUSED);
}

// -------------------------------------------------------------------------------------------------------------------
// ASCII -> nt4 (nst_nt4_table, FM_index/bntseq.c:46-63); bytes 0..4 pass through as CompSeed does (comp_seed.cpp:2259)
__device__ __forceinline__ uint32_t nt4_of(uint32_t b)
{
	uint32_t u = b & 0xdfu; // fold case
	uint32_t v = (u == 'A') ? 0u : (u == 'C') ? 1u : (u == 'G') ? 2u : (u == 'T') ? 3u : (b == '-') ? 5u : 4u;
	return b <= 4 ? b : v;
}
// 16 bytes per lane when both buffers are 16-byte aligned (they are: hipMalloc / torch allocations), bytes otherwise
__global__ void nt4_kernel(const uint8_t *in, uint8_t *out, uint64_t n)
{
	uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
	uint64_t nv = ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) ? n >> 4 : 0;
	const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
	uint4 *out4 = reinterpret_cast<uint4 *>(out);
	for (uint64_t i = tid; i < nv; i += stride) {
		uint4 v = in4[i], o;
		uint32_t *pv = &v.x, *po = &o.x;
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			uint32_t x = pv[w];
			po[w] = nt4_of(x & 0xff) | nt4_of((x >> 8) & 0xff) << 8 | nt4_of((x >> 16) & 0xff) << 16 | nt4_of(x >> 24) << 24;
		}
		out4[i] = o;
	}
	for (uint64_t i = (nv << 4) + tid; i < n; i += stride) out[i] = (uint8_t)nt4_of(in[i]);
}

// per-read sort by info (comp_seed.cpp:2301) fused with the CSR compaction: rank every mem among its read's mems
__global__ void sort_compact_kernel(const OutMem *raw, const uint32_t *cnt, uint32_t cap, const uint64_t *mem_off, uint64_t base_off,
                                    int64_t n_reads, const uint32_t *slot_of_read, OutMem *mems)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	uint32_t n = cnt[r];
	if (n > cap) return; // overflowed read, filled by the second pass
	const OutMem *src = raw + (size_t)(slot_of_read ? slot_of_read[r] : r) * cap;
	OutMem *dst = mems + base_off + mem_off[r];
	for (uint32_t a = 0; a < n; ++a) {
		uint64_t ka = src[a].info; uint32_t rank = 0;
		for (uint32_t b = 0; b < n; ++b) { uint64_t kb = src[b].info; rank += (kb < ka) || (kb == ka && b < a); }
		dst[rank] = src[a];
	}
}

// ------------------------------------------------------------------------------------------------------------ SAL
// comp_seed.cpp:2313-2325: a mem with x2 occurrences asks for min(x2, max_occ) SA slots x0 + k*step
__global__ void sal_count_kernel(const OutMem *mems, uint64_t n_mems, uint32_t max_occ, uint64_t *cnt)
{
	uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= n_mems) return;
	uint64_t x2 = mems[m].x2;
	cnt[m] = x2 < max_occ ? x2 : max_occ;
}

struct OutSeed { int64_t rbeg; int32_t qbeg, len; }; // == cs_seed_t

// GATHER: the slot is looked up in the HBM-resident full suffix array right here (one pass over the seeds instead of two).
// A lane per mem, but a mem with more than SAL_LIGHT slots (a repeat: up to max_occ = 500 of them) is expanded by the whole wave, 64 slots
// at a time with coalesced stores -- one lane looping over 500 slots while 63 wait was what SAL cost on repeat-rich genomes.
constexpr uint32_t SAL_LIGHT = 4;
template <bool GATHER>
__global__ void sal_expand_kernel(const DevIndex ix, const OutMem *mems, uint64_t n_mems, uint32_t max_occ, const uint64_t *seed_of_mem, OutSeed *seeds)
{
	const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	const bool valid = m < n_mems;
	OutMem v = valid ? mems[m] : OutMem{0, 0, 0, 0};
	const uint64_t step = v.x2 > max_occ ? v.x2 / max_occ : 1;
	const uint64_t first = valid ? seed_of_mem[m] : 0;
	// slots of this mem (comp_seed.cpp:2313-2325: k = 0, step, 2 step, ... below x2, at most max_occ of them)
	const uint64_t by_step = (v.x2 + step - 1) / step;
	const uint32_t cnt = valid ? (uint32_t)(by_step < max_occ ? by_step : max_occ) : 0;
	const int32_t qb = (int32_t)(v.info >> 32), ln = (int32_t)(uint32_t)v.info - qb;
	if (cnt <= SAL_LIGHT) {
		OutSeed *dst = seeds + first;
		for (uint32_t c = 0; c < cnt; ++c) {
			const uint64_t slot = v.x0 + (uint64_t)c * step;
			OutSeed s = {GATHER ? (int64_t)sa_direct(ix, slot) : (int64_t)slot, qb, ln};
			dst[c] = s;
		}
	}
	unsigned long long heavy = __ballot(cnt > SAL_LIGHT);
	while (heavy) { // wave-uniform
		const int j = __builtin_ctzll(heavy); heavy &= heavy - 1;
		const uint64_t x0 = __shfl(v.x0, j), st = __shfl(step, j), fj = __shfl(first, j);
		const uint32_t cj = __shfl(cnt, j);
		const int32_t qbj = __shfl(qb, j), lnj = __shfl(ln, j);
		for (uint32_t c = (uint32_t)lane; c < cj; c += 64) {
			const uint64_t slot = x0 + (uint64_t)c * st;
			OutSeed s = {GATHER ? (int64_t)sa_direct(ix, slot) : (int64_t)slot, qbj, lnj};
			seeds[fj + c] = s;
		}
	}
}

// bwt_sa on every requested slot (comp_seed.cpp:2336-2345); rbeg holds the slot on entry, the coordinate on exit
__global__ void sal_walk_kernel(const DevIndex ix, OutSeed *seeds, uint64_t n_seeds)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_seeds) return;
	seeds[s].rbeg = (int64_t)sa_lookup(ix, (uint64_t)seeds[s].rbeg);
}
// the same through the HBM-resident full suffix array: one gather per slot
__global__ void sal_gather_kernel(const DevIndex ix, OutSeed *seeds, uint64_t n_seeds)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_seeds) return;
	seeds[s].rbeg = (int64_t)sa_direct(ix, (uint64_t)seeds[s].rbeg);
}

// Materialise SA[row] for every row from the 1-in-sa_intv samples: lane t starts at sampled row t*sa_intv, whose value is
// known, and follows bwt_invPsi (one text position back per step, bwt.c:53-59) writing SA = value - steps until it
// reaches the next sampled row.  Every row lies on exactly one such chain, so all seq_len+1 rows get written once.
template <typename T>
__global__ void sa_fill_kernel(const DevIndex ix, T *full)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ix.n_sa) return;
	uint64_t k = t << ix.sa_shift;
	uint64_t s = (t == 0) ? ix.seq_len : ix.sa[t]; // row 0 is the "$" suffix at text position seq_len
	full[k] = (T)s;
	for (;;) {
		k = inv_psi(ix, k);
		if ((k & ix.sa_mask) == 0) break;
		--s;
		full[k] = (T)s;
	}
}

__global__ void seed_off_kernel(const uint64_t *mem_off, const uint64_t *seed_of_mem, int64_t n_reads, uint64_t *seed_off)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads) return;
	seed_off[r] = seed_of_mem[mem_off[r]];
}

// ------------------------------------------------------------------------------------------------------------ primitive test entries
__global__ void occ4_kernel(const DevIndex ix, const uint64_t *k, uint64_t *cnt, int64_t n)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	uint64_t c[4]; occ4(ix, k[t], c);
	cnt[4 * t] = c[0]; cnt[4 * t + 1] = c[1]; cnt[4 * t + 2] = c[2]; cnt[4 * t + 3] = c[3];
}
__global__ void extend_kernel(const DevIndex ix, const OutMem *ik, const uint8_t *is_back, OutMem *ok, int64_t n)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	Intv64 v64 = {ik[t].x0, ik[t].x1, ik[t].x2}, o[4];
	extend4(ix, v64, is_back[t] != 0, o);
	const bool small = ik[t].x2 < (1ull << 32);            // the single-child forms serve the search: sizes below 2^32
	Intv v = {ik[t].x0, ik[t].x1, (uint32_t)ik[t].x2};
	for (int c = 0; c < 4; ++c) {
		OutMem m = {o[c].x0, o[c].x1, o[c].x2, 0};
		ok[4 * t + c] = m;
		// the single-child paths used by the search must agree with the four-child one
		NoCtr W;
		Intv o1 = extend1_rt(ix, v, is_back[t] != 0, c, W);
		Intv o2 = is_back[t] ? extend1<true>(ix, v, c, W) : extend1<false>(ix, v, c, W);
		if (small && v.x0 != 0 && v.x1 != 0 && (o2.x0 != o[c].x0 || o2.x1 != o[c].x1 || o2.x2 != o[c].x2)) ok[4 * t + c].info = 2;
		if (small && (o1.x0 != o[c].x0 || o1.x1 != o[c].x1 || o1.x2 != o[c].x2)) ok[4 * t + c].info = 1;
	}
}
__global__ void sa_kernel(const DevIndex ix, const uint64_t *k, uint64_t *sa, int64_t n)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	uint64_t walked = sa_lookup(ix, k[t]);
	// when the full suffix array is resident it must agree with the walk on every row (mismatch => poison the answer)
	if ((ix.fsa32 || ix.fsa64) && sa_direct(ix, k[t]) != walked) walked = 0xdeadbeefdeadbeefull;
	sa[t] = walked;
}

} // namespace csd
