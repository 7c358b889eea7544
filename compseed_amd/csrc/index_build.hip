// index_build.hip -- FM-index construction on the GPU (replaces `bwaidx`, FM_index/index_main.c:257-325).
//
// Produces, for a forward-strand genome given as 2-bit codes, exactly the arrays the reference's builder writes:
//   text  T = forward ++ reverse-complement            (bns_fasta2bntseq with for_only = 0, FM_index/bntseq.c:306-312)
//   BWT of T$ with the $ row removed, `primary` = its row, L2 = cumulative base counts   (bwt_pac2bwt, index_main.c:66-127)
//   Occ interleaved every 128 rows + one trailing record  (bwt_bwtupdate_core, index_main.c:152-174)
//   SA sampled every 32 rows by ROW index, sa[0] = -1      (bwt_cal_sa, FM_index/bwt.c:62-84)
// The reference gets there with SA-IS (is.c) or BWT-SW (bwt_gen.c) on one CPU thread; both yield the unique BWT of
// T$, so any correct suffix sort gives byte-identical .bwt/.sa files (tests compare against the bwaidx-built fixture).
//
// GPU algorithm: prefix doubling with discarding.  One radix sort of (29-mer + end marker) keys orders all but the
// suffixes inside long repeats; those stay in "unresolved" groups that are re-sorted by the rank of the suffix h
// symbols further on (h = 29, 58, 116, ...) until every group is a singleton.  Only unresolved suffixes are touched
// after the first pass, so a genome with few long repeats costs ~one 64-bit radix sort of N keys.
#include "cs_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#define HIPB(expr)                                                                                         \
	do {                                                                                                   \
		hipError_t e__ = (expr);                                                                           \
		if (e__ != hipSuccess) {                                                                           \
			(void)hipGetLastError();                                                                       \
			rc = cs_fail_(e__ == hipErrorOutOfMemory ? CS_ENOMEM : CS_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
			goto done;                                                                                     \
		}                                                                                                  \
	} while (0)

namespace {

// idx_t = type of suffix positions / ranks: uint32_t while N + 1 < 2^32, else uint64_t (hg19: N = 6.2e9)

constexpr int K0 = 29; // symbols in the first-pass key: 58 bits + 6 bits of min(suffix length, 29)

__global__ void make_text_kernel(const uint8_t *fwd, uint64_t g, uint8_t *T)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * g; i += (uint64_t)gridDim.x * blockDim.x) {
		T[i] = i < g ? (fwd[i] & 3) : (uint8_t)(3 - (fwd[2 * g - 1 - i] & 3));
	}
}

template <typename idx_t>
__global__ void make_keys_kernel(const uint8_t *T, uint64_t n, uint64_t *key, idx_t *sfx)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t k = 0;
		for (int t = 0; t < K0; ++t) { uint64_t p = i + t; k = k << 2 | (p < n ? T[p] : 0); }
		uint64_t len = n - i;
		key[i] = k << 6 | (len < K0 ? len : K0); // a suffix that ends inside the window sorts before its A-padded look-alikes
		sfx[i] = (idx_t)i;
	}
}

// head[j] = 1 when slot j starts a new group of equal keys
template <typename idx_t>
__global__ void heads_kernel(const uint64_t *key, uint64_t m, idx_t *headpos)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
		headpos[j] = (j == 0 || key[j] != key[j - 1]) ? (idx_t)j : 0;
	}
}
template <typename idx_t> struct MaxOp { __device__ idx_t operator()(idx_t a, idx_t b) const { return a > b ? a : b; } };

// first pass: rank[suffix] = slot of its group head; flag the members of groups larger than one
template <typename idx_t>
__global__ void first_ranks_kernel(const idx_t *sfx, const idx_t *head_of, uint64_t m, idx_t *rank, uint8_t *unres)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
		rank[sfx[j]] = head_of[j];
		bool head = (uint64_t)head_of[j] == j;
		bool next_head = (j + 1 == m || (uint64_t)head_of[j + 1] == j + 1);
		unres[j] = (head && next_head) ? 0u : 1u;
	}
}
template <typename idx_t>
__global__ void compact_first_kernel(const uint8_t *unres, const uint64_t *pos, const idx_t *sfx, const idx_t *head_of, uint64_t m,
                                     idx_t *slot, idx_t *isfx, idx_t *grp)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x) {
		if (!unres[j]) continue;
		uint64_t p = pos[j];
		slot[p] = (idx_t)j; isfx[p] = sfx[j]; grp[p] = head_of[j];
	}
}
// dense ordinal of each item's group (items are grouped and ordered by slot): flag = 1 at the first item of a group
template <typename idx_t>
__global__ void group_flag_kernel(const idx_t *grp, uint64_t u, uint8_t *flag)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < u; p += (uint64_t)gridDim.x * blockDim.x) {
		flag[p] = (p == 0 || grp[p] != grp[p - 1]) ? 1u : 0u;
	}
}
template <typename idx_t>
__global__ void round_keys_kernel(const idx_t *isfx, const uint64_t *gord_incl, const idx_t *rank, uint64_t u, uint64_t h, uint64_t *key)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < u; p += (uint64_t)gridDim.x * blockDim.x) {
		// every member of an unresolved group has at least h symbols, so isfx + h <= n and rank[] is defined there
		key[p] = (gord_incl[p] - 1) << 33 | (uint64_t)rank[(uint64_t)isfx[p] + h];
	}
}
// after sorting a round's items: new group head positions (in item space)
template <typename idx_t>
__global__ void round_heads_kernel(const uint64_t *key, uint64_t u, idx_t *headpos)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < u; p += (uint64_t)gridDim.x * blockDim.x) {
		headpos[p] = (p == 0 || key[p] != key[p - 1]) ? (idx_t)p : 0;
	}
}
template <typename idx_t>
__global__ void round_update_kernel(const uint64_t *key, const idx_t *slot, const idx_t *isfx_sorted, const idx_t *head_of, uint64_t u,
                                    idx_t *sa, idx_t *rank, idx_t *new_grp, uint8_t *unres)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < u; p += (uint64_t)gridDim.x * blockDim.x) {
		idx_t s = isfx_sorted[p];
		idx_t g = slot[head_of[p]]; // SA slot of the (sub)group head = the rank shared by the subgroup
		sa[slot[p]] = s;
		rank[s] = g;
		new_grp[p] = g;
		bool head = (p == 0 || key[p] != key[p - 1]);
		bool next_head = (p + 1 == u || key[p + 1] != key[p]);
		unres[p] = (head && next_head) ? 0u : 1u;
	}
}
template <typename idx_t>
__global__ void compact_round_kernel(const uint8_t *unres, const uint64_t *pos, const idx_t *slot, const idx_t *isfx, const idx_t *grp,
                                     uint64_t u, idx_t *slot2, idx_t *isfx2, idx_t *grp2)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < u; p += (uint64_t)gridDim.x * blockDim.x) {
		if (!unres[p]) continue;
		uint64_t q = pos[p];
		slot2[q] = slot[p]; isfx2[q] = isfx[p]; grp2[q] = grp[p];
	}
}

template <typename idx_t>
__global__ void find_primary_kernel(const idx_t *sa, uint64_t m, unsigned long long *primary)
{
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (uint64_t)gridDim.x * blockDim.x)
		if (sa[r] == 0) *primary = r;
}
// one thread per 16-base word of the $-removed BWT string (index_main.c:123-125); also per-word base counts
template <typename idx_t>
__global__ void bwt_words_kernel(const idx_t *sa, const uint8_t *T, uint64_t n, uint64_t primary, uint64_t n_words, uint32_t *words)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t v = 0;
		for (int t = 0; t < 16; ++t) {
			uint64_t pos = w * 16 + t;
			if (pos >= n) break;
			uint64_t row = pos + (pos >= primary);
			uint32_t c = T[(uint64_t)sa[row] - 1];
			v |= c << ((15 - t) << 1);
		}
		words[w] = v;
	}
}
__device__ inline void count_word(uint32_t w, uint32_t nb, uint32_t c[4])
{
	uint32_t m = nb >= 16 ? 0x55555555u : (nb == 0 ? 0u : (0x55555555u & ~((1u << ((16 - nb) << 1)) - 1u)));
	uint32_t lo = w, hi = w >> 1;
	uint32_t c1 = __builtin_popcount(lo & ~hi & m), c2 = __builtin_popcount(hi & ~lo & m), c3 = __builtin_popcount(lo & hi & m);
	c[0] += nb - c1 - c2 - c3; c[1] += c1; c[2] += c2; c[3] += c3;
}
// per 128-row block: number of A,C,G,T in it
__global__ void block_counts_kernel(const uint32_t *words, uint64_t n, uint64_t n_blocks, uint64_t *cA, uint64_t *cC, uint64_t *cG, uint64_t *cT)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c[4] = {0, 0, 0, 0};
		for (int w = 0; w < 8; ++w) {
			uint64_t first = b * 128 + (uint64_t)w * 16;
			if (first >= n) break;
			uint32_t nb = (n - first) < 16 ? (uint32_t)(n - first) : 16u;
			count_word(words[b * 8 + w], nb, c);
		}
		cA[b] = c[0]; cC[b] = c[1]; cG[b] = c[2]; cT[b] = c[3];
	}
}
// final interleaved array (bwt_bwtupdate_core, index_main.c:152-174); oA..oT are exclusive prefix sums with n_blocks+1 entries
__global__ void interleave_kernel(const uint32_t *words, uint64_t n_words, uint64_t n_blocks, const uint64_t *oA, const uint64_t *oC,
                                  const uint64_t *oG, const uint64_t *oT, uint32_t *out)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t *dst = out + b * 16;
		uint64_t h[4] = {oA[b], oC[b], oG[b], oT[b]};
		for (int i = 0; i < 4; ++i) { dst[2 * i] = (uint32_t)h[i]; dst[2 * i + 1] = (uint32_t)(h[i] >> 32); }
		if (b == n_blocks) continue; // trailing record: counts only
		for (int w = 0; w < 8; ++w) { uint64_t wi = b * 8 + w; if (wi < n_words) dst[8 + w] = words[wi]; }
	}
}
template <typename idx_t>
__global__ void sa_sample_kernel(const idx_t *sa, uint64_t m, uint32_t shift, uint64_t n_sa, uint64_t *out)
{
	// grid-stride: an HSA dispatch holds fewer than 2^32 work-items, hg19 has 6.2e9 rows
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_sa; t += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t r = t << shift;
		out[t] = (t == 0) ? ~0ull : (r < m ? (uint64_t)sa[r] : 0);
	}
}

inline unsigned gridof(uint64_t n, unsigned block = 256) { return (unsigned)std::min<uint64_t>(1u << 22, std::max<uint64_t>(1, (n + block - 1) / block)); } // kernels are grid-stride

struct U8ToU64 { __device__ uint64_t operator()(uint8_t v) const { return (uint64_t)v; } };
__global__ void add_flag_kernel(const uint8_t *flag, uint64_t *pos, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) pos[i] += flag[i];
}
template <typename idx_t> __global__ void carry_max_kernel(idx_t *p) { if (p[-1] > p[0]) p[0] = p[-1]; }

} // namespace

template <typename idx_t>
static int build_impl(const uint8_t *fwd_nt4, uint64_t l_pac, int device, bool verbose, cs_index_t **out)
{
	int rc = CS_OK;
	const uint64_t n = 2 * l_pac, m = n + 1; // m rows including the $ row

	hipStream_t s = nullptr;
	uint8_t *d_fwd = nullptr, *d_T = nullptr; void *d_tmp = nullptr; size_t tmp_cap = 0;
	uint64_t *d_key = nullptr, *d_key2 = nullptr, *d_pos = nullptr;
	idx_t *d_sfx = nullptr, *d_sfx2 = nullptr, *d_head = nullptr, *d_rank = nullptr;
	uint8_t *d_flag = nullptr;
	idx_t *slot = nullptr, *isfx = nullptr, *grp = nullptr, *slot2 = nullptr, *isfx2 = nullptr, *grp2 = nullptr, *isfx_s = nullptr;
	unsigned long long *d_prim = nullptr; unsigned long long h_prim = 0;
	uint32_t *d_words = nullptr, *d_bwt = nullptr; uint64_t *d_cnt = nullptr, *d_occ = nullptr, *d_sa = nullptr;
	uint64_t u = 0, hstep = K0;
	cs_index *ix = nullptr;
	uint64_t n_words = (n + 15) >> 4, n_blocks = (n + 127) >> 7, bwt_size = n_words + (n_blocks + 1) * 8, n_sa = (n + 32) / 32;
	uint64_t last = 0;
	idx_t *d_sa32 = nullptr;

	auto need_tmp = [&](size_t bytes) -> hipError_t {
		if (bytes <= tmp_cap) return hipSuccess;
		if (d_tmp) (void)hipFree(d_tmp);
		d_tmp = nullptr; tmp_cap = 0;
		hipError_t e = hipMalloc(&d_tmp, bytes + 256);
		if (e == hipSuccess) tmp_cap = bytes + 256;
		return e;
	};
	// rocPRIM 4.2's device scans truncate their size to 32 bits (measured: a 5e9-element scan stops after 705 M elements),
	// so every scan that can exceed 2^32 rows runs in 2^30-element pieces with the running value carried between them.
	const uint64_t PIECE = 1ull << 30;
	auto scan_flags = [&](const uint8_t *flag, uint64_t *pos, uint64_t cnt, bool inclusive) -> hipError_t {
		uint64_t carry = 0;
		for (uint64_t o = 0; o < cnt; o += PIECE) {
			uint64_t c = std::min<uint64_t>(PIECE, cnt - o);
			auto in = rocprim::make_transform_iterator(flag + o, U8ToU64());
			size_t tb = 0;
			hipError_t e = rocprim::exclusive_scan(nullptr, tb, in, pos + o, carry, (size_t)c, rocprim::plus<uint64_t>(), s);
			if (e != hipSuccess) return e;
			if ((e = need_tmp(tb)) != hipSuccess) return e;
			if ((e = rocprim::exclusive_scan(d_tmp, tb, in, pos + o, carry, (size_t)c, rocprim::plus<uint64_t>(), s)) != hipSuccess) return e;
			if (o + c < cnt || inclusive) { // carry = exclusive value of the last element + its flag
				uint64_t lastv = 0; uint8_t lastf = 0;
				if ((e = hipMemcpyAsync(&lastv, pos + o + c - 1, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
				if ((e = hipMemcpyAsync(&lastf, flag + o + c - 1, 1, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
				if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
				carry = lastv + lastf;
			}
		}
		if (inclusive) hipLaunchKernelGGL(add_flag_kernel, dim3(gridof(cnt)), dim3(256), 0, s, flag, pos, cnt); // exclusive -> inclusive
		return hipGetLastError();
	};
	auto scan_max = [&](idx_t *io, uint64_t cnt) -> hipError_t {
		for (uint64_t o = 0; o < cnt; o += PIECE) {
			uint64_t c = std::min<uint64_t>(PIECE, cnt - o);
			if (o) hipLaunchKernelGGL((carry_max_kernel<idx_t>), dim3(1), dim3(1), 0, s, io + o); // io[o] = max(io[o], io[o-1])
			size_t tb = 0;
			hipError_t e = rocprim::inclusive_scan(nullptr, tb, io + o, io + o, (size_t)c, MaxOp<idx_t>(), s);
			if (e != hipSuccess) return e;
			if ((e = need_tmp(tb)) != hipSuccess) return e;
			if ((e = rocprim::inclusive_scan(d_tmp, tb, io + o, io + o, (size_t)c, MaxOp<idx_t>(), s)) != hipSuccess) return e;
		}
		return hipSuccess;
	};
	// sorts (k0,v0) using (k1,v1) as the ping-pong partner; on return k0/v0 point at the sorted data and k1/v1 at the
	// scratch copy (rocPRIM's double_buffer form: no hidden third buffer, which matters at 50 GB per array)
	auto sort_pairs = [&](uint64_t *&k0, uint64_t *&k1, idx_t *&v0, idx_t *&v1, uint64_t cnt, unsigned bits) -> hipError_t {
		rocprim::double_buffer<uint64_t> kb(k0, k1);
		rocprim::double_buffer<idx_t> vb(v0, v1);
		size_t tb = 0;
		hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, kb, vb, (size_t)cnt, 0u, bits, s);
		if (e != hipSuccess) return e;
		if ((e = need_tmp(tb)) != hipSuccess) return e;
		e = rocprim::radix_sort_pairs(d_tmp, tb, kb, vb, (size_t)cnt, 0u, bits, s);
		if (e != hipSuccess) return e;
		if (kb.current() != k0) std::swap(k0, k1);
		if (vb.current() != v0) std::swap(v0, v1);
		return hipSuccess;
	};

	auto stage = [&](const char *what) {
		if (!verbose) return;
		size_t fr = 0, to = 0; (void)hipMemGetInfo(&fr, &to);
		fprintf(stderr, "[cs_index_build] %s (idx %zu B, rows %llu, free %.1f GB)\n", what, sizeof(idx_t), (unsigned long long)m, fr / 1e9); fflush(stderr);
	};
	HIPB(hipSetDevice(device));
	HIPB(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	stage("start");
	HIPB(hipMalloc((void **)&d_fwd, l_pac));
	HIPB(hipMalloc((void **)&d_T, n + 64));
	HIPB(hipMemcpyAsync(d_fwd, fwd_nt4, l_pac, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(make_text_kernel, dim3(gridof(n)), dim3(256), 0, s, d_fwd, l_pac, d_T);
	HIPB(hipStreamSynchronize(s));
	(void)hipFree(d_fwd); d_fwd = nullptr;

	stage("text ready");
	// ---- first pass: sort all m suffixes by their 29-symbol key
	HIPB(hipMalloc((void **)&d_key, m * 8)); HIPB(hipMalloc((void **)&d_key2, m * 8));
	HIPB(hipMalloc((void **)&d_sfx, m * sizeof(idx_t))); HIPB(hipMalloc((void **)&d_sfx2, m * sizeof(idx_t)));
	hipLaunchKernelGGL((make_keys_kernel<idx_t>), dim3(gridof(m)), dim3(256), 0, s, d_T, n, d_key, d_sfx);
	stage("keys made");
	HIPB(sort_pairs(d_key, d_key2, d_sfx, d_sfx2, m, 64));
	HIPB(hipStreamSynchronize(s));
	stage("first sort done");
	if (d_tmp) { (void)hipFree(d_tmp); d_tmp = nullptr; tmp_cap = 0; }
	std::swap(d_key, d_key2); std::swap(d_sfx, d_sfx2);
	// d_key2 / d_sfx2 now hold the sorted keys and the provisional suffix array.  Memory matters at hg19 scale (6.2e9
	// rows x 8 B = 50 GB per array), so the scratch copies are recycled: d_sfx becomes the group-head array and d_key the
	// prefix-sum array, and the sorted keys are dropped as soon as the group heads are known.
	d_head = d_sfx; d_sfx = nullptr;
	d_pos = d_key; d_key = nullptr;
	hipLaunchKernelGGL((heads_kernel<idx_t>), dim3(gridof(m)), dim3(256), 0, s, d_key2, m, d_head);
	HIPB(scan_max(d_head, m));
	HIPB(hipStreamSynchronize(s));
	stage("group heads done");
	(void)hipFree(d_key2); d_key2 = nullptr;
	HIPB(hipMalloc((void **)&d_rank, (m + 1) * sizeof(idx_t)));
	HIPB(hipMalloc((void **)&d_flag, m + 16));
	hipLaunchKernelGGL((first_ranks_kernel<idx_t>), dim3(gridof(m)), dim3(256), 0, s, d_sfx2, d_head, m, d_rank, d_flag);
	HIPB(scan_flags(d_flag, d_pos, m, false));
	HIPB(hipMemcpyAsync(&last, d_pos + (m - 1), 8, hipMemcpyDeviceToHost, s));
	{ uint8_t lf = 0; HIPB(hipMemcpyAsync(&lf, d_flag + (m - 1), 1, hipMemcpyDeviceToHost, s)); HIPB(hipStreamSynchronize(s)); u = last + lf; }
	if (verbose) fprintf(stderr, "[cs_index_build] unresolved suffixes after the first pass: %llu\n", (unsigned long long)u);
	stage("ranks done");
	if (u) {
		HIPB(hipMalloc((void **)&slot, u * sizeof(idx_t))); HIPB(hipMalloc((void **)&isfx, u * sizeof(idx_t))); HIPB(hipMalloc((void **)&grp, u * sizeof(idx_t)));
		hipLaunchKernelGGL((compact_first_kernel<idx_t>), dim3(gridof(m)), dim3(256), 0, s, d_flag, d_pos, d_sfx2, d_head, m, slot, isfx, grp);
	}
	HIPB(hipStreamSynchronize(s));
	(void)hipFree(d_head); d_head = nullptr; (void)hipFree(d_pos); d_pos = nullptr; // 2 x 50 GB back before the item arrays
	if (u) { // item-space work arrays
		HIPB(hipMalloc((void **)&slot2, u * sizeof(idx_t))); HIPB(hipMalloc((void **)&isfx2, u * sizeof(idx_t))); HIPB(hipMalloc((void **)&grp2, u * sizeof(idx_t)));
		HIPB(hipMalloc((void **)&isfx_s, u * sizeof(idx_t)));
		HIPB(hipMalloc((void **)&d_pos, (u + 1) * 8));
		HIPB(hipMalloc((void **)&d_key, u * 8)); HIPB(hipMalloc((void **)&d_key2, u * 8)); HIPB(hipMalloc((void **)&d_head, u * sizeof(idx_t)));
	}
	// ---- doubling rounds over the unresolved items only
	for (int round = 0; u > 0; ++round) {
		if (round > 40) { rc = cs_fail_(CS_EDEVICE, "cs_index_build: prefix doubling did not converge"); goto done; }
		hipLaunchKernelGGL((group_flag_kernel<idx_t>), dim3(gridof(u)), dim3(256), 0, s, grp, u, d_flag);
		HIPB(scan_flags(d_flag, d_pos, u, true));
		hipLaunchKernelGGL((round_keys_kernel<idx_t>), dim3(gridof(u)), dim3(256), 0, s, isfx, d_pos, d_rank, u, hstep, d_key);
		HIPB(sort_pairs(d_key, d_key2, isfx, isfx_s, u, 64));
		std::swap(d_key, d_key2); std::swap(isfx, isfx_s); // sorted keys / suffixes are now d_key2 / isfx_s as below
		hipLaunchKernelGGL((round_heads_kernel<idx_t>), dim3(gridof(u)), dim3(256), 0, s, d_key2, u, d_head);
		HIPB(scan_max(d_head, u));
		hipLaunchKernelGGL((round_update_kernel<idx_t>), dim3(gridof(u)), dim3(256), 0, s, d_key2, slot, isfx_s, d_head, u, d_sfx2, d_rank, grp, d_flag);
		HIPB(scan_flags(d_flag, d_pos, u, false));
		uint8_t lf = 0;
		HIPB(hipMemcpyAsync(&last, d_pos + (u - 1), 8, hipMemcpyDeviceToHost, s));
		HIPB(hipMemcpyAsync(&lf, d_flag + (u - 1), 1, hipMemcpyDeviceToHost, s));
		HIPB(hipStreamSynchronize(s));
		uint64_t u2 = last + lf;
		if (u2) hipLaunchKernelGGL((compact_round_kernel<idx_t>), dim3(gridof(u)), dim3(256), 0, s, d_flag, d_pos, slot, isfx_s, grp, u, slot2, isfx2, grp2);
		HIPB(hipStreamSynchronize(s));
		std::swap(slot, slot2); std::swap(isfx, isfx2); std::swap(grp, grp2);
		u = u2; hstep <<= 1;
		if (verbose) { fprintf(stderr, "[cs_index_build] round %d done, h = %llu, unresolved %llu\n", round, (unsigned long long)hstep, (unsigned long long)u); fflush(stderr); }
	}
	d_sa32 = d_sfx2;
	for (void *p : {(void *)d_key, (void *)d_key2, (void *)d_head, (void *)d_rank, (void *)d_flag, (void *)d_pos, (void *)slot, (void *)isfx, (void *)grp,
	                (void *)slot2, (void *)isfx2, (void *)grp2, (void *)isfx_s}) if (p) (void)hipFree(p);
	d_key = d_key2 = d_pos = nullptr; d_head = d_rank = nullptr; d_flag = nullptr; slot = isfx = grp = slot2 = isfx2 = grp2 = isfx_s = nullptr;

	stage("suffix array complete");
	// ---- BWT, Occ, SA samples
	HIPB(hipMalloc((void **)&d_prim, 8));
	hipLaunchKernelGGL((find_primary_kernel<idx_t>), dim3(gridof(m)), dim3(256), 0, s, d_sa32, m, d_prim);
	HIPB(hipMemcpyAsync(&h_prim, d_prim, 8, hipMemcpyDeviceToHost, s));
	HIPB(hipStreamSynchronize(s));
	HIPB(hipMalloc((void **)&d_words, (n_words + 8) * 4));
	hipLaunchKernelGGL((bwt_words_kernel<idx_t>), dim3(gridof(n_words)), dim3(256), 0, s, d_sa32, d_T, n, (uint64_t)h_prim, n_words, d_words);
	HIPB(hipMalloc((void **)&d_cnt, 4 * (n_blocks + 1) * 8)); HIPB(hipMalloc((void **)&d_occ, 4 * (n_blocks + 1) * 8));
	HIPB(hipMemsetAsync(d_cnt, 0, 4 * (n_blocks + 1) * 8, s));
	hipLaunchKernelGGL(block_counts_kernel, dim3(gridof(n_blocks)), dim3(256), 0, s, d_words, n, n_blocks, d_cnt, d_cnt + (n_blocks + 1),
	                   d_cnt + 2 * (n_blocks + 1), d_cnt + 3 * (n_blocks + 1));
	for (int c = 0; c < 4; ++c) {
		size_t tb = 0;
		uint64_t *in = d_cnt + c * (n_blocks + 1), *o = d_occ + c * (n_blocks + 1);
		HIPB(rocprim::exclusive_scan(nullptr, tb, in, o, (uint64_t)0, (size_t)n_blocks + 1, rocprim::plus<uint64_t>(), s));
		HIPB(need_tmp(tb));
		HIPB(rocprim::exclusive_scan(d_tmp, tb, in, o, (uint64_t)0, (size_t)n_blocks + 1, rocprim::plus<uint64_t>(), s));
	}
	HIPB(hipMalloc((void **)&d_bwt, (bwt_size + 16) * 4));
	HIPB(hipMemsetAsync(d_bwt, 0, (bwt_size + 16) * 4, s));
	// note: a partial last block is followed immediately by the trailing record in the reference layout, so the
	// record of block n_blocks starts at word n_words + 8*n_blocks, not at 16*n_blocks
	hipLaunchKernelGGL(interleave_kernel, dim3(gridof(n_blocks + 1)), dim3(256), 0, s, d_words, n_words, n_blocks, d_occ, d_occ + (n_blocks + 1),
	                   d_occ + 2 * (n_blocks + 1), d_occ + 3 * (n_blocks + 1), d_bwt);
	HIPB(hipMalloc((void **)&d_sa, n_sa * 8));
	hipLaunchKernelGGL((sa_sample_kernel<idx_t>), dim3(gridof(n_sa)), dim3(256), 0, s, d_sa32, m, 5u, n_sa, d_sa);
	HIPB(hipGetLastError());

	HIPB(hipStreamSynchronize(s));
	stage("bwt/occ/sa kernels done");
	ix = new cs_index();
	memset(&ix->v, 0, sizeof ix->v);
	ix->bwt.resize(bwt_size); ix->sa.resize(n_sa);
	{
		uint64_t tot[4];
		for (int c = 0; c < 4; ++c) HIPB(hipMemcpyAsync(&tot[c], d_occ + c * (n_blocks + 1) + n_blocks, 8, hipMemcpyDeviceToHost, s));
		// the trailing record sits right behind the last (possibly partial) block
		uint64_t full_words = n_blocks * 16; // where interleave_kernel put it
		HIPB(hipMemcpyAsync(ix->bwt.data(), d_bwt, std::min<uint64_t>(bwt_size, full_words) * 4, hipMemcpyDeviceToHost, s));
		HIPB(hipMemcpyAsync(ix->sa.data(), d_sa, n_sa * 8, hipMemcpyDeviceToHost, s));
		HIPB(hipStreamSynchronize(s));
		uint64_t rec_at = n_words + 8 * n_blocks; // == bwt_size - 8
		for (int c = 0; c < 4; ++c) { ix->bwt[rec_at + 2 * c] = (uint32_t)tot[c]; ix->bwt[rec_at + 2 * c + 1] = (uint32_t)(tot[c] >> 32); }
		cs_index_view_t &v = ix->v;
		v.primary = h_prim; v.L2[0] = 0;
		for (int c = 0; c < 4; ++c) v.L2[c + 1] = v.L2[c] + tot[c];
		v.seq_len = n; v.bwt_size = bwt_size; v.bwt = ix->bwt.data(); v.sa_intv = 32; v.n_sa = n_sa; v.sa = ix->sa.data();
		if (v.L2[4] != n) { rc = cs_fail_(CS_EDEVICE, "cs_index_build: base counts do not add up"); goto done; }
	}
	*out = ix; ix = nullptr;
done:
	for (void *p : {(void *)d_fwd, (void *)d_T, d_tmp, (void *)d_key, (void *)d_key2, (void *)d_pos, (void *)d_sfx, (void *)d_sfx2, (void *)d_head,
	                (void *)d_rank, (void *)d_flag, (void *)slot, (void *)isfx, (void *)grp, (void *)slot2, (void *)isfx2, (void *)grp2, (void *)isfx_s,
	                (void *)d_prim, (void *)d_words, (void *)d_bwt, (void *)d_cnt, (void *)d_occ, (void *)d_sa}) if (p) (void)hipFree(p);
	if (s) (void)hipStreamDestroy(s);
	delete ix;
	return rc;
}

extern "C" int cs_index_build(const uint8_t *fwd_nt4, uint64_t l_pac, int device, cs_index_t **out)
{
	return cs_index_build_flags(fwd_nt4, l_pac, device, 0u, out);
}
extern "C" int cs_index_build_flags(const uint8_t *fwd_nt4, uint64_t l_pac, int device, uint32_t flags, cs_index_t **out)
{
	if (!fwd_nt4 || !out || l_pac == 0) return cs_fail_(CS_EINVAL, "cs_index_build: bad argument");
	*out = nullptr;
	if (2 * l_pac + 1 >= (1ull << 33)) return cs_fail_(CS_ERANGE, "cs_index_build: genome longer than 2^32 bp");
	for (uint64_t i = 0; i < l_pac; ++i) if (fwd_nt4[i] > 3) return cs_fail_(CS_EINVAL, "cs_index_build: base code > 3 (replace ambiguous bases first, bntseq.c:295)");
	const bool verbose = (flags & CS_BUILD_VERBOSE) != 0;
	if (2 * l_pac + 1 < 0xffffffffull && !(flags & CS_BUILD_FORCE_64BIT)) return build_impl<uint32_t>(fwd_nt4, l_pac, device, verbose, out);
	return build_impl<uint64_t>(fwd_nt4, l_pac, device, verbose, out);
}

// write <prefix>.bwt and <prefix>.sa in the reference's formats (bwt_dump_bwt / bwt_dump_sa, FM_index/bwt.c:385-407)
extern "C" int cs_index_save(const cs_index_t *idx, const char *prefix)
{
	if (!idx || !prefix) return cs_fail_(CS_EINVAL, "cs_index_save: null argument");
	const cs_index_view_t &v = idx->v;
	std::string p(prefix);
	FILE *fp = fopen((p + ".bwt").c_str(), "wb");
	if (!fp) return cs_fail_(CS_EIO, "cannot write " + p + ".bwt");
	bool ok = fwrite(&v.primary, 8, 1, fp) == 1 && fwrite(&v.L2[1], 8, 4, fp) == 4 && fwrite(v.bwt, 4, v.bwt_size, fp) == v.bwt_size;
	ok = (fclose(fp) == 0) && ok;
	if (!ok) return cs_fail_(CS_EIO, "short write on " + p + ".bwt");
	fp = fopen((p + ".sa").c_str(), "wb");
	if (!fp) return cs_fail_(CS_EIO, "cannot write " + p + ".sa");
	ok = fwrite(&v.primary, 8, 1, fp) == 1 && fwrite(&v.L2[1], 8, 4, fp) == 4 && fwrite(&v.sa_intv, 8, 1, fp) == 1 && fwrite(&v.seq_len, 8, 1, fp) == 1 &&
	     fwrite(v.sa + 1, 8, v.n_sa - 1, fp) == v.n_sa - 1;
	ok = (fclose(fp) == 0) && ok;
	if (!ok) return cs_fail_(CS_EIO, "short write on " + p + ".sa");
	return CS_OK;
}
