// The reads as the seeding kernels want them -- 16-byte records of 32 bases (smem_split.hpp, pack_reads_kernel) -- made on the host, so
// that a batch crosses PCIe as 0.6 bytes per base instead of one.  The records are bit-identical to pack_reads_kernel<true>'s:
// .x/.y the bases, 2 bits each, base j in bits 2j..2j+1; .z one bit per base that is ambiguous or behind the end of the read; .w 0;
// record k of read r at (off[r] >> 5) + r + k with off counted from the part's first base; letters through nst_nt4_table's rule
// (FM_index/bntseq.c:46-63; codes 0..3 pass through as in comp_seed.cpp:2259, everything else is ambiguous).
#include "cs_internal.hpp"
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <immintrin.h>

namespace {

struct Rec { uint32_t x, y, z, w; };

inline uint8_t code_of(uint8_t c)
{
	const uint32_t t = (c & 0xdfu) - 0x41u;
	const bool letter = t < 20u && ((0x80045u >> t) & 1u);
	return (uint8_t)(c < 4u ? c : letter ? ((c >> 1) ^ (c >> 2)) & 3u : 4u);
}

// 32 bytes at p (all readable) -> bases (2 bits each) and the mask of ambiguous ones
inline void pack32_scalar(const uint8_t *p, uint64_t &bases, uint32_t &bad)
{
	uint64_t b = 0; uint32_t m = 0;
	for (int j = 0; j < 32; ++j) {
		const uint8_t c = code_of(p[j]);
		if (c > 3) m |= 1u << j; else b |= (uint64_t)c << (2 * j);
	}
	bases = b; bad = m;
}

__attribute__((target("avx2,bmi2"))) inline void pack32_avx2(const uint8_t *p, uint64_t &bases, uint32_t &bad)
{
	const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
	const __m256i up = _mm256_and_si256(v, _mm256_set1_epi8((char)0xdf));
	__m256i letter = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(up, _mm256_set1_epi8(0x41)), _mm256_cmpeq_epi8(up, _mm256_set1_epi8(0x43))),
	                                 _mm256_or_si256(_mm256_cmpeq_epi8(up, _mm256_set1_epi8(0x47)), _mm256_cmpeq_epi8(up, _mm256_set1_epi8(0x54))));
	const __m256i small = _mm256_cmpeq_epi8(_mm256_and_si256(v, _mm256_set1_epi8((char)0xfc)), _mm256_setzero_si256());
	// letters: ((c >> 1) ^ (c >> 2)) & 3 (bits 1..3 of the same byte: the 16-bit shifts bring nothing in from the neighbour below bit 6)
	const __m256i lc = _mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2));
	__m256i code = _mm256_blendv_epi8(v, lc, letter);
	const __m256i valid = _mm256_or_si256(letter, small);
	code = _mm256_and_si256(_mm256_and_si256(code, _mm256_set1_epi8(3)), valid);
	const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(code, 7));
	const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(code, 6));
	bases = _pdep_u64(lo, 0x5555555555555555ull) | _pdep_u64(hi, 0xaaaaaaaaaaaaaaaaull);
	bad = ~(uint32_t)_mm256_movemask_epi8(valid);
}

inline void finish(Rec &o, uint64_t b, uint32_t m, uint64_t left)
{
	if (left < 32) { m |= ~0u << (uint32_t)left; b &= (1ull << (2 * (uint32_t)left)) - 1ull; }
	o.x = (uint32_t)b; o.y = (uint32_t)(b >> 32); o.z = m; o.w = 0;
}

// reads r_lo..r_hi-1 of a part whose first read starts at absolute base `base0`; `end_all` = one past the last readable byte
__attribute__((target("avx2,bmi2"))) void pack_range_v(const uint8_t *bases, const uint64_t *off, int64_t r_lo, int64_t r_hi, uint64_t base0, uint64_t end_all, Rec *rec)
{
	for (int64_t r = r_lo; r < r_hi; ++r) {
		const uint64_t rb = off[r] - base0, re = off[r + 1] - base0, len = re - rb;
		Rec *o = rec + (rb >> 5) + (uint64_t)r;
		const uint64_t nrec = (re >> 5) + (uint64_t)r + 1 - ((rb >> 5) + (uint64_t)r);
		const uint8_t *p = bases + off[r];
		for (uint64_t k = 0; k < nrec; ++k) {
			if (k * 32 >= len) { o[k].x = o[k].y = 0; o[k].z = ~0u; o[k].w = 0; continue; }
			uint64_t b; uint32_t m;
			const uint64_t a = off[r] + k * 32;
			if (a + 32 <= end_all) pack32_avx2(p + k * 32, b, m);
			else { uint8_t t[32]; memset(t, 4, 32); memcpy(t, p + k * 32, (size_t)(end_all - a)); pack32_avx2(t, b, m); }
			finish(o[k], b, m, len - k * 32);
		}
	}
}
void pack_range_s(const uint8_t *bases, const uint64_t *off, int64_t r_lo, int64_t r_hi, uint64_t base0, uint64_t end_all, Rec *rec)
{
	for (int64_t r = r_lo; r < r_hi; ++r) {
		const uint64_t rb = off[r] - base0, re = off[r + 1] - base0, len = re - rb;
		Rec *o = rec + (rb >> 5) + (uint64_t)r;
		const uint64_t nrec = (re >> 5) + (uint64_t)r + 1 - ((rb >> 5) + (uint64_t)r);
		const uint8_t *p = bases + off[r];
		for (uint64_t k = 0; k < nrec; ++k) {
			if (k * 32 >= len) { o[k].x = o[k].y = 0; o[k].z = ~0u; o[k].w = 0; continue; }
			uint64_t b; uint32_t m;
			const uint64_t a = off[r] + k * 32;
			if (a + 32 <= end_all) pack32_scalar(p + k * 32, b, m);
			else { uint8_t t[32]; memset(t, 4, 32); memcpy(t, p + k * 32, (size_t)(end_all - a)); pack32_scalar(t, b, m); }
			finish(o[k], b, m, len - k * 32);
		}
	}
}

} // namespace

// Records of reads [lo, hi) of the part [r0, r0 + n) of a batch (offsets = the batch's absolute offsets, bases = its first byte; lo / hi
// counted from r0) into rec[], which holds the whole part: (nb >> 5) + n records, nb = offsets[r0 + n] - offsets[r0].
void cs_pack_reads_host_(const uint8_t *bases, const uint64_t *offsets, int64_t r0, int64_t n, int64_t lo, int64_t hi, void *rec_out, int threads, int force_scalar)
{
	if (n <= 0 || hi <= lo) return;
	Rec *rec = static_cast<Rec *>(rec_out);
	const uint64_t base0 = offsets[r0], end_all = offsets[r0 + n];
	const uint64_t *off = offsets + r0;
	const bool vec = !force_scalar && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
	auto run = [&](int64_t a, int64_t b) { if (vec) pack_range_v(bases, off, a, b, base0, end_all, rec); else pack_range_s(bases, off, a, b, base0, end_all, rec); };
	const int64_t m = hi - lo;
	threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, m / 4096 + 1));
	if (threads == 1) { run(lo, hi); return; }
	const int64_t chunk = std::max<int64_t>(4096, std::min<int64_t>(65536, m / (threads * 4) + 1));
	std::atomic<int64_t> next{lo};
	std::vector<std::thread> th;
	for (int t = 0; t < threads; ++t)
		th.emplace_back([&] { for (;;) { const int64_t a = next.fetch_add(chunk); if (a >= hi) break; run(a, std::min(hi, a + chunk)); } });
	for (auto &t : th) t.join();
}

extern "C" int cs_pack_reads(const uint8_t *bases, const uint64_t *offsets, int64_t n_reads, void *records, int threads, uint32_t flags)
{
	if (n_reads < 0 || (n_reads > 0 && (!offsets || !records)) || (n_reads > 0 && offsets[n_reads] > 0 && !bases)) return cs_fail_(CS_EINVAL, "cs_pack_reads: bad argument");
	for (int64_t r = 0; r < n_reads; ++r) if (offsets[r + 1] < offsets[r]) return cs_fail_(CS_EINVAL, "cs_pack_reads: offsets must be non-decreasing");
	cs_pack_reads_host_(bases, offsets, 0, n_reads, 0, n_reads, records, threads, (flags & CS_PACK_SCALAR) != 0);
	return CS_OK;
}
