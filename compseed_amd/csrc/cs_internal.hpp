// cs_internal.hpp -- declarations shared by the translation units of libcompseed_amd.so (not part of the C ABI)
#pragma once
#include "../../include/compseed_amd.h"
#include <algorithm>
#include <string>
#include <vector>

struct cs_index { // host copy of an index: the arrays behind a cs_index_view_t
	cs_index_view_t v;
	std::vector<uint32_t> bwt;
	std::vector<uint64_t> sa;
};

int cs_fail_(int code, const std::string &msg); // records the calling thread's error message, returns code

// contig table of an index (<prefix>.ann, ALT flags from <prefix>.alt): shared by the chainer and the extension driver
struct cs_refseq_view { int64_t l_pac; std::vector<int64_t> offset; std::vector<int32_t> len; std::vector<uint8_t> is_alt; };
int cs_load_contigs_(const char *prefix, cs_refseq_view &ref);
int cs_load_pac_(const char *prefix, int64_t l_pac, std::vector<uint8_t> &pac); // <prefix>.pac: four bases per byte, first base in the top bits (bntseq.c:236-237)
inline uint8_t cs_pac_base_(const std::vector<uint8_t> &pac, int64_t p) { return (uint8_t)((pac[(size_t)(p >> 2)] >> ((~p & 3) << 1)) & 3); }
// ASCII -> code as the reference's table does it (nst_nt4_table, bntseq.c:46-63): ACGT in either case 0..3, '-' 5, everything else 4;
// bytes 0..4 are codes already (comp_seed.cpp:2258-2260 converts only bytes above 4)
inline uint8_t cs_base_code_(uint8_t c)
{
	if (c <= 4) return c;
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

// the chainer (chain.cpp: cs_chain_batch; chain_filter.cpp: cs_chain_filter)
struct cs_chainer {
	cs_refseq_view ref; std::string prefix;
	std::vector<cs_chain_t> chains; std::vector<uint64_t> chain_off, cseed_off; std::vector<cs_seed_t> cseeds;                          // cs_chain_batch's result
	std::vector<uint8_t> pac;                                                                                                            // loaded when cs_chain_filter first needs it
	std::vector<cs_chain_t> f_chains; std::vector<uint64_t> f_chain_off, f_cseed_off; std::vector<cs_seed_t> f_cseeds; std::vector<int32_t> f_score; // cs_chain_filter's
};

// the reads of a part as 16-byte records of 32 bases, made on host threads (host_pack.cpp; bit-identical to pack_reads_kernel's)
void cs_pack_reads_host_(const uint8_t *bases, const uint64_t *offsets, int64_t r0, int64_t n, int64_t lo, int64_t hi, void *rec_out, int threads, int force_scalar);

// the part of an aligner the host-side region passes need (dedup.cpp: mem_sort_dedup_patch)
struct cs_aligner_core { const cs_refseq_view *ref; const std::vector<uint8_t> *pac; const cs_aln_params_t *par; };
int cs_dedup_regions_(const cs_aligner_core &A, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets,
                      std::vector<uint64_t> &out_off, std::vector<cs_alnreg_t> &out_regs, std::vector<int32_t> &out_ncomp);

// the device side of cs_extend_chains (align_gpu.hip): windows, regions, pair lists, result passes and the purge as kernels
struct cs_aligner_gpu;
void cs_aligner_gpu_release_(cs_aligner_gpu *g);
int cs_extend_chains_gpu_(cs_aligner_gpu **gp, int device, cs_extender_t *ext, const cs_refseq_view &R, const std::vector<uint8_t> &pac, const cs_aln_params_t &o,
                          const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases, const uint64_t *read_offsets,
                          std::vector<uint64_t> &reg_off, std::vector<cs_alnreg_t> &regs, cs_aln_stats_t &st);

// ---- the host passes' work sharing: fn(k) for k in [0, n_chunks) on T threads, chunks of reads handed out by a counter.  (A read inside a
// repeat costs a hundred times the average in the quadratic passes -- chain filter, dedup --; equal shares left fifteen threads waiting for
// the one that drew them: cs_chain_filter 107 -> 40 ms per million reads on 16 threads.)
#include <atomic>
#include <thread>
template <class F> inline void cs_for_chunks_(int T, int64_t n_chunks, F fn)
{
	if (T <= 1 || n_chunks <= 1) { for (int64_t k = 0; k < n_chunks; ++k) fn(k); return; }
	std::atomic<int64_t> next{0};
	std::vector<std::thread> th;
	const int nt = (int)std::min<int64_t>(T, n_chunks);
	for (int t = 0; t < nt; ++t) th.emplace_back([&]() { for (int64_t k; (k = next.fetch_add(1)) < n_chunks; ) fn(k); });
	for (auto &t : th) t.join();
}
inline int64_t cs_chunk_reads_(int64_t n, int T) { return std::max<int64_t>(256, std::min<int64_t>(2048, n / ((int64_t)T * 8) + 1)); }

