// engine.hip -- C-ABI implementation (include/compseed_amd.h): device residency of the index, batch orchestration.
//
// Host-side counterpart of mem_process_seqs -> seed_and_extend (mapping/comp_seed.cpp:2527, 2242) for the seeding
// and SAL blocks only.  No CPU fallback exists: without a HIP device every entry point fails with CS_EDEVICE.
#include "cs_internal.hpp"
#include "seed_kernels.hpp"
#include "smem_split.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <shared_mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

using namespace csd;

static const int g_lep_lds = 20; // LEP entries per lane kept in LDS by the fused kernel (13 and 10 were measured: slower)
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
int cs_fail_(int code, const std::string &msg) { return fail(code, msg); }

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e__ = (expr);                                                                   \
		if (e__ != hipSuccess) {                                                                   \
			(void)hipGetLastError();                                                               \
			return fail(e__ == hipErrorOutOfMemory ? CS_ENOMEM : CS_EDEVICE,                       \
			            std::string(#expr) + ": " + hipGetErrorString(e__));                       \
		}                                                                                          \
	} while (0)
#define CS_TRY(expr) do { int rc__ = (expr); if (rc__ != CS_OK) return rc__; } while (0)

extern "C" const char *cs_last_error(void) { return g_err.c_str(); }
extern "C" const char *cs_version(void) { return "compseed_amd 0.1 (gfx950)"; }

extern "C" void cs_params_default(cs_params_t *p)
{
	if (!p) return;
	p->min_seed_len = 19; p->split_factor = 1.5f; p->split_width = 10; p->max_occ = 500; p->max_mem_intv = 20;
	p->want_sal = 1; p->sst_mode = 1; p->disable = 0; p->count_traffic = 0;
}
extern "C" void cs_engine_options_default(cs_engine_options_t *o)
{
	if (!o) return;
	memset(o, 0, sizeof *o);
	o->full_sa = 1; o->sa64 = 0; o->text_mode = 1; o->text_arrays = 1; o->jump_k = 15; o->kmer_filter = 1; o->fused = 0;
	o->mem_cap = 64; o->lep_arena_mb = 16384; o->max_raw_mb = 24576; o->r3_text_iter = 5; o->count_sal_merged = 0; o->verbose = 0;
	o->pipeline_reads = 5000000; o->expand_threads = 16; o->host_pack_threads = 8; o->passes_in_flight = 2;
}

// ------------------------------------------------------------------------------------------------ grow-only buffers
template <typename T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;
	int reserve(size_t n, bool keep = false, hipStream_t s = nullptr, size_t keep_n = 0)
	{
		if (n <= cap) return CS_OK;
		size_t want = std::max(n, cap + cap / 2);
		T *q = nullptr;
		HIP_TRY(hipMalloc((void **)&q, want * sizeof(T)));
		if (keep && p && keep_n) {
			hipError_t e = hipMemcpyAsync(q, p, keep_n * sizeof(T), hipMemcpyDeviceToDevice, s);
			if (e == hipSuccess) e = hipStreamSynchronize(s);
			if (e != hipSuccess) { (void)hipFree(q); return fail(CS_EDEVICE, hipGetErrorString(e)); }
		}
		if (p) (void)hipFree(p);
		p = q; cap = want;
		return CS_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
// grow-only plain host memory (the expanded results of cs_engine_seed_batch; `keep_n` elements survive a reallocation)
template <typename T> struct HostBuf {
	T *p = nullptr; size_t cap = 0;
	int reserve(size_t n, size_t keep_n = 0)
	{
		if (n <= cap) return 0;
		size_t want = std::max(n, cap + cap / 4);
		T *q = (T *)malloc(want * sizeof(T));
		if (!q) return 1;
		if (p && keep_n) memcpy(q, p, keep_n * sizeof(T));
		free(p);
		p = q; cap = want;
		return 0;
	}
	void release() { free(p); p = nullptr; cap = 0; }
};
template <typename T> struct PinBuf {
	T *p = nullptr, *dp = nullptr; size_t cap = 0; // dp: the same memory as the device addresses it (kernels may store into it)
	int reserve(size_t n, bool keep = false, size_t keep_n = 0)
	{
		if (n <= cap) return CS_OK;
		size_t want = std::max(n, cap + cap / 2);
		T *q = nullptr;
		HIP_TRY(hipHostMalloc((void **)&q, want * sizeof(T), hipHostMallocDefault));
		if (keep && p && keep_n) memcpy(q, p, keep_n * sizeof(T));
		if (p) (void)hipHostFree(p);
		p = q; cap = want; dp = nullptr;
		void *d = nullptr;
		if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) dp = (T *)d; else (void)hipGetLastError();
		return CS_OK;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; }
};

constexpr int PIPE_DEPTH = 4; // batches in flight in the host pipeline (cs_engine_submit): one pinned result slot each
struct cs_engine {
	int device = 0;
	int n_cu = 256;
	cs_engine_options_t opt{};
	hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, stream4 = nullptr; // stream2: round 3 (low priority); stream3: calls without LEPs; stream4: wide sweeps
	hipEvent_t ev_r3a = nullptr, ev_r3b = nullptr, ev_wa = nullptr, ev_wb = nullptr, ev_wc = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	DevIndex ix{};
	DevBuf<uint4> d_bwt; DevBuf<uint64_t> d_sa;
	DevBuf<uint32_t> d_fsa32; DevBuf<uint64_t> d_fsa64; // full suffix array (one of the two)
	DevBuf<uint32_t> d_text2, d_isa32; DevBuf<uint64_t> d_isa64; // text mode: 2-bit text + inverse suffix array
	DevBuf<uint64_t> d_bloom; int bloom_k = 0; uint32_t bloom_bits = 0; // k-mer filter of the text for the min_seed_len in use (built on first use)
	DevBuf<uint8_t> d_pending; // r3text_kernel: reads with calls of rounds 1/2 still queued when it starts
	DevBuf<uint32_t> d_cnt_snap; DevBuf<uint8_t> d_lcp, d_rep; DevBuf<uint64_t> d_auxA, d_auxB; // re-seeding from the text: capped LCP by row, repeat length by position
	// inputs
	DevBuf<uint8_t> d_raw, d_seq; DevBuf<uint64_t> d_off; DevBuf<uint4> d_seqp; const uint4 *seqp_cur = nullptr; const uint64_t *off_base = nullptr; // d_seqp: pack_reads_kernel's records for the batch whose offsets start at off_base
	// SMEM stage
	DevBuf<OutMem> d_out, d_out2; DevBuf<uint32_t> d_cnt, d_cnt2, d_ovf; DevBuf<uint4> d_spill;
	DevBuf<unsigned long long> d_ctr; // [0] task counter, [1] queries, [2] overflow count, [3] max len
	DevBuf<uint8_t> d_tmp, d_tmp2;
	// results (device)
	DevBuf<uint64_t> d_mem_off, d_seed_off, d_seed_of_mem; DevBuf<OutMem> d_mems; DevBuf<uint64_t> d_salcnt; DevBuf<OutSeed> d_seeds; // d_salcnt: SA slots per mem, written by the sort that makes d_mems
	// results (pinned host)
	PinBuf<uint64_t> h_mem_off, h_seed_off; PinBuf<OutMem> h_mems; PinBuf<OutSeed> h_seeds;
	PinBuf<unsigned long long> h_ctr;
	// split (forward / cooperative backward) SMEM path
	DevBuf<uint64_t> d_fqA, d_fqB, d_fqR; DevBuf<uint4> d_sst2, d_jump; int jump_k = 0; DevBuf<BTask> d_bq; DevBuf<uint4> d_lep; DevBuf<OvfRec> d_ovfrec;
	DevBuf<uint32_t> d_okey, d_oidx, d_okey2, d_oidx2; DevBuf<uint64_t> d_okey64, d_okey64b; DevBuf<unsigned long long> d_sctr; PinBuf<unsigned long long> h_sctr;
	int smem_mode = 1;          // 1 = split kernels (default), 0 = fused one-lane-per-read kernel (CS_SMEM_MODE=fused)
	int occ_win = 5; // ... of bwd_win_kernel
	int occ_fwd = 4, occ_bwd = 4; // resident 256-thread blocks per CU of fwd_kernel / bwd_kernel
	size_t lep_arena_bytes = (size_t)32 << 30;
	// host variants (seed_host_pipelined): copy streams, three input slots, two pack slots, pinned packed results, expanded results
	hipStream_t s_up = nullptr, s_down = nullptr; hipEvent_t hp_ev_pk[2] = {nullptr, nullptr}, hp_ev_dn[4] = {nullptr, nullptr, nullptr, nullptr}, hp_ev_done[PIPE_DEPTH] = {};
	PinBuf<uint4> hp_stage[3]; // records made by the host (host_pack.cpp), staged for the upload into hp_in[slot]
	DevBuf<uint8_t> hp_in[3], hp_pk_mems[2]; DevBuf<uint64_t> hp_inoff[3], hp_pk_moff[2], hp_pk_soff[2]; DevBuf<uint32_t> hp_pk_rlo[2]; DevBuf<uint8_t> hp_pk_rhi[2]; // seeds: low words and fifth bytes of rbeg
	PinBuf<uint64_t> hp_moff[PIPE_DEPTH], hp_soff[PIPE_DEPTH]; PinBuf<uint8_t> hp_mems[PIPE_DEPTH]; PinBuf<uint32_t> hp_rlo[PIPE_DEPTH]; PinBuf<uint8_t> hp_rhi[PIPE_DEPTH]; // pinned result slots (slot = batch % PIPE_DEPTH)
	struct HostPipe *hp = nullptr;
	HostBuf<cs_intv_t> x_mems; HostBuf<cs_seed_t> x_seeds;
	cs_stats_t st{};
	DevBuf<unsigned long long> d_evc; uint64_t stream_bytes = 0; // byte model: event counters [N_KID][N_EV] on the device, stream part on the host
	struct { bool valid = false; int64_t n_reads = 0; uint64_t n_mems = 0, n_seeds = 0; int want_sal = 0; } last; // the result held in d_mems / d_seeds
	DevBuf<uint64_t> d_sel, d_sel_moff, d_sel_soff; DevBuf<OutMem> d_sel_mems; DevBuf<OutSeed> d_sel_seeds;
	uint32_t cap = 64;          // mems per read kept by the first pass
	size_t max_raw_bytes = (size_t)24 << 30;
	int blocks_per_cu = 2;
	// Two seeding passes in flight.  The tail of a pass (late iterations with a few thousand calls each, the sort, SAL, ten host round
	// trips) leaves most of the GPU idle, and a small part of a batch is nearly all tail; a second pass fills it.  The second pass
	// context is a second cs_engine (`twin`) with streams, events and every working buffer of its own and the index arrays of this one
	// (ix, jump table, k-mer filter: aliases, never freed by the twin); everything that runs a pass takes "the engine it runs on".
	cs_engine *twin = nullptr, *owner = nullptr;   // owner: set in the twin
	cs_engine *last_ctx = nullptr;                 // which of the two holds the last whole-batch result (`last` lives in that one)
	std::shared_mutex filter_rw;                   // passes hold it shared; rebuilding the k-mer filter for another min_seed_len takes it exclusively
	int bloom_tried_k = 0;                         // last min_seed_len the filter was (re)built or found not to fit for
	bool in_shared_pass = false;                   // a pass under filter_rw: build_kmer_filter is not to touch the filter
	struct DevPipe *dp = nullptr;                  // cs_engine_submit_device / cs_engine_collect_device
};

// ------------------------------------------------------------------------------------------------ index files
static bool read_file(const std::string &fn, std::vector<uint8_t> &buf)
{
	FILE *fp = fopen(fn.c_str(), "rb");
	if (!fp) return false;
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	buf.resize((size_t)sz);
	size_t got = sz ? fread(buf.data(), 1, (size_t)sz, fp) : 0;
	fclose(fp);
	return got == (size_t)sz;
}

extern "C" int cs_index_load(const char *prefix, cs_index_t **out)
{
	if (!prefix || !out) return fail(CS_EINVAL, "cs_index_load: null argument");
	*out = nullptr;
	std::string pre(prefix);
	{ // bwa_idx_infer_prefix (bwalib/bwa.c:244-268): accept "<hint>.64" if it exists
		FILE *fp = fopen((pre + ".64.bwt").c_str(), "rb");
		if (fp) { fclose(fp); pre += ".64"; }
	}
	std::vector<uint8_t> raw;
	if (!read_file(pre + ".bwt", raw) || raw.size() < 40 + 64) return fail(CS_EIO, "cannot read " + pre + ".bwt");
	cs_index *ix = new cs_index();
	cs_index_view_t &v = ix->v;
	memset(&v, 0, sizeof v);
	memcpy(&v.primary, raw.data(), 8);           // bwt_restore_bwt, bwt.c:443-462
	memcpy(&v.L2[1], raw.data() + 8, 32);
	v.L2[0] = 0; v.seq_len = v.L2[4];
	v.bwt_size = (raw.size() - 40) >> 2;
	ix->bwt.resize(v.bwt_size);
	memcpy(ix->bwt.data(), raw.data() + 40, v.bwt_size * 4);
	if (!read_file(pre + ".sa", raw) || raw.size() < 56) { delete ix; return fail(CS_EIO, "cannot read " + pre + ".sa"); }
	uint64_t h[7];
	memcpy(h, raw.data(), 56);                   // bwt_restore_sa, bwt.c:421-441
	if (h[0] != v.primary) { delete ix; return fail(CS_EIO, "SA-BWT inconsistency: primary is not the same"); }
	if (h[6] != v.seq_len) { delete ix; return fail(CS_EIO, "SA-BWT inconsistency: seq_len is not the same"); }
	v.sa_intv = h[5];
	if (v.sa_intv == 0 || (v.sa_intv & (v.sa_intv - 1))) { delete ix; return fail(CS_EIO, "SA sample interval is not a power of 2"); }
	v.n_sa = (v.seq_len + v.sa_intv) / v.sa_intv;
	if ((raw.size() - 56) / 8 < v.n_sa - 1) { delete ix; return fail(CS_EIO, pre + ".sa is truncated"); }
	ix->sa.resize(v.n_sa);
	ix->sa[0] = ~0ull;
	memcpy(ix->sa.data() + 1, raw.data() + 56, (v.n_sa - 1) * 8);
	v.bwt = ix->bwt.data(); v.sa = ix->sa.data();
	*out = ix;
	return CS_OK;
}
extern "C" int cs_index_view(const cs_index_t *idx, cs_index_view_t *view)
{
	if (!idx || !view) return fail(CS_EINVAL, "cs_index_view: null argument");
	*view = idx->v;
	return CS_OK;
}
extern "C" void cs_index_free(cs_index_t *idx) { delete idx; }

// ------------------------------------------------------------------------------------------------ engine
extern "C" int cs_device_count(int *n)
{
	if (!n) return fail(CS_EINVAL, "null argument");
	*n = 0;
	HIP_TRY(hipGetDeviceCount(n));
	return CS_OK;
}

static int build_kmer_filter(cs_engine *e, int k);
// streams and events of one pass context (the engine itself, or its twin)
static int create_pass_streams(cs_engine *e)
{
	HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
	for (auto &ev : e->ev) HIP_TRY(hipEventCreate(&ev));
	int lo = 0, hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = least urgent
	HIP_TRY(hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, lo));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_r3a, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_r3b, hipEventDisableTiming));
	HIP_TRY(hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_wa, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_wb, hipEventDisableTiming));
	HIP_TRY(hipStreamCreateWithFlags(&e->stream4, hipStreamNonBlocking));
	HIP_TRY(hipEventCreateWithFlags(&e->ev_wc, hipEventDisableTiming));
	return CS_OK;
}
static int engine_init(cs_engine *e, const cs_index_view_t *v)
{
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if (ndev <= 0) return fail(CS_EDEVICE, "no HIP device: the seeding engine has no CPU path");
	if (e->device < 0 || e->device >= ndev) return fail(CS_EINVAL, "device ordinal out of range");
	HIP_TRY(hipSetDevice(e->device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, e->device));
	e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	const cs_engine_options_t &opt = e->opt;
	const bool verbose = opt.verbose != 0;
	if (opt.mem_cap < 1 || opt.mem_cap > 4096 || opt.lep_arena_mb < 1 || opt.max_raw_mb < 1 || (opt.jump_k != 0 && (opt.jump_k < 6 || opt.jump_k > 15)))
		return fail(CS_EINVAL, "cs_engine_options_t: mem_cap 1..4096, lep_arena_mb >= 1, max_raw_mb >= 1, jump_k 0 or 6..15");
	for (int r : opt.reserved) if (r) return fail(CS_EINVAL, "cs_engine_options_t.reserved must be 0");
	if (opt.passes_in_flight < 1 || opt.passes_in_flight > 2 || opt.host_pack_threads < 0) return fail(CS_EINVAL, "cs_engine_options_t: passes_in_flight 1 or 2, host_pack_threads >= 0");
	if (verbose) { fprintf(stderr, "[cs_engine] creating engine on device %d, seq_len %llu\n", e->device, (unsigned long long)v->seq_len); fflush(stderr); }
	CS_TRY(create_pass_streams(e));
	{
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = least urgent
		// The runtime multiplexes the normal-priority streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
		// default), and a stream that shares its queue with a 30-ms download stands still for 30 ms -- measured: the seeding kernels
		// of a sub-batch took 65 instead of 45 ms beside the download of the previous one.  So the engine keeps to three normal
		// streams (main and two side streams; with the process's default stream that makes four); round 3 runs at low priority and
		// the two copy streams at high priority, which have queues of their own.
		HIP_TRY(hipStreamCreateWithPriority(&e->s_up, hipStreamNonBlocking, hi));
		HIP_TRY(hipStreamCreateWithPriority(&e->s_down, hipStreamNonBlocking, hi));
		for (auto &ev : e->hp_ev_pk) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		for (auto &ev : e->hp_ev_dn) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		for (auto &ev : e->hp_ev_done) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	}

	if (v->seq_len == 0 || v->seq_len != v->L2[4] || v->L2[0] != 0) return fail(CS_EINVAL, "index view: L2 / seq_len inconsistent");
	if (v->seq_len >> 37) return fail(CS_ERANGE, "index longer than 2^37 symbols does not fit the packed LEP entries");
	if (v->primary > v->seq_len) return fail(CS_EINVAL, "index view: primary out of range");
	uint64_t n_blocks = (v->seq_len + 127) >> 7;
	// the file holds one extra count record after the last block (bwt_bwtupdate_core, index_main.c:152-174)
	if (v->bwt_size < n_blocks * 16) return fail(CS_EINVAL, "index view: bwt array shorter than seq_len requires");
	if (v->sa_intv == 0 || (v->sa_intv & (v->sa_intv - 1))) return fail(CS_EINVAL, "index view: sa_intv is not a power of two");
	if (v->n_sa != (v->seq_len + v->sa_intv) / v->sa_intv) return fail(CS_EINVAL, "index view: n_sa inconsistent");

	size_t quads = (size_t)((v->bwt_size + 3) >> 2) + 8; // pad: a block load never leaves the allocation
	CS_TRY(e->d_bwt.reserve(quads));
	HIP_TRY(hipMemsetAsync(e->d_bwt.p, 0, quads * sizeof(uint4), e->stream));
	HIP_TRY(hipMemcpyAsync(e->d_bwt.p, v->bwt, (size_t)v->bwt_size * 4, hipMemcpyHostToDevice, e->stream));
	CS_TRY(e->d_sa.reserve((size_t)v->n_sa));
	HIP_TRY(hipMemcpyAsync(e->d_sa.p, v->sa, (size_t)v->n_sa * 8, hipMemcpyHostToDevice, e->stream));
	// one-time conversion of the 2-bit packed bases of every block into bit planes (fm_device.hpp)
	CS_TRY(e->d_sctr.reserve(32));
	CS_TRY(e->h_sctr.reserve(32));
	CS_TRY(e->d_sst2.reserve(SST2_ENTRIES));
	HIP_TRY(hipMemsetAsync(e->d_sst2.p, 0xff, SST2_ENTRIES * sizeof(uint4), e->stream)); // empty second-level SST
	HIP_TRY(hipMemsetAsync(e->d_sctr.p, 0, 32 * sizeof(unsigned long long), e->stream));
	hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, e->stream, e->d_bwt.p, n_blocks, e->d_sctr.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_sctr.p, e->d_sctr.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (e->h_sctr.p[0]) return fail(CS_ERANGE, "a single base occurs 2^32 times or more: 32-bit Occ counts of the device layout overflow");
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fwd_kernel<256, false>, 256, 0) == hipSuccess && nb > 0) e->occ_fwd = std::min(nb, 8);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bwd_all_kernel<256, false>, 256, 0) == hipSuccess && nb > 0) e->occ_bwd = std::min(nb, 8);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bwd_win_kernel<256, false>, 256, 0) == hipSuccess && nb > 0) e->occ_win = std::min(nb, 8);
		(void)hipGetLastError();
	}
	e->smem_mode = opt.fused ? 0 : 1;
	e->lep_arena_bytes = (size_t)opt.lep_arena_mb << 20;
	e->cap = (uint32_t)opt.mem_cap;
	e->max_raw_bytes = (size_t)opt.max_raw_mb << 20;
	CS_TRY(e->d_ctr.reserve(8));
	CS_TRY(e->h_ctr.reserve(8));
	HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, 8 * sizeof(unsigned long long), e->stream));
	static_assert(N_KID == CS_N_KERNELS && N_EV == CS_N_EVENTS, "cs_traffic_t mirrors the device-side event table");
	CS_TRY(e->d_evc.reserve((size_t)N_KID * N_EV));
	HIP_TRY(hipMemsetAsync(e->d_evc.p, 0, (size_t)N_KID * N_EV * sizeof(unsigned long long), e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));

	DevIndex &ix = e->ix;
	ix.bwt = e->d_bwt.p; ix.sa = e->d_sa.p;
	ix.primary = v->primary; ix.seq_len = v->seq_len; ix.n_sa = v->n_sa; ix.n_blocks = n_blocks;
	for (int i = 0; i < 5; ++i) ix.L2[i] = v->L2[i];
	ix.sa_mask = (uint32_t)(v->sa_intv - 1);
	ix.sa_shift = (uint32_t)__builtin_ctzll(v->sa_intv);

	if (verbose) { fprintf(stderr, "[cs_engine] index uploaded and re-laid out\n"); fflush(stderr); }
	// full suffix array in HBM (4 B/row below 2^32 rows, else 8 B/row): 50 GB for hg19 of the 288 GB on board
	ix.fsa32 = nullptr; ix.fsa64 = nullptr;
	if (opt.full_sa) {
		uint64_t rows = v->seq_len + 1;
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		bool small = rows < 0xffffffffull && !opt.sa64; // sa64: 8-byte entries on a small index (tests of the hg19-scale instantiation)
		size_t need = (size_t)rows * (small ? 4 : 8);
		if (need + ((size_t)8 << 30) < free_b) {
			unsigned grid = (unsigned)((v->n_sa + 255) / 256);
			if (small) {
				CS_TRY(e->d_fsa32.reserve((size_t)rows + 16));
				hipLaunchKernelGGL(sa_fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, e->stream, ix, e->d_fsa32.p);
				HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
				ix.fsa32 = e->d_fsa32.p;
			} else {
				CS_TRY(e->d_fsa64.reserve((size_t)rows + 16));
				hipLaunchKernelGGL(sa_fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, e->stream, ix, e->d_fsa64.p);
				HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
				ix.fsa64 = e->d_fsa64.p;
			}
		}
	}

	// text mode (smem_split.hpp): the 2-bit text and the inverse suffix array, derived from the full suffix array
	ix.text2 = nullptr; ix.isa32 = nullptr; ix.isa64 = nullptr;
	{
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		uint64_t rows = v->seq_len + 1;
		size_t need = (size_t)rows * (ix.fsa32 ? 4 : 8) + (size_t)v->seq_len + (size_t)v->seq_len / 4 + ((size_t)24 << 30);
		if (opt.text_mode && (ix.fsa32 || ix.fsa64) && need < free_b) {
			DevBuf<uint8_t> tbytes;
			CS_TRY(tbytes.reserve((size_t)v->seq_len + 64));
			CS_TRY(e->d_text2.reserve((size_t)((v->seq_len + 15) >> 4) + 16));
			unsigned grid = (unsigned)std::min<uint64_t>((rows + 255) / 256, 1u << 22);
			if (ix.fsa32) {
				CS_TRY(e->d_isa32.reserve((size_t)rows + 16));
				hipLaunchKernelGGL(text_isa_fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa32, tbytes.p, e->d_isa32.p);
			} else {
				CS_TRY(e->d_isa64.reserve((size_t)rows + 16));
				hipLaunchKernelGGL(text_isa_fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa64, tbytes.p, e->d_isa64.p);
			}
			hipLaunchKernelGGL(text_pack_kernel, dim3(grid), dim3(256), 0, e->stream, tbytes.p, v->seq_len, e->d_text2.p);
			HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
			tbytes.release();
			ix.text2 = e->d_text2.p; ix.isa32 = e->d_isa32.p; ix.isa64 = e->d_isa64.p;
		}
		if (verbose) { fprintf(stderr, "[cs_engine] text mode: %s\n", ix.text2 ? "on" : "off"); fflush(stderr); }
	}
	// re-seeding from the text (smem_split.hpp, r2text_kernel): capped LCP array and repeat-length array, 1 byte per row each
	ix.lcp = nullptr; ix.rep = nullptr;
	{
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		size_t need = (size_t)v->seq_len * 2 + ((size_t)24 << 30);
		if (opt.text_arrays && ix.text2 && need < free_b) {
			uint64_t rows = v->seq_len + 1;
			CS_TRY(e->d_lcp.reserve((size_t)rows + 64)); CS_TRY(e->d_rep.reserve((size_t)rows + 64));
			unsigned grid = (unsigned)std::min<uint64_t>((rows + 256) / 256, 1u << 22);
			if (ix.fsa32) {
				hipLaunchKernelGGL(lcp_fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa32, e->d_lcp.p);
				hipLaunchKernelGGL(rep_fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa32, (const uint8_t *)e->d_lcp.p, e->d_rep.p);
			} else {
				hipLaunchKernelGGL(lcp_fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa64, e->d_lcp.p);
				hipLaunchKernelGGL(rep_fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, e->stream, ix, ix.fsa64, (const uint8_t *)e->d_lcp.p, e->d_rep.p);
			}
			HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
			ix.lcp = e->d_lcp.p; ix.rep = e->d_rep.p;
		}
		if (verbose) {
			HIP_TRY(hipMemGetInfo(&free_b, &total_b));
			fprintf(stderr, "[cs_engine] re-seeding from the text: %s; device memory free %.1f of %.1f GB\n", ix.rep ? "on" : "off", free_b / 1e9, total_b / 1e9); fflush(stderr);
		}
	}
	// round-3 jump table (smem_split.hpp): every 15-mer, 17 GB (13: 1 GB, measured 2 % slower); CS_JUMP_K = 0 disables
	{
		const int jk = opt.jump_k;
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		if (jk >= 6 && jk <= 15 && ((size_t)16 << (2 * jk)) + ((size_t)8 << 30) < free_b) {
			CS_TRY(e->d_jump.reserve((size_t)1 << (2 * jk)));
			hipLaunchKernelGGL(jump_fill_kernel, dim3((unsigned)(e->n_cu * 32)), dim3(256), 0, e->stream, ix, jk, e->d_jump.p);
			HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
			e->jump_k = jk;
		}
	}
	if (e->jump_k) { // the k-mer filter of the window lanes for the default min_seed_len (mem_opt_init: 19); other values on first use
		cs_params_t dp; cs_params_default(&dp);
		CS_TRY(build_kmer_filter(e, dp.min_seed_len));
		if (verbose) { fprintf(stderr, "[cs_engine] k-mer filter: %s\n", e->bloom_k ? "on" : "off"); fflush(stderr); }
	}
	if (verbose) { fprintf(stderr, "[cs_engine] full suffix array: %s\n", ix.fsa32 ? "4-byte" : ix.fsa64 ? "8-byte" : "off"); fflush(stderr); }
	return CS_OK;
}

// The second pass context of an engine (see cs_engine::twin): made on the first call that can use two passes at a time.
static int twin_create(cs_engine *e)
{
	if (e->twin || e->owner || e->opt.passes_in_flight < 2) return CS_OK;
	cs_engine *t = new cs_engine();
	e->twin = t;                                  // (destroyed with e, also when the rest of this function fails)
	t->owner = e; t->device = e->device; t->n_cu = e->n_cu; t->opt = e->opt;
	CS_TRY(create_pass_streams(t));
	t->ix = e->ix;
	t->jump_k = e->jump_k; t->d_jump.p = e->d_jump.p;                 // aliases (cap stays 0): cs_engine_destroy clears them before the release
	t->d_bloom.p = e->d_bloom.p; t->bloom_k = e->bloom_k; t->bloom_bits = e->bloom_bits;
	t->smem_mode = e->smem_mode; t->occ_win = e->occ_win; t->occ_fwd = e->occ_fwd; t->occ_bwd = e->occ_bwd;
	t->lep_arena_bytes = e->lep_arena_bytes; t->cap = e->cap; t->max_raw_bytes = e->max_raw_bytes; t->blocks_per_cu = e->blocks_per_cu;
	CS_TRY(t->d_sctr.reserve(32)); CS_TRY(t->h_sctr.reserve(32));
	CS_TRY(t->d_sst2.reserve(SST2_ENTRIES));
	HIP_TRY(hipMemsetAsync(t->d_sst2.p, 0xff, SST2_ENTRIES * sizeof(uint4), t->stream));
	HIP_TRY(hipMemsetAsync(t->d_sctr.p, 0, 32 * sizeof(unsigned long long), t->stream));
	CS_TRY(t->d_ctr.reserve(8)); CS_TRY(t->h_ctr.reserve(8));
	HIP_TRY(hipMemsetAsync(t->d_ctr.p, 0, 8 * sizeof(unsigned long long), t->stream));
	CS_TRY(t->d_evc.reserve((size_t)N_KID * N_EV));
	HIP_TRY(hipMemsetAsync(t->d_evc.p, 0, (size_t)N_KID * N_EV * sizeof(unsigned long long), t->stream));
	HIP_TRY(hipStreamSynchronize(t->stream));
	if (e->opt.verbose) { fprintf(stderr, "[cs_engine] second pass context created\n"); fflush(stderr); }
	return CS_OK;
}
static void invalidate_last(cs_engine *e) { e->last.valid = false; if (e->twin) e->twin->last.valid = false; e->last_ctx = nullptr; }
static cs_engine *pass_ctx(cs_engine *e, int ci) { return ci && e->twin ? e->twin : e; }
static int n_pass_ctx(const cs_engine *e) { return e->twin ? 2 : 1; }

// One seeding pass on context c of engine P (c == P or c == P->twin).  The k-mer filter of the window lanes belongs to P and is built
// for one min_seed_len at a time: a pass holds filter_rw shared; a pass that wants the filter for another value waits for the others
// to end, rebuilds it alone and starts over.  (If it cannot be had -- no room, another k in use by a pass of the blocking kind -- the
// window lanes do without: results never depend on it.)
static int seed_device_impl(cs_engine *e, const cs_params_t *par, int64_t n_reads, const uint8_t *d_bases, const uint64_t *d_off,
                            uint64_t n_bases, uint64_t *n_mems_out, uint64_t *n_seeds_out, const uint4 *d_recs);
static int pass_on_ctx(cs_engine *P, cs_engine *c, const cs_params_t *par, int64_t n_reads, const uint8_t *d_bases, const uint64_t *d_off,
                       uint64_t n_bases, uint64_t *nm, uint64_t *ns, const uint4 *d_recs)
{
	const int k = par->min_seed_len;
	auto wants_build = [&]() { return P->jump_k && P->ix.text2 && P->opt.kmer_filter && P->smem_mode == 1 && par->sst_mode != 0 && k >= 8 && k <= 24 && P->bloom_k != k && P->bloom_tried_k != k; };
	for (;;) {
		{
			std::shared_lock<std::shared_mutex> sl(P->filter_rw);
			if (!wants_build()) {
				if (c != P) { c->d_bloom.p = P->d_bloom.p; c->bloom_k = P->bloom_k; c->bloom_bits = P->bloom_bits; }
				c->in_shared_pass = true;
				const int rc = seed_device_impl(c, par, n_reads, d_bases, d_off, n_bases, nm, ns, d_recs);
				c->in_shared_pass = false;
				return rc;
			}
		}
		std::unique_lock<std::shared_mutex> ul(P->filter_rw);
		if (wants_build()) { P->bloom_tried_k = k; CS_TRY(build_kmer_filter(P, k)); }
	}
}

extern "C" int cs_engine_create(const cs_index_view_t *index, int device, cs_engine_t **out)
{
	return cs_engine_create_opts(index, device, nullptr, out);
}
extern "C" int cs_engine_create_opts(const cs_index_view_t *index, int device, const cs_engine_options_t *opts, cs_engine_t **out)
{
	if (!index || !out || !index->bwt || !index->sa) return fail(CS_EINVAL, "cs_engine_create: null argument");
	*out = nullptr;
	cs_engine *e = new cs_engine();
	e->device = device;
	if (opts) e->opt = *opts; else cs_engine_options_default(&e->opt);
	int rc = engine_init(e, index);
	if (rc != CS_OK) { std::string keep = g_err; cs_engine_destroy(e); g_err = keep; return rc; }
	*out = e;
	return CS_OK;
}

static void pipe_stop(cs_engine *e);
static void dev_pipe_stop(cs_engine *e);
static bool pipe_busy(const cs_engine *e);
extern "C" void cs_engine_destroy(cs_engine_t *e)
{
	if (!e) return;
	(void)hipSetDevice(e->device);
	pipe_stop(e);
	dev_pipe_stop(e);
	if (e->stream) (void)hipStreamSynchronize(e->stream);
	if (e->twin) { e->twin->d_jump.p = nullptr; e->twin->d_bloom.p = nullptr; cs_engine_destroy(e->twin); e->twin = nullptr; } // (its index arrays are this engine's)
	e->d_bwt.release(); e->d_sa.release(); e->d_fsa32.release(); e->d_fsa64.release(); e->d_text2.release(); e->d_isa32.release(); e->d_isa64.release(); e->d_bloom.release(); e->d_cnt_snap.release(); e->d_pending.release(); e->d_lcp.release(); e->d_rep.release(); e->d_auxA.release(); e->d_auxB.release(); e->d_raw.release(); e->d_seq.release(); e->d_seqp.release(); e->d_off.release();
	e->d_out.release(); e->d_out2.release(); e->d_cnt.release(); e->d_cnt2.release(); e->d_ovf.release(); e->d_spill.release();
	e->d_ctr.release(); e->d_tmp.release(); e->d_tmp2.release(); e->d_mem_off.release(); e->d_seed_off.release(); e->d_seed_of_mem.release();
	e->d_mems.release(); e->d_salcnt.release(); e->d_seeds.release();
	e->d_fqA.release(); e->d_fqB.release(); e->d_fqR.release(); e->d_sst2.release(); e->d_jump.release(); e->d_bq.release(); e->d_lep.release(); e->d_ovfrec.release();
	e->d_evc.release(); e->d_sel.release(); e->d_sel_moff.release(); e->d_sel_soff.release(); e->d_sel_mems.release(); e->d_sel_seeds.release();
	e->d_okey64.release(); e->d_okey64b.release(); e->d_okey.release(); e->d_oidx.release(); e->d_okey2.release(); e->d_oidx2.release(); e->d_sctr.release(); e->h_sctr.release();
	e->h_mem_off.release(); e->h_seed_off.release(); e->h_mems.release(); e->h_seeds.release(); e->h_ctr.release();
	for (int k = 0; k < 3; ++k) { e->hp_stage[k].release(); e->hp_in[k].release(); e->hp_inoff[k].release(); }
	for (int k = 0; k < 2; ++k) { e->hp_pk_mems[k].release(); e->hp_pk_moff[k].release(); e->hp_pk_soff[k].release(); e->hp_pk_rlo[k].release(); e->hp_pk_rhi[k].release(); }
	for (int k = 0; k < PIPE_DEPTH; ++k) { e->hp_moff[k].release(); e->hp_soff[k].release(); e->hp_mems[k].release(); e->hp_rlo[k].release(); e->hp_rhi[k].release(); }
	e->x_mems.release(); e->x_seeds.release();
	for (auto &ev : e->hp_ev_done) if (ev) (void)hipEventDestroy(ev);
	for (auto &ev : e->hp_ev_pk) if (ev) (void)hipEventDestroy(ev);
	for (auto &ev : e->hp_ev_dn) if (ev) (void)hipEventDestroy(ev);
	if (e->s_up) (void)hipStreamDestroy(e->s_up);
	if (e->s_down) (void)hipStreamDestroy(e->s_down);
	for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
	if (e->ev_r3a) (void)hipEventDestroy(e->ev_r3a);
	if (e->ev_r3b) (void)hipEventDestroy(e->ev_r3b);
	if (e->ev_wa) (void)hipEventDestroy(e->ev_wa);
	if (e->ev_wb) (void)hipEventDestroy(e->ev_wb);
	if (e->stream3) (void)hipStreamDestroy(e->stream3);
	if (e->ev_wc) (void)hipEventDestroy(e->ev_wc);
	if (e->stream4) (void)hipStreamDestroy(e->stream4);
	if (e->stream2) (void)hipStreamDestroy(e->stream2);
	if (e->stream) (void)hipStreamDestroy(e->stream);
	delete e;
}

extern "C" int cs_engine_stats(const cs_engine_t *e, cs_stats_t *st)
{
	if (!e || !st) return fail(CS_EINVAL, "null argument");
	*st = e->st;
	if (e->twin) { // both pass contexts count
		const cs_stats_t &t = e->twin->st;
		st->reads += t.reads; st->bases += t.bases; st->mems += t.mems; st->seeds += t.seeds; st->bwt_queries += t.bwt_queries; st->bwt_calls += t.bwt_calls;
		st->sal_queries += t.sal_queries; st->sal_calls += t.sal_calls; st->overflow_mems += t.overflow_mems; st->seed_kernel_ms += t.seed_kernel_ms;
		st->sal_kernel_ms += t.sal_kernel_ms; st->total_ms += t.total_ms; st->seed_kernel_launches += t.seed_kernel_launches; st->overflow_kernel_ms += t.overflow_kernel_ms;
		st->overflow_kernel_launches += t.overflow_kernel_launches; st->reseed_text_calls += t.reseed_text_calls; st->reseed_index_calls += t.reseed_index_calls;
		st->sweep_text_calls += t.sweep_text_calls; st->r3_text_seeds += t.r3_text_seeds;
	}
	return CS_OK;
}
extern "C" void cs_engine_reset_stats(cs_engine_t *e)
{
	if (!e || pipe_busy(e)) return; // (the seeding thread owns the counters while batches are in flight)
	(void)hipSetDevice(e->device);
	for (int ci = 0; ci < n_pass_ctx(e); ++ci) {
		cs_engine *c = pass_ctx(e, ci);
		memset(&c->st, 0, sizeof c->st);
		c->stream_bytes = 0;
		(void)hipMemsetAsync(c->d_evc.p, 0, (size_t)N_KID * N_EV * sizeof(unsigned long long), c->stream);
		(void)hipStreamSynchronize(c->stream);
	}
}
extern "C" int cs_engine_traffic_model(cs_engine_t *e, cs_traffic_t *out)
{
	if (!e || !out) return fail(CS_EINVAL, "null argument");
	if (pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_traffic_model: submitted batches are in flight, collect them first");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMemcpyAsync(&out->events[0][0], e->d_evc.p, (size_t)N_KID * N_EV * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	uint64_t stream_bytes = e->stream_bytes;
	if (e->twin) { // both pass contexts count
		std::vector<unsigned long long> ev2((size_t)N_KID * N_EV);
		HIP_TRY(hipMemcpy(ev2.data(), e->twin->d_evc.p, ev2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		for (int a = 0; a < N_KID; ++a) for (int b = 0; b < N_EV; ++b) out->events[a][b] += ev2[(size_t)a * N_EV + b];
		stream_bytes += e->twin->stream_bytes;
	}
	const uint64_t sa_b = e->ix.fsa64 ? 8 : 4;
	const uint64_t eb[N_EV] = {32, 16, 8, sa_b, sa_b, 4, 8, 1, 16, 32};
	for (int i = 0; i < N_EV; ++i) out->event_bytes[i] = eb[i];
	out->stream_bytes = stream_bytes;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ helpers
struct U32ToU64 { __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; } };

// exclusive prefix sum of n u32 counts into n+1 u64 offsets starting at `init`
static int scan_counts(cs_engine *e, const uint32_t *cnt, uint64_t *off, size_t n, uint64_t init)
{
	auto in = rocprim::make_transform_iterator(cnt, U32ToU64());
	size_t tmp = 0;
	// n+1 outputs: the input iterator is read one past the end, so cnt has a zeroed tail slot
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, off, init, n + 1, rocprim::plus<uint64_t>(), e->stream));
	CS_TRY(e->d_tmp.reserve(tmp + 16));
	HIP_TRY(rocprim::exclusive_scan(e->d_tmp.p, tmp, in, off, init, n + 1, rocprim::plus<uint64_t>(), e->stream));
	return CS_OK;
}

__global__ void max_len_kernel(const uint64_t *off, int64_t n, uint64_t n_bases, unsigned long long *out_max, unsigned long long *bad)
{
	unsigned long long len = 0;
	if (blockIdx.x == 0 && threadIdx.x == 0 && (off[0] != 0 || off[n] != n_bases)) atomicAdd(bad, 1ull); // the reads must tile [0, n_bases)
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) { // few waves: few atomics
		uint64_t a = off[r], b = off[r + 1];
		if (b < a || b > n_bases) atomicAdd(bad, 1ull); else if (b - a > len) len = b - a;
	}
	for (int o = 32; o > 0; o >>= 1) { unsigned long long other = __shfl_xor(len, o); len = other > len ? other : len; } // one atomic per wave
	if ((threadIdx.x & 63) == 0) atomicMax(out_max, len);
}
__global__ void collect_overflow_kernel(const uint32_t *cnt, int64_t n, uint32_t cap, uint32_t first_read, uint32_t *list, unsigned long long *n_ovf)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n || cnt[r] <= cap) return;
	unsigned long long slot = atomicAdd(n_ovf, 1ull);
	list[slot] = first_read + (uint32_t)r;
}
__global__ void patch_counts_kernel(const uint32_t *cnt2, const uint32_t *list, int64_t n_ovf, uint32_t first_read, uint32_t *cnt)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_ovf) return;
	cnt[list[t] - first_read] = cnt2[t];
}
// second-pass variant of sort_compact_kernel: task t holds read list[t]
__global__ void sort_compact_list_kernel(const OutMem *raw, const uint32_t *cnt2, uint32_t cap2, const uint32_t *list, int64_t n_tasks,
                                         const uint64_t *mem_off, OutMem *mems)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_tasks) return;
	uint32_t n = cnt2[t];
	const OutMem *src = raw + (size_t)t * cap2;
	OutMem *dst = mems + mem_off[list[t]];
	for (uint32_t a = 0; a < n; ++a) {
		uint64_t ka = src[a].info; uint32_t rank = 0;
		for (uint32_t b = 0; b < n; ++b) { uint64_t kb = src[b].info; rank += (kb < ka) || (kb == ka && b < a); }
		dst[rank] = src[a];
	}
}

// SA slots as CompSeed merges them (comp_seed.cpp:2327-2334): identical slots inside one 512-read batch are looked up once.
// key = batch << 37 | slot (slots < 2^37: checked at engine creation); sorted, then the distinct keys are counted.
__global__ void sal_keys_kernel(const OutSeed *seeds, const uint64_t *seed_off, int64_t n_reads, uint64_t *keys)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t hi = (uint64_t)(r >> 9) << 37; // BATCH_SIZE 512, comp_seed.h:36
		for (uint64_t j = seed_off[r]; j < seed_off[r + 1]; ++j) keys[j] = hi | (uint64_t)seeds[j].rbeg;
	}
}
__global__ void count_distinct_kernel(const uint64_t *keys, uint64_t n, unsigned long long *out)
{
	unsigned long long c = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		c += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// A few counter words from the device to the host.  Not a hipMemcpyAsync: that would queue behind whatever large transfer the
// copy engine is busy with (the results of the previous sub-batch on their way to the host, cs_engine_seed_batch), and the
// SMEM stage reads its counters back ten times per pass.  A one-wave kernel stores the words straight into pinned host memory.
__global__ void fetch_words_kernel(unsigned long long *dst_host, const unsigned long long *src, int n)
{
	if ((int)threadIdx.x < n) dst_host[threadIdx.x] = src[threadIdx.x];
	__threadfence_system();
}
template <typename T>
static int fetch_words(PinBuf<unsigned long long> &h, size_t at, const T *d_src, int n, hipStream_t s)
{
	static_assert(sizeof(T) == 8, "64-bit words");
	if (h.dp && n <= 64) {
		hipLaunchKernelGGL(fetch_words_kernel, dim3(1), dim3(64), 0, s, h.dp + at, (const unsigned long long *)d_src, n);
		HIP_TRY(hipGetLastError());
	} else HIP_TRY(hipMemcpyAsync(h.p + at, d_src, (size_t)n * 8, hipMemcpyDeviceToHost, s));
	return CS_OK;
}

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)std::max<int64_t>(1, (n + block - 1) / block); }

constexpr int SMEM_BLOCK = 256;
// LEP entries kept in LDS per lane: 20 x 16 B x 256 lanes = 80 KiB per workgroup => two workgroups (8 waves) per CU;
// 10 => 40 KiB => four workgroups (16 waves) per CU, more of the list spilling to global memory.  CS_LEP_LDS selects.

static int launch_smem(cs_engine *e, const cs_params_t *par, const uint64_t *d_off, const uint32_t *d_ids, int64_t n_tasks,
                       OutMem *out, uint32_t *cnt, uint32_t cap, uint32_t max_len)
{
	unsigned blocks = (unsigned)std::min<int64_t>((int64_t)e->n_cu * e->blocks_per_cu, (n_tasks + SMEM_BLOCK - 1) / SMEM_BLOCK);
	if (blocks == 0) return CS_OK;
	const int SMEM_LEP_LDS = g_lep_lds;
	blocks = (unsigned)std::min<int64_t>((int64_t)e->n_cu * 2, (n_tasks + SMEM_BLOCK - 1) / SMEM_BLOCK);
	uint32_t spill_cap = max_len + 1 > (uint32_t)SMEM_LEP_LDS ? max_len + 1 - SMEM_LEP_LDS : 1;
	// long reads: fewer resident workgroups rather than an unbounded spill area (one LEP list per lane, worst case = read length)
	size_t per_block = (size_t)SMEM_BLOCK * spill_cap * sizeof(uint4);
	blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, ((size_t)8 << 30) / per_block));
	CS_TRY(e->d_spill.reserve((size_t)blocks * SMEM_BLOCK * spill_cap));
	SeedArgs A;
	A.ix = e->ix; A.seq = e->d_seq.p; A.off = d_off; A.read_ids = d_ids; A.n_tasks = n_tasks;
	A.out = out; A.out_cnt = cnt; A.cap = cap;
	A.min_seed_len = par->min_seed_len;
	A.split_len = (int)(1.0 * par->min_seed_len * par->split_factor + .499); // comp_seed.cpp:2279 (double arithmetic)
	A.split_width = (uint32_t)par->split_width;
	A.max_mem_intv = par->max_mem_intv;
	A.task_counter = e->d_ctr.p; A.spill = e->d_spill.p; A.spill_cap = spill_cap; A.n_queries = e->d_ctr.p + 1; A.evc = e->d_evc.p;
	HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, sizeof(unsigned long long), e->stream));
	HIP_TRY(hipEventRecord(e->ev[0], e->stream));
	if (par->count_traffic) hipLaunchKernelGGL((smem_kernel<SMEM_BLOCK, 20, true>), dim3(blocks), dim3(SMEM_BLOCK), 0, e->stream, A);
	else hipLaunchKernelGGL((smem_kernel<SMEM_BLOCK, 20, false>), dim3(blocks), dim3(SMEM_BLOCK), 0, e->stream, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e->ev[1], e->stream));
	if (d_ids) e->st.overflow_kernel_launches++; else e->st.seed_kernel_launches++;
	return CS_OK;
}

static int add_event_ms(cs_engine *e, hipEvent_t a, hipEvent_t b, double *acc)
{
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, a, b));
	*acc += ms;
	return CS_OK;
}


// k-mer filter of the text for the window lanes (smem_split.hpp, kmer_filter_*): ~22 bits per text position, one per engine,
// rebuilt when a call uses another min_seed_len (0.3 s at hg19 scale)
static int build_kmer_filter(cs_engine *e, int k)
{
	if (!e->ix.text2 || k < 8 || k > 24 || !e->opt.kmer_filter) return CS_OK;
	if (e->bloom_k == k) return CS_OK;
	if (e->owner || e->in_shared_pass) return CS_OK; // the filter is the owner's, and is not rebuilt under a running pass (pass_on_ctx does that, alone)
	uint32_t bits = 10; // 2^bits words: at least seq_len / 3 of them
	while (bits < 34 && ((uint64_t)1 << bits) * 3 < e->ix.seq_len) ++bits;
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	e->bloom_k = 0;
	if ((((size_t)8) << bits) + ((size_t)24 << 30) >= free_b + e->d_bloom.cap * 8) return CS_OK; // no room: the window lanes do without
	CS_TRY(e->d_bloom.reserve((size_t)1 << bits));
	HIP_TRY(hipMemsetAsync(e->d_bloom.p, 0, ((size_t)8) << bits, e->stream));
	hipLaunchKernelGGL(kmer_filter_fill_kernel, dim3((unsigned)(e->n_cu * 32)), dim3(256), 0, e->stream, e->ix, k, e->d_bloom.p, bits);
	HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(e->stream));
	e->bloom_k = k; e->bloom_bits = bits; e->bloom_tried_k = k;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ split SMEM path
// Runs the three rounds for reads [0, nb) of d_off with fwd_kernel / bwd_kernel (smem_split.hpp).  On return d_cnt holds
// the number of mems per read, d_out the first `cap` of each, d_ovfrec/*n_ovf the rest.  Returns 1 when a task queue
// overflowed (the caller then falls back to the fused kernel for this sub-batch).
// launch the counting instantiation of a kernel (cs_params_t.count_traffic) or the plain one
#define LAUNCH_CT(count, KERN, grid, stream, ...)                                                              \
	do {                                                                                                       \
		if (count) hipLaunchKernelGGL((KERN<256, true>), grid, dim3(256), 0, stream, __VA_ARGS__);             \
		else hipLaunchKernelGGL((KERN<256, false>), grid, dim3(256), 0, stream, __VA_ARGS__);                  \
	} while (0)
static int run_smem_split_body(cs_engine *e, const cs_params_t *par, const uint64_t *d_off, int64_t nb, uint32_t max_len, uint64_t *n_ovf_out);
static int run_smem_split(cs_engine *e, const cs_params_t *par, const uint64_t *d_off, int64_t nb, uint32_t max_len, uint64_t *n_ovf_out)
{
	const int rc = run_smem_split_body(e, par, d_off, nb, max_len, n_ovf_out);
	if (rc != CS_OK) { // every early exit: kernels on the side streams may still be appending to buffers the next call reuses
		const std::string keep = g_err;
		(void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream2); (void)hipStreamSynchronize(e->stream3); (void)hipStreamSynchronize(e->stream4);
		(void)hipGetLastError();
		g_err = keep;
	}
	return rc;
}
static int run_smem_split_body(cs_engine *e, const cs_params_t *par, const uint64_t *d_off, int64_t nb, uint32_t max_len, uint64_t *n_ovf_out)
{
	hipStream_t s = e->stream;
	const uint32_t dis = par->sst_mode != 0 ? par->disable : ~0u; // sst_mode 0: the literal algorithm, every shortcut off
	const bool count = par->count_traffic != 0;
	*n_ovf_out = 0;
	if (par->split_width > 16382) return 1; // min_intv does not fit the 14-bit task field: use the fused kernel
	const uint32_t stride = max_len + 1;
	const uint64_t fq_cap = (uint64_t)nb * 8 + 4096, ovf_cap = (uint64_t)nb * 4 + 65536;
	uint64_t chunk = std::max<uint64_t>(4096, e->lep_arena_bytes / ((size_t)stride * sizeof(uint4)));
	chunk = std::min<uint64_t>(chunk, fq_cap);
	CS_TRY(e->d_fqA.reserve(fq_cap)); CS_TRY(e->d_fqB.reserve(fq_cap)); CS_TRY(e->d_fqR.reserve((size_t)nb + 1));
	const bool have_arrays = e->ix.rep != nullptr && par->sst_mode != 0;
	const bool r2text = have_arrays && !(dis & CS_DISABLE_R2_TEXT);
	if (r2text) { CS_TRY(e->d_auxA.reserve(fq_cap)); CS_TRY(e->d_auxB.reserve(fq_cap)); }
	CS_TRY(e->d_bq.reserve(chunk)); CS_TRY(e->d_lep.reserve(chunk * stride));
	CS_TRY(e->d_ovfrec.reserve(ovf_cap));
	unsigned long long *C = e->d_sctr.p, *H = e->h_sctr.p; // [0] task ctr [1] next-queue length [2..5] backward queues [6] overflow mems [7] error [8] queries
	HIP_TRY(hipMemsetAsync(C, 0, 32 * sizeof(unsigned long long), s));
	HIP_TRY(hipMemsetAsync(e->d_cnt.p, 0, ((size_t)nb + 1) * sizeof(uint32_t), s));

	SplitArgs A;
	A.ix = e->ix; A.seq = e->d_seq.p; A.off = d_off; A.n_reads = nb;
	A.seqp = e->seqp_cur + (d_off - e->off_base); // record index = (off[r] >> 5) + r with r counted from the batch's first read
	if (dis & CS_DISABLE_TEXT_MODE) A.ix.text2 = nullptr;
	A.out = e->d_out.p; A.out_cnt = e->d_cnt.p; A.cap = e->cap;
	A.ovf = e->d_ovfrec.p; A.ovf_cnt = C + 6; A.ovf_cap = ovf_cap;
	A.min_seed_len = par->min_seed_len;
	A.split_len = (int)(1.0 * par->min_seed_len * par->split_factor + .499); // comp_seed.cpp:2279 (double arithmetic)
	A.split_width = (uint32_t)par->split_width; A.max_mem_intv = par->max_mem_intv;
	A.bq = e->d_bq.p;
	A.lep = e->d_lep.p; A.lep_stride = stride;
	A.task_ctr = C; A.n_queries = C + 8; A.err = C + 7; A.n_sst_hits = C + 9; A.sst = par->sst_mode; A.sst2 = e->d_sst2.p; A.jump = e->jump_k ? e->d_jump.p : nullptr; A.jump_k = e->jump_k;
	A.evc = e->d_evc.p;
	A.fq_cap = fq_cap; A.n_f_next = C + 1; A.n_btasks = C + 13; A.n_text_sweeps = C + 14; A.n_r2_quick = C + 11;
	A.text_sweep = (dis & CS_DISABLE_TEXT_SWEEP) ? 0 : 1;
	// window scheme for the backward sweeps (smem_split.hpp, bwd_win_run): needs the jump table and jump_k <= min_seed_len <= jump_k + 4
	A.win = !(dis & CS_DISABLE_WINDOW) && par->sst_mode != 0 && A.jump && A.jump_k <= A.min_seed_len && A.min_seed_len - 1 <= WIN_LANES ? 1 : 0;
	A.bloom = nullptr; A.bloom_bits = 0;
	if (A.win && !(dis & CS_DISABLE_KMER_FILTER)) { // k-mer filter for the window lanes: built at engine creation for -k 19, here for any other value on its first use
		CS_TRY(build_kmer_filter(e, A.min_seed_len));
		if (e->bloom_k == A.min_seed_len) { A.bloom = e->d_bloom.p; A.bloom_bits = e->bloom_bits; }
	}

	uint64_t *cur = e->d_fqA.p, *nxt = e->d_fqB.p;
	uint64_t *aux_cur = r2text ? e->d_auxA.p : nullptr, *aux_nxt = r2text ? e->d_auxB.p : nullptr;
	A.fq = cur; A.n_f = 0; A.fq_next = nxt; A.aux_next = aux_nxt;
	HIP_TRY(hipEventRecord(e->ev[0], s));
	hipLaunchKernelGGL(init_tasks_kernel, dim3(grid_for(nb, 256)), dim3(256), 0, s, A, cur, e->d_fqR.p);
	// Round 3 depends on nothing: it runs on a low-priority second stream and fills the tails of the launches below.
	// round 3 after rounds 1/2, mostly from the text (r3text_kernel).  Its text paths take "fewer than max_mem_intv occurrences" as
	// "unique" and compare the 255-capped rep[] bytes with min_seed_len + 1, so -y 1 and -k >= 254 stay on the index (fwd_kernel)
	const bool r3_text = have_arrays && !(dis & CS_DISABLE_R3_TEXT) && A.max_mem_intv >= 2 && A.min_seed_len + 1 <= 254;
	bool r3_async = A.max_mem_intv > 0 && !r3_text;
	const int r3_after = 0; // forward launches before round 3 starts on the index (measured: at once is best)
	auto launch_r3 = [&]() -> int {
		SplitArgs R = A;
		R.fq = e->d_fqR.p; R.n_f = (uint64_t)nb; R.task_ctr = C + 10;
		HIP_TRY(hipEventRecord(e->ev_r3a, s));
		HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_r3a, 0));
		unsigned gr = (unsigned)std::min<uint64_t>((uint64_t)e->n_cu * e->occ_fwd, ((uint64_t)nb + 255) / 256);
		LAUNCH_CT(count, fwd_kernel, dim3(gr), e->stream2, R);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e->ev_r3b, e->stream2));
		return CS_OK;
	};
	bool r3_launched = false;
	if (r3_async && r3_after <= 0) { CS_TRY(launch_r3()); r3_launched = true; }
	// r3text_kernel runs on the second stream beside the late iterations (from the 5th on they carry < 2 % of the tasks but
	// still cost a launch chain and a host round trip each); it works from a snapshot of the mem counts
	const int r3t_iter = e->opt.r3_text_iter; // measured in round 2, one pass at a time: 2: 67.8, 3: 66.9, 4: 66.0, 5: 66.8, 6: 68.0 ms; with two passes in flight (round 3): 4: 47.6, 5: 47.0, 6: 47.0 ms per step, one at a time 54.6 / 54.6 / 55.7; repeat50: 4 and 5 the same (110.8 / 111.1 ms per step)
	bool r3t_launched = false;
	if (r3_text) { CS_TRY(e->d_cnt_snap.reserve((size_t)nb + 1)); CS_TRY(e->d_pending.reserve((size_t)nb + 1)); }
	auto launch_r3text = [&](const uint64_t *queue, const unsigned long long *queue_n) -> int {
		// which reads still have calls in the queue (their mem lists are not final; for all others the text answers everything)
		HIP_TRY(hipMemsetAsync(e->d_pending.p, 0, (size_t)nb, s));
		if (queue) hipLaunchKernelGGL(mark_pending_kernel, dim3((unsigned)e->n_cu * 4), dim3(256), 0, s, queue, queue_n, fq_cap, nb, e->d_pending.p);
		// the snapshot of the mem counts is taken on the main stream, between two iterations: every entry below a count is complete
		// (on the side stream it could run beside the next iteration's kernels, which bump a count before they store the mem)
		HIP_TRY(hipMemcpyAsync(e->d_cnt_snap.p, e->d_cnt.p, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipEventRecord(e->ev_r3a, s));
		HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_r3a, 0));
		hipLaunchKernelGGL(r3text_kernel, dim3((unsigned)std::min<uint64_t>((uint64_t)e->n_cu * 16, ((uint64_t)nb + 255) / 256)), dim3(256), 0, e->stream2, A,
		                   (const uint32_t *)e->d_cnt_snap.p, C + 15, (const uint8_t *)e->d_pending.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e->ev_r3b, e->stream2));
		return CS_OK;
	};
	const bool fwd0_on = par->sst_mode != 0 && A.ix.text2 && A.jump && A.jump_k >= 8 && !(dis & CS_DISABLE_FWD0);
	uint64_t n_f = (uint64_t)nb;
	for (int iter = 0; n_f > 0; ++iter) {
		A.fq_next = nxt; A.aux_next = aux_nxt;
		for (uint64_t c0 = 0; c0 < n_f; ) {
			uint64_t cn = std::min<uint64_t>(chunk, n_f - c0);
			A.fq = cur + c0; A.n_f = cn;
			HIP_TRY(hipMemsetAsync(C, 0, sizeof(unsigned long long), s));
			HIP_TRY(hipMemsetAsync(C + 13, 0, sizeof(unsigned long long), s));
			const bool r3_only = false;
			HIP_TRY(hipMemsetAsync(e->d_bq.p, 0xff, cn * sizeof(BTask), s)); // slots without a call stay "no class"
			unsigned gf = (unsigned)std::min<uint64_t>((uint64_t)e->n_cu * e->occ_fwd, (cn + 255) / 256);
			if (iter == 0 && fwd0_on) { // the calls at the first base of each read: a kernel without LEPs, backward tasks, SST (smem_split.hpp)
				unsigned g0 = (unsigned)std::min<uint64_t>((uint64_t)e->n_cu * 8, (cn + 255) / 256);
				LAUNCH_CT(count, fwd0_kernel, dim3(g0), s, A, cur + c0);
				HIP_TRY(hipMemsetAsync(C, 0, sizeof(unsigned long long), s));
			}
			LAUNCH_CT(count, fwd_kernel, dim3(gf), s, A);
			HIP_TRY(hipGetLastError());
			if (r3_async && !r3_launched && iter + 1 >= r3_after) { CS_TRY(launch_r3()); r3_launched = true; }
			if (!r3_only) { // one launch works through all four size classes of the chunk's backward sweeps
				unsigned cap_blocks = (unsigned)(e->n_cu * (A.win ? e->occ_win : e->occ_bwd));
				const int win0_occ = 8; // (sharing the CUs between the two window kernels by grid size was measured: slower in every split)
				HIP_TRY(hipMemsetAsync(C + 2, 0, 4 * sizeof(unsigned long long), s));
				HIP_TRY(hipEventRecord(e->ev_wa, s)); // forward launch done, counters zeroed
				HIP_TRY(hipStreamWaitEvent(e->stream3, e->ev_wa, 0));
				// side streams: the calls with more than 46 / 64 LEPs, one wave each (few on a mostly unique genome, many on a repeat-rich one;
				// chains of dependent reads, so what counts is waves in flight: six blocks per CU, registers spilled and all, run a
				// repeat-rich genome 6 % faster than four), and the calls without stored LEPs (the bulk of the calls)
				HIP_TRY(hipStreamWaitEvent(e->stream4, e->ev_wa, 0));
				if (count) hipLaunchKernelGGL(bwd_wide_kernel<true>, dim3((unsigned)std::min<uint64_t>((uint64_t)e->n_cu * CS_WIDE_BLOCKS, (cn + 255) / 256)), dim3(256), 0, e->stream4, A,
				                              (const BTask *)e->d_bq.p, cn, C + 5);
				else hipLaunchKernelGGL(bwd_wide_kernel<false>, dim3((unsigned)std::min<uint64_t>((uint64_t)e->n_cu * CS_WIDE_BLOCKS, (cn + 255) / 256)), dim3(256), 0, e->stream4, A,
				                        (const BTask *)e->d_bq.p, cn, C + 5);
				HIP_TRY(hipEventRecord(e->ev_wc, e->stream4));
				if (A.win)
					LAUNCH_CT(count, bwd_win0_kernel, dim3((unsigned)std::min<uint64_t>((uint64_t)e->n_cu * win0_occ, (cn + 255) / 256)), e->stream3, A,
					          (const BTask *)e->d_bq.p, cn);
				HIP_TRY(hipEventRecord(e->ev_wb, e->stream3));
				if (A.win) LAUNCH_CT(count, bwd_win_kernel, dim3((unsigned)std::min<uint64_t>(cap_blocks, (cn + 7) / 8)), s, A, (const BTask *)e->d_bq.p, cn, C + 2);
				else LAUNCH_CT(count, bwd_all_kernel, dim3((unsigned)std::min<uint64_t>(cap_blocks, (cn + 15) / 16)), s, A, (const BTask *)e->d_bq.p, cn, C + 2);
				HIP_TRY(hipGetLastError());
				HIP_TRY(hipStreamWaitEvent(s, e->ev_wb, 0)); // all must be done before the slots and the LEP arena are reused
				HIP_TRY(hipStreamWaitEvent(s, e->ev_wc, 0));
			}
			c0 += cn;
		}
		if (r2text) { // re-seeding calls of unique SMEMs pushed by this iteration: answer from the text what the text can answer
			// ... and copy what is left, without the no-op slots, into the queue this iteration has just consumed
			HIP_TRY(hipMemsetAsync(C + 16, 0, sizeof(unsigned long long), s));
			hipLaunchKernelGGL(r2text_kernel, dim3((unsigned)e->n_cu * 8), dim3(256), 0, s, A, (const uint64_t *)nxt, (const uint64_t *)aux_nxt,
			                   (const unsigned long long *)(C + 1), C + 11, C + 12, cur, C + 16);
			HIP_TRY(hipGetLastError());
		}
		if (r3_text && !r3t_launched && iter + 1 >= r3t_iter) { // (the queue of the next iteration: what r2text_kernel has left, or what was pushed)
			CS_TRY(launch_r3text(r2text ? cur : nxt, r2text ? C + 16 : C + 1)); r3t_launched = true;
		}
		CS_TRY(fetch_words(e->h_sctr, 0, C, 32, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (H[7]) return 1; // a queue or the overflow records ran full: the caller redoes the sub-batch with the fused kernel
		// byte model, stream part: this iteration's queue words read (8 B), words pushed (8 B + 8 B side word), and per slot a
		// backward task record cleared, written and scanned by three kernels (16 B each)
		e->stream_bytes += n_f * (8 + 16 * 5) + H[1] * (r2text ? 16 + 16 + 8 : 16);
		n_f = r2text ? H[16] : H[1];
		if (e->opt.verbose > 1) fprintf(stderr, "[cs_engine] iter %d: next queue %llu, sweeps created (last chunk) %llu, text sweeps so far %llu, reseed text %llu / index %llu\n", iter, H[1], H[13], H[14], H[11], H[12]);
		HIP_TRY(hipMemsetAsync(C + 1, 0, sizeof(unsigned long long), s));
		if (!r2text) { std::swap(cur, nxt); std::swap(aux_cur, aux_nxt); } // (r2text_kernel has compacted the next queue into `cur`)
		if (iter > (int)max_len + 8) return fail(CS_EDEVICE, "SMEM task chain did not terminate"); // a read has at most len pivots
	}
	if (r3_text && !r3t_launched) { CS_TRY(launch_r3text(nullptr, nullptr)); r3t_launched = true; }
	if (r3_async && !r3_launched) { CS_TRY(launch_r3()); r3_launched = true; }
	if (r3_async || r3_text) HIP_TRY(hipStreamWaitEvent(s, e->ev_r3b, 0)); // join the round-3 stream
	HIP_TRY(hipEventRecord(e->ev[1], s));
	CS_TRY(fetch_words(e->h_sctr, 0, C, 32, s));
	HIP_TRY(hipStreamSynchronize(s));
	if (H[7]) return 1; // round 3 is joined only here: it may have run the overflow records full after the last check in the loop
	CS_TRY(add_event_ms(e, e->ev[0], e->ev[1], &e->st.seed_kernel_ms));
	e->st.seed_kernel_launches++;
	e->st.bwt_queries += H[8]; e->st.bwt_calls += H[8] - H[9]; // calls = queries not answered by the on-device SST
	e->st.reseed_text_calls += H[11]; e->st.reseed_index_calls += H[12]; e->st.sweep_text_calls += H[14]; e->st.r3_text_seeds += H[15];
	*n_ovf_out = H[6]; // (<= ovf_cap: a record beyond it sets the error flag)
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ the hot path
static bool pipe_busy(const cs_engine *e);
// d_recs: the reads as pack_reads_kernel's records when the host made them (d_bases is then null), else null
static int seed_device_impl(cs_engine *e, const cs_params_t *par, int64_t n_reads, const uint8_t *d_bases, const uint64_t *d_off,
                            uint64_t n_bases, uint64_t *n_mems_out, uint64_t *n_seeds_out, const uint4 *d_recs)
{
	hipStream_t s = e->stream;
	*n_mems_out = *n_seeds_out = 0;
	if (par->min_seed_len < 1 || par->max_occ < 1 || par->split_width < 0) return fail(CS_EINVAL, "bad seeding parameters");
	CS_TRY(e->d_mem_off.reserve((size_t)n_reads + 2));
	if (n_reads == 0) {
		HIP_TRY(hipMemsetAsync(e->d_mem_off.p, 0, 8, s));
		if (par->want_sal) { CS_TRY(e->d_seed_off.reserve(2)); HIP_TRY(hipMemsetAsync(e->d_seed_off.p, 0, 8, s)); }
		HIP_TRY(hipStreamSynchronize(s));
		return CS_OK;
	}
	HIP_TRY(hipEventRecord(e->ev[2], s));
	// read lengths: MAX_READ_LEN 65535 (comp_seed.h:39; the reference aborts at main.cpp:83-86)
	HIP_TRY(hipMemsetAsync(e->d_ctr.p + 2, 0, 3 * sizeof(unsigned long long), s));
	hipLaunchKernelGGL(max_len_kernel, dim3((unsigned)std::min<int64_t>(grid_for(n_reads, 256), (int64_t)e->n_cu * 8)), dim3(256), 0, s, d_off, n_reads, n_bases, e->d_ctr.p + 3, e->d_ctr.p + 4);
	CS_TRY(fetch_words(e->h_ctr, 0, e->d_ctr.p, 8, s));
	HIP_TRY(hipStreamSynchronize(s));
	if (e->h_ctr.p[4]) return fail(CS_EINVAL, "offsets must start at 0, be non-decreasing and end at n_bases");
	uint32_t max_len = (uint32_t)e->h_ctr.p[3];
	if (e->h_ctr.p[3] >= 65535) return fail(CS_ERANGE, "read length exceeds the limit 65535 (MAX_READ_LEN)");

	// The split kernels read the reads as 16-byte records of 32 bases (pack_reads_kernel), made straight from the caller's bytes
	// (which stay untouched).  The byte-per-base nt4 copy is what the fused kernel reads: made only when that one runs.
	if (d_recs && e->smem_mode != 1) return fail(CS_EINVAL, "host-made records need the split kernels");
	const bool raw_ok = e->smem_mode == 1 && (d_recs || ((uintptr_t)d_bases & 7u) == 0);
	bool have_nt4 = false;
	auto make_nt4 = [&]() -> int {
		if (have_nt4) return CS_OK;
		CS_TRY(e->d_seq.reserve((size_t)n_bases + 64));
		if (n_bases && d_recs) hipLaunchKernelGGL(unpack_reads_kernel, dim3((unsigned)std::min<int64_t>(grid_for(n_reads * 8, 256), (int64_t)e->n_cu * 16)), dim3(256), 0, s, d_recs, d_off, n_reads, e->d_seq.p);
		else if (n_bases) {
			unsigned g = (unsigned)std::min<uint64_t>((n_bases + 255) / 256, (uint64_t)e->n_cu * 16);
			hipLaunchKernelGGL(nt4_kernel, dim3(g), dim3(256), 0, s, d_bases, e->d_seq.p, n_bases);
		}
		HIP_TRY(hipMemsetAsync(e->d_seq.p + n_bases, 4, 64, s));
		have_nt4 = true;
		return CS_OK;
	};
	if (!raw_ok) CS_TRY(make_nt4());
	const uint64_t n_rec = (n_bases >> 5) + (uint64_t)n_reads;
	if (d_recs) { e->seqp_cur = d_recs; e->off_base = d_off; }
	else if (e->smem_mode == 1) {
		CS_TRY(e->d_seqp.reserve((size_t)n_rec + 4));
		e->seqp_cur = e->d_seqp.p;
		const dim3 gp((unsigned)std::min<int64_t>(grid_for(n_reads * 8, 256), (int64_t)e->n_cu * 16));
		if (raw_ok) hipLaunchKernelGGL(pack_reads_kernel<true>, gp, dim3(256), 0, s, d_bases, d_off, n_reads, n_bases, e->d_seqp.p);
		else hipLaunchKernelGGL(pack_reads_kernel<false>, gp, dim3(256), 0, s, (const uint8_t *)e->d_seq.p, d_off, n_reads, n_bases, e->d_seqp.p);
		e->off_base = d_off;
	}
	// byte model, stream part: the bases are read once (twice and written once where the nt4 copy is made), the records written, and
	// read by the forward, backward and round-3 kernels
	e->stream_bytes += (d_recs ? 0 : n_bases * (raw_ok ? 1 : 3)) + (e->smem_mode == 1 ? 16 * n_rec * (d_recs ? 3 : 4) : n_bases * 3);

	const uint32_t cap = e->cap;
	int64_t per_launch = (int64_t)std::max<size_t>(1024, e->max_raw_bytes / ((size_t)cap * sizeof(OutMem)));
	per_launch = std::min<int64_t>(per_launch, n_reads);
	CS_TRY(e->d_out.reserve((size_t)per_launch * cap));
	CS_TRY(e->d_cnt.reserve((size_t)per_launch + 1));
	CS_TRY(e->d_ovf.reserve((size_t)per_launch));
	CS_TRY(e->d_mems.reserve((size_t)n_reads * 10 + 1024)); CS_TRY(e->d_salcnt.reserve((size_t)n_reads * 10 + 1024));
	bool salcnt_ok = true; // every mem's slot count was written by a sort_compact*_kernel of the split path

	uint64_t total_mems = 0;
	for (int64_t b0 = 0; b0 < n_reads; b0 += per_launch) {
		int64_t nb = std::min<int64_t>(per_launch, n_reads - b0);
		if (e->smem_mode == 1) {
			uint64_t n_ovf2 = 0;
			int rc = run_smem_split(e, par, d_off + b0, nb, max_len, &n_ovf2);
			if (rc < 0) return rc;
			if (rc == 0) {
				if (n_ovf2) { // the few mems beyond a read's first `cap`: sort their records by read id
					e->st.overflow_mems += n_ovf2;
					CS_TRY(e->d_okey.reserve(n_ovf2)); CS_TRY(e->d_oidx.reserve(n_ovf2)); CS_TRY(e->d_okey2.reserve(n_ovf2)); CS_TRY(e->d_oidx2.reserve(n_ovf2));
					hipLaunchKernelGGL(ovf_keys_kernel, dim3(grid_for((int64_t)n_ovf2, 256)), dim3(256), 0, s, e->d_ovfrec.p, n_ovf2, e->d_okey.p, e->d_oidx.p);
					size_t tb = 0;
					HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, e->d_okey.p, e->d_okey2.p, e->d_oidx.p, e->d_oidx2.p, (size_t)n_ovf2, 0u, 32u, s));
					CS_TRY(e->d_tmp2.reserve(tb + 16));
					HIP_TRY(rocprim::radix_sort_pairs((void *)e->d_tmp2.p, tb, e->d_okey.p, e->d_okey2.p, e->d_oidx.p, e->d_oidx2.p, (size_t)n_ovf2, 0u, 32u, s));
				}
				CS_TRY(scan_counts(e, e->d_cnt.p, e->d_mem_off.p + b0, (size_t)nb, total_mems));
				CS_TRY(fetch_words(e->h_ctr, 0, e->d_mem_off.p + b0 + nb, 1, s));
				HIP_TRY(hipStreamSynchronize(s));
				uint64_t new_total = e->h_ctr.p[0];
				CS_TRY(e->d_mems.reserve((size_t)new_total + 16, true, s, (size_t)total_mems)); CS_TRY(e->d_salcnt.reserve((size_t)new_total + 16, true, s, (size_t)total_mems));
				const uint32_t mo = (uint32_t)par->max_occ;
				int fast16 = cap >= 16 ? 1 : 0;
				if (fast16) hipLaunchKernelGGL(sort_compact16_kernel, dim3(grid_for(nb * 16, 256)), dim3(256), 0, s, e->d_out.p, e->d_cnt.p, cap,
				                               e->d_mem_off.p + b0, nb, e->d_mems.p, e->d_salcnt.p, mo);
				if (fast16) hipLaunchKernelGGL(sort_compact_wave_kernel, dim3(grid_for(nb, 256)), dim3(256), 0, s, e->d_out.p, e->d_cnt.p, cap, e->d_ovfrec.p,
				                               e->d_okey2.p, e->d_oidx2.p, n_ovf2, e->d_mem_off.p + b0, nb, e->d_mems.p, e->d_salcnt.p, mo);
				else hipLaunchKernelGGL(sort_compact2_kernel, dim3(grid_for(nb, 128)), dim3(128), 0, s, e->d_out.p, e->d_cnt.p, cap, e->d_ovfrec.p,
				                        e->d_okey2.p, e->d_oidx2.p, n_ovf2, e->d_mem_off.p + b0, nb, e->d_mems.p, 0, e->d_salcnt.p, mo);
				HIP_TRY(hipGetLastError());
				total_mems = new_total;
				continue;
			}
			// rc == 1: a task queue overflowed -- redo this sub-batch with the fused kernel
		}
		CS_TRY(make_nt4()); // (the fused kernel reads a byte per base)
		salcnt_ok = false;
		HIP_TRY(hipMemsetAsync(e->d_cnt.p + nb, 0, sizeof(uint32_t), s));
		CS_TRY(launch_smem(e, par, d_off + b0, nullptr, nb, e->d_out.p, e->d_cnt.p, cap, max_len));
		HIP_TRY(hipMemsetAsync(e->d_ctr.p + 2, 0, sizeof(unsigned long long), s));
		hipLaunchKernelGGL(collect_overflow_kernel, dim3(grid_for(nb, 256)), dim3(256), 0, s, e->d_cnt.p, nb, cap, 0u, e->d_ovf.p, e->d_ctr.p + 2);
		CS_TRY(fetch_words(e->h_ctr, 0, e->d_ctr.p, 8, s));
		HIP_TRY(hipStreamSynchronize(s));
		CS_TRY(add_event_ms(e, e->ev[0], e->ev[1], &e->st.seed_kernel_ms));
		e->st.bwt_queries += e->h_ctr.p[1]; e->st.bwt_calls += e->h_ctr.p[1];
		HIP_TRY(hipMemsetAsync(e->d_ctr.p + 1, 0, sizeof(unsigned long long), s));
		int64_t n_ovf = (int64_t)e->h_ctr.p[2];
		uint32_t cap2 = 0;
		if (n_ovf > 0) { // second pass over the few reads with more than `cap` mems, capacity grown until everything fits
			e->st.overflow_mems += (uint64_t)n_ovf;
			cap2 = std::max<uint32_t>(256, cap * 8);
			for (;;) {
				CS_TRY(e->d_out2.reserve((size_t)n_ovf * cap2));
				CS_TRY(e->d_cnt2.reserve((size_t)n_ovf));
				CS_TRY(e->d_tmp2.reserve((size_t)n_ovf * 4 + 16));
				CS_TRY(launch_smem(e, par, d_off + b0, e->d_ovf.p, n_ovf, e->d_out2.p, e->d_cnt2.p, cap2, max_len));
				HIP_TRY(hipMemsetAsync(e->d_ctr.p + 2, 0, sizeof(unsigned long long), s));
				// reuse the overflow counter to see whether any task still does not fit
				hipLaunchKernelGGL(collect_overflow_kernel, dim3(grid_for(n_ovf, 256)), dim3(256), 0, s, e->d_cnt2.p, n_ovf, cap2, 0u,
				                   (uint32_t *)e->d_tmp2.p, e->d_ctr.p + 2);
				CS_TRY(fetch_words(e->h_ctr, 0, e->d_ctr.p, 8, s));
				HIP_TRY(hipStreamSynchronize(s));
				CS_TRY(add_event_ms(e, e->ev[0], e->ev[1], &e->st.overflow_kernel_ms));
				e->st.bwt_queries += e->h_ctr.p[1]; e->st.bwt_calls += e->h_ctr.p[1];
				HIP_TRY(hipMemsetAsync(e->d_ctr.p + 1, 0, sizeof(unsigned long long), s));
				if (e->h_ctr.p[2] == 0) break;
				if (cap2 >= (1u << 22)) return fail(CS_ERANGE, "a read produced more than 4M mems");
				cap2 *= 8;
			}
			hipLaunchKernelGGL(patch_counts_kernel, dim3(grid_for(n_ovf, 256)), dim3(256), 0, s, e->d_cnt2.p, e->d_ovf.p, n_ovf, 0u, e->d_cnt.p);
		}
		// offsets of this sub-batch, continuing the running total
		CS_TRY(scan_counts(e, e->d_cnt.p, e->d_mem_off.p + b0, (size_t)nb, total_mems));
		CS_TRY(fetch_words(e->h_ctr, 0, e->d_mem_off.p + b0 + nb, 1, s));
		HIP_TRY(hipStreamSynchronize(s));
		uint64_t new_total = e->h_ctr.p[0];
		CS_TRY(e->d_mems.reserve((size_t)new_total + 16, true, s, (size_t)total_mems));
		// mem_off already holds absolute offsets, so base_off = 0 and the per-read offset array is shifted by b0
		hipLaunchKernelGGL(sort_compact_kernel, dim3(grid_for(nb, 128)), dim3(128), 0, s, e->d_out.p, e->d_cnt.p, cap, e->d_mem_off.p + b0,
		                   (uint64_t)0, nb, (const uint32_t *)nullptr, e->d_mems.p);
		if (n_ovf > 0)
			hipLaunchKernelGGL(sort_compact_list_kernel, dim3(grid_for(n_ovf, 64)), dim3(64), 0, s, e->d_out2.p, e->d_cnt2.p, cap2, e->d_ovf.p,
			                   n_ovf, e->d_mem_off.p + b0, e->d_mems.p);
		HIP_TRY(hipGetLastError());
		total_mems = new_total;
	}
	*n_mems_out = total_mems;

	if (par->want_sal) { // comp_seed.cpp:2306-2347
		CS_TRY(e->d_seed_off.reserve((size_t)n_reads + 2));
		CS_TRY(e->d_seed_of_mem.reserve((size_t)total_mems + 2));
		// per-mem slot counts are written into the tail of d_seed_of_mem's own storage via a temp
		DevBuf<uint64_t> &som = e->d_seed_of_mem;
		CS_TRY(e->d_tmp.reserve(((size_t)total_mems + 2) * 8 + 1024));
		uint64_t *cnt64 = (uint64_t *)e->d_tmp.p;
		if (salcnt_ok && e->smem_mode == 1) { CS_TRY(e->d_salcnt.reserve((size_t)total_mems + 16, true, s, (size_t)total_mems)); cnt64 = e->d_salcnt.p; } // (counted while sorting)
		HIP_TRY(hipEventRecord(e->ev[0], s));
		HIP_TRY(hipMemsetAsync(cnt64 + total_mems, 0, 8, s));
		if (total_mems && cnt64 != e->d_salcnt.p)
			hipLaunchKernelGGL(sal_count_kernel, dim3(grid_for((int64_t)total_mems, 256)), dim3(256), 0, s, e->d_mems.p, total_mems,
			                   (uint32_t)par->max_occ, cnt64);
		{ // scan needs its own temp storage: keep the counts where they are and scan with a second buffer
			size_t tmp = 0;
			HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, cnt64, som.p, (uint64_t)0, (size_t)total_mems + 1, rocprim::plus<uint64_t>(), s));
			CS_TRY(e->d_tmp2.reserve(tmp + 16));
			HIP_TRY(rocprim::exclusive_scan((void *)e->d_tmp2.p, tmp, cnt64, som.p, (uint64_t)0, (size_t)total_mems + 1, rocprim::plus<uint64_t>(), s));
		}
		CS_TRY(fetch_words(e->h_ctr, 0, som.p + total_mems, 1, s));
		HIP_TRY(hipStreamSynchronize(s));
		uint64_t total_seeds = e->h_ctr.p[0];
		CS_TRY(e->d_seeds.reserve((size_t)total_seeds + 16));
		const bool fused_gather = !e->opt.count_sal_merged && (e->ix.fsa32 || e->ix.fsa64); // (the merged-call statistic needs the slots)
		if (total_mems) {
			if (fused_gather) hipLaunchKernelGGL(sal_expand_kernel<true>, dim3(grid_for((int64_t)total_mems, 256)), dim3(256), 0, s, e->ix, e->d_mems.p, total_mems,
			                                     (uint32_t)par->max_occ, som.p, e->d_seeds.p);
			else hipLaunchKernelGGL(sal_expand_kernel<false>, dim3(grid_for((int64_t)total_mems, 256)), dim3(256), 0, s, e->ix, e->d_mems.p, total_mems,
			                        (uint32_t)par->max_occ, som.p, e->d_seeds.p);
		}
		hipLaunchKernelGGL(seed_off_kernel, dim3(grid_for(n_reads + 1, 256)), dim3(256), 0, s, e->d_mem_off.p, som.p, n_reads, e->d_seed_off.p);
		uint64_t sal_calls = total_seeds;
		if (e->opt.count_sal_merged && total_seeds) { // statistics option; the slots are still in rbeg here (the gather below overwrites them)
			unsigned bits = 38;
			while (bits < 64 && ((uint64_t)(n_reads >> 9) >> (bits - 37))) ++bits;
			CS_TRY(e->d_okey64.reserve((size_t)total_seeds)); CS_TRY(e->d_okey64b.reserve((size_t)total_seeds));
			hipLaunchKernelGGL(sal_keys_kernel, dim3((unsigned)std::min<int64_t>(grid_for(n_reads, 256), (int64_t)e->n_cu * 16)), dim3(256), 0, s,
			                   (const OutSeed *)e->d_seeds.p, (const uint64_t *)e->d_seed_off.p, n_reads, e->d_okey64.p);
			rocprim::double_buffer<uint64_t> kb(e->d_okey64.p, e->d_okey64b.p);
			size_t tb = 0;
			HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, kb, (size_t)total_seeds, 0u, bits, s));
			CS_TRY(e->d_tmp2.reserve(tb + 16));
			HIP_TRY(rocprim::radix_sort_keys((void *)e->d_tmp2.p, tb, kb, (size_t)total_seeds, 0u, bits, s));
			HIP_TRY(hipMemsetAsync(e->d_ctr.p + 5, 0, sizeof(unsigned long long), s));
			hipLaunchKernelGGL(count_distinct_kernel, dim3((unsigned)e->n_cu * 8), dim3(256), 0, s, (const uint64_t *)kb.current(), total_seeds, e->d_ctr.p + 5);
			CS_TRY(fetch_words(e->h_ctr, 5, e->d_ctr.p + 5, 1, s));
			HIP_TRY(hipStreamSynchronize(s));
			sal_calls = e->h_ctr.p[5];
		}
		if (total_seeds && !fused_gather) {
			if (e->ix.fsa32 || e->ix.fsa64)
				hipLaunchKernelGGL(sal_gather_kernel, dim3(grid_for((int64_t)total_seeds, 256)), dim3(256), 0, s, e->ix, e->d_seeds.p, total_seeds);
			else
				hipLaunchKernelGGL(sal_walk_kernel, dim3(grid_for((int64_t)total_seeds, 256)), dim3(256), 0, s, e->ix, e->d_seeds.p, total_seeds);
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e->ev[1], s));
		HIP_TRY(hipStreamSynchronize(s));
		CS_TRY(add_event_ms(e, e->ev[0], e->ev[1], &e->st.sal_kernel_ms));
		*n_seeds_out = total_seeds;
		e->st.sal_queries += total_seeds; e->st.sal_calls += sal_calls;
	}
	HIP_TRY(hipEventRecord(e->ev[3], s));
	HIP_TRY(hipStreamSynchronize(s));
	CS_TRY(add_event_ms(e, e->ev[2], e->ev[3], &e->st.total_ms));
	e->st.reads += (uint64_t)n_reads; e->st.bases += n_bases; e->st.mems += total_mems; e->st.seeds += *n_seeds_out;
	// byte model, stream part: a mem is written raw, read by the sort and written again (32 B each)
	e->stream_bytes += total_mems * 96;
	return CS_OK;
}

extern "C" int cs_engine_seed_batch_device(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *d_bases,
                                           const uint64_t *d_offsets, uint64_t n_bases, cs_result_t *out)
{
	if (!e || !par || !out || n_reads < 0 || (n_reads > 0 && !d_offsets) || (n_bases > 0 && !d_bases))
		return fail(CS_EINVAL, "cs_engine_seed_batch_device: bad argument");
	if (n_reads >= (int64_t)0xffffffffll) return fail(CS_ERANGE, "more than 2^32-1 reads in one call");
	if (pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_seed_batch_device: submitted batches are in flight, collect them first");
	HIP_TRY(hipSetDevice(e->device));
	uint64_t nm = 0, ns = 0;
	invalidate_last(e);
	CS_TRY(seed_device_impl(e, par, n_reads, d_bases, d_offsets, n_bases, &nm, &ns, nullptr));
	e->last.valid = true; e->last.n_reads = n_reads; e->last.n_mems = nm; e->last.n_seeds = ns; e->last.want_sal = par->want_sal;
	out->n_reads = n_reads; out->n_mems = nm; out->n_seeds = ns;
	out->mem_off = e->d_mem_off.p; out->mems = (const cs_intv_t *)e->d_mems.p;
	out->seed_off = par->want_sal ? e->d_seed_off.p : nullptr;
	out->seeds = par->want_sal ? (const cs_seed_t *)e->d_seeds.p : nullptr;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ the host variants
// cs_engine_seed_batch_packed / cs_engine_seed_batch: the boundary the reference-side patch calls, i.e. the part
// of the path the reference overlaps with kt_pipeline (main.cpp:438, cstl/kthread.c:121: read the next chunk / process / write).
// The batch is cut into sub-batches; an upload thread stages sub-batch i+1 while the calling thread seeds sub-batch i, whose
// results are packed on the device (16-byte mems, 8-byte seeds: include/compseed_amd.h) into one of two buffers and go to pinned
// host memory on a copy stream of their own while sub-batch i+1 is seeded; cs_engine_seed_batch additionally expands finished
// sub-batches to cs_intv_t / cs_seed_t on an expander thread (itself multi-threaded) beside all that.
__global__ void pack_mems16_kernel(const OutMem *m, uint64_t n, uint4 *out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const OutMem v = m[i];
		const uint64_t beg = v.info >> 32, end = v.info & 0xffffffffull;
		const uint64_t w0 = v.x0 | (v.x2 & 0x7fffffffull) << 33, w1 = v.x1 | beg << 33 | end << 48 | (v.x2 >> 31) << 63;
		out[i] = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
	}
}
// a seed travels as its rbeg only, in 40 bits: positions are below 2^37 (checked at engine creation), so the low word and the fifth byte go
// into two planes (coalesced stores, aligned loads for the consumer: cs_packed_seed_rbeg) -- 5 instead of 8 bytes of PCIe traffic per seed
__global__ void pack_rbeg_kernel(const OutSeed *sd, uint64_t n, uint32_t *lo, uint8_t *hi)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint64_t v = (uint64_t)sd[i].rbeg; lo[i] = (uint32_t)v; hi[i] = (uint8_t)(v >> 32); }
}
__global__ void shift_words_kernel(const uint64_t *in, uint64_t n, uint64_t add, uint64_t *out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i] + add;
}
__global__ void rebase_words_kernel(uint64_t *io, uint64_t n)
{
	const uint64_t base = io[0]; // read by every thread before the grid-wide... single block: see launch
	__syncthreads();
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) io[i] -= base;
}

extern "C" int cs_host_alloc(size_t bytes, void **ptr)
{
	if (!ptr) return fail(CS_EINVAL, "null argument");
	*ptr = nullptr;
	HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
	return CS_OK;
}
extern "C" int cs_host_free(void *ptr)
{
	if (ptr) HIP_TRY(hipHostFree(ptr));
	return CS_OK;
}

namespace {
struct SubBatch { int64_t r0 = 0, n = 0; uint64_t b0 = 0, nb = 0, mem_base = 0, nm = 0, seed_base = 0, ns = 0; int slot = 0; };

// expand packed sub-batch results into cs_intv_t / cs_seed_t arrays, `threads` workers over contiguous read ranges
void expand_range(const cs_packed_result_t &P, cs_intv_t *mems, cs_seed_t *seeds, int64_t r0, int64_t r1)
{
	for (int64_t r = r0; r < r1; ++r) {
		uint64_t sd = P.seed_off ? P.seed_off[r] : 0;
		for (uint64_t m = P.mem_off[r]; m < P.mem_off[r + 1]; ++m) {
			cs_intv_t v; cs_unpack_mem(&P, m, &v);
			mems[m] = v;
			if (P.seed_off) {
				const int32_t qb = (int32_t)(v.info >> 32), ln = (int32_t)(uint32_t)v.info - qb;
				const uint32_t c = cs_mem_seed_count(&v, P.max_occ);
				for (uint32_t j = 0; j < c; ++j) { cs_seed_t x = {cs_packed_seed_rbeg(&P, sd + j), qb, ln}; seeds[sd + j] = x; }
				sd += c;
			}
		}
	}
}
void expand_parallel(const cs_packed_result_t &P, cs_intv_t *mems, cs_seed_t *seeds, int64_t r0, int64_t r1, int threads)
{
	if (threads < 1) threads = 1;
	if (r1 - r0 < 4096 || threads == 1) { expand_range(P, mems, seeds, r0, r1); return; }
	std::vector<std::thread> th;
	// equal shares of the MEMS, not of the reads: find read boundaries by bisection on mem_off
	const uint64_t m0 = P.mem_off[r0], m1 = P.mem_off[r1];
	int64_t prev = r0;
	for (int t = 1; t <= threads; ++t) {
		int64_t cut = r1;
		if (t < threads) {
			const uint64_t want = m0 + (m1 - m0) * (uint64_t)t / (uint64_t)threads;
			cut = std::lower_bound(P.mem_off + r0, P.mem_off + r1, want) - P.mem_off;
			if (cut < prev) cut = prev;
		}
		if (cut > prev) th.emplace_back(expand_range, std::cref(P), mems, seeds, prev, cut);
		prev = cut;
	}
	for (auto &t : th) t.join();
}
} // namespace

// ---- the engine's host pipeline: three threads behind cs_engine_submit / cs_engine_collect_packed
//   upload thread   stages the parts (sub-batches) of submitted batches, in order, into one of two device input slots
//   seeding thread  seeds a staged part (seed_device_impl), packs its results into one of two device pack slots and queues their
//                   download into the batch's pinned result slot (one of two) on the copy stream
//   expander thread (cs_engine_seed_batch only) expands downloaded parts into cs_intv_t / cs_seed_t arrays
// so that, for a caller that keeps two batches submitted, the upload of batch n+1, the seeding of batch n and the download of
// batch n-1 run at the same time -- what kt_pipeline (main.cpp:438) does for the reference's read / process / write steps.
// All engine state touched by seed_device_impl belongs to the seeding thread while a batch is in flight: the blocking entry
// points (device variant, digest, gather, primitives) refuse to run then.
static double pipe_ms() // wall clock of the verbose log lines, from the first one
{
	static const auto t_epoch = std::chrono::steady_clock::now();
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_epoch).count();
}
struct HostJob {
	uint64_t batch = 0; int part = 0, n_parts = 0;
	const uint8_t *bases = nullptr; const uint64_t *offsets = nullptr;
	int64_t r0 = 0, n = 0, n_reads = 0; uint64_t b0 = 0, nb = 0;
	cs_params_t par{}; bool pk16 = false, expand = false, packed = false; // packed: the host makes the records (host_pack.cpp)
	int in_slot = 0;
};
struct XJob { uint64_t batch; int64_t r0, n; uint64_t mem_base, nm, seed_base, ns; hipEvent_t ev; bool last; };
struct BatchState {
	uint64_t id = ~0ull; int64_t n_reads = 0; int parts_total = 0, parts_queued = 0; uint64_t mem_base = 0, seed_base = 0;
	int rc = CS_OK; std::string err; bool pk16 = false, sal = false, expand = false, expanded = false; int max_occ = 0;
	int ctx = 0;                         // pass context that seeded its last part (the whole batch, if it was not cut)
};
struct HostPipe {
	std::thread th_up, th_seed[2], th_x;
	std::mutex mu; std::condition_variable cv;
	bool started = false, quit = false;
	std::deque<HostJob> q_up, q_seed; std::deque<XJob> q_x; bool x_busy = false;
	int in_free[3] = {1, 1, 1};
	uint64_t pack_turn = 0;              // running number of the part whose results are packed and sent home next: parts are seeded by two threads, packed in order
	std::atomic<uint64_t> n_submitted{0}, n_collected{0}; uint64_t parts_seen = 0; // (one submitting and one collecting thread may run at the same time)
	long long handed = -1;               // batch whose pinned result slot the caller currently holds (until its next collect)
	BatchState bs[PIPE_DEPTH];           // batch id % PIPE_DEPTH
};

static void pipe_upload_thread(cs_engine *e)
{
	HostPipe &hp = *e->hp;
	(void)hipSetDevice(e->device);
	for (;;) {
		HostJob j;
		{
			std::unique_lock<std::mutex> lk(hp.mu);
			hp.cv.wait(lk, [&] { return hp.quit || (!hp.q_up.empty() && (hp.in_free[0] || hp.in_free[1] || hp.in_free[2])); });
			if (hp.quit) return;
			j = hp.q_up.front(); hp.q_up.pop_front();
			j.in_slot = hp.in_free[0] ? 0 : hp.in_free[1] ? 1 : 2; hp.in_free[j.in_slot] = 0;
		}
		hipError_t he = hipSuccess;
		const double tu0 = e->opt.verbose > 1 ? pipe_ms() : 0.0;
		if (j.packed) {
			// the reads as records, made here chunk by chunk: the copy of chunk i runs beside the packing of chunk i + 1
			uint4 *st = e->hp_stage[j.in_slot].p;
			const int64_t csz = std::max<int64_t>(262144, (j.n + 7) / 8);
			for (int64_t c0 = 0; c0 < j.n && he == hipSuccess; c0 += csz) {
				const int64_t c1 = std::min<int64_t>(j.n, c0 + csz);
				cs_pack_reads_host_(j.bases, j.offsets, j.r0, j.n, c0, c1, st, e->opt.host_pack_threads, 0);
				const uint64_t f = ((j.offsets[j.r0 + c0] - j.b0) >> 5) + (uint64_t)c0, l = ((j.offsets[j.r0 + c1] - j.b0) >> 5) + (uint64_t)c1;
				he = hipMemcpyAsync(e->hp_in[j.in_slot].p + f * 16, st + f, (size_t)(l - f) * 16, hipMemcpyHostToDevice, e->s_up);
			}
		} else if (j.nb) he = hipMemcpyAsync(e->hp_in[j.in_slot].p, j.bases + j.b0, (size_t)j.nb, hipMemcpyHostToDevice, e->s_up);
		if (he == hipSuccess && j.offsets) he = hipMemcpyAsync(e->hp_inoff[j.in_slot].p, j.offsets + j.r0, ((size_t)j.n + 1) * 8, hipMemcpyHostToDevice, e->s_up);
		if (he == hipSuccess && j.offsets) { hipLaunchKernelGGL(rebase_words_kernel, dim3(1), dim3(1024), 0, e->s_up, e->hp_inoff[j.in_slot].p, (uint64_t)j.n + 1); he = hipGetLastError(); }
		if (he == hipSuccess) he = hipStreamSynchronize(e->s_up);
		if (e->opt.verbose > 1) fprintf(stderr, "[cs_engine] batch %llu part %d/%d: %.1f MB %s in %.1f ms (from %.1f to %.1f ms)\n", (unsigned long long)j.batch, j.part + 1, j.n_parts, (double)(j.packed ? ((j.nb >> 5) + (uint64_t)j.n) * 16 : j.nb) / 1e6, j.packed ? "packed on the host and uploaded" : "uploaded", pipe_ms() - tu0, tu0, pipe_ms());
		std::lock_guard<std::mutex> lk(hp.mu);
		if (he != hipSuccess) { (void)hipGetLastError(); BatchState &b = hp.bs[j.batch % PIPE_DEPTH]; if (b.rc == CS_OK) { b.rc = CS_EDEVICE; b.err = std::string("upload: ") + hipGetErrorString(he); } }
		hp.q_seed.push_back(j);
		hp.cv.notify_all();
	}
}

static void pipe_expand_thread(cs_engine *e)
{
	HostPipe &hp = *e->hp;
	(void)hipSetDevice(e->device);
	for (;;) {
		XJob x;
		{
			std::unique_lock<std::mutex> lk(hp.mu);
			hp.cv.wait(lk, [&] { return hp.quit || !hp.q_x.empty(); });
			if (hp.quit) return;
			x = hp.q_x.front(); hp.q_x.pop_front(); hp.x_busy = true;
		}
		BatchState &b = hp.bs[x.batch % PIPE_DEPTH];
		const int rs = (int)(x.batch % PIPE_DEPTH);
		bool ok = hipEventSynchronize(x.ev) == hipSuccess;
		if (ok) {
			const double scale = x.last ? 1.0 : (double)b.n_reads / (double)(x.r0 + x.n) * ((x.r0 + x.n) * 4 < b.n_reads ? 1.2 : 1.08); // room for the whole batch at the first growth
			ok = !e->x_mems.reserve((size_t)((double)(x.mem_base + x.nm) * scale) + 1, (size_t)x.mem_base) &&
			     (!b.sal || !e->x_seeds.reserve((size_t)((double)(x.seed_base + x.ns) * scale) + 1, (size_t)x.seed_base));
		}
		if (ok) {
			cs_packed_result_t Q; memset(&Q, 0, sizeof Q);
			Q.n_reads = b.n_reads; Q.mem_format = b.pk16 ? CS_MEM_PACKED16 : CS_MEM_FULL32; Q.max_occ = b.max_occ;
			Q.mem_off = e->hp_moff[rs].p; Q.mems = e->hp_mems[rs].p; Q.seed_off = b.sal ? e->hp_soff[rs].p : nullptr; Q.seed_format = CS_SEED_RBEG40; Q.seed_rbeg_lo = b.sal ? e->hp_rlo[rs].p : nullptr; Q.seed_rbeg_hi = b.sal ? e->hp_rhi[rs].p : nullptr;
			expand_parallel(Q, e->x_mems.p, e->x_seeds.p, x.r0, x.r0 + x.n, e->opt.expand_threads);
		}
		std::lock_guard<std::mutex> lk(hp.mu);
		if (!ok && b.rc == CS_OK) { b.rc = CS_ENOMEM; b.err = "expanding the packed results failed"; }
		if (x.last) b.expanded = true;
		hp.x_busy = false;
		hp.cv.notify_all();
	}
}

// One of the (up to) two seeding threads: thread ci runs its passes on pass context ci.  Parts are taken in order; the running number a
// part gets when it is taken (k) fixes the order of the second stage -- packing the results and queueing their download, which needs
// the mem / seed totals of all earlier parts of the batch -- so a part that was seeded faster than its predecessor waits for it there.
static void pipe_seed_thread(cs_engine *e, int ci)
{
	HostPipe &hp = *e->hp;
	(void)hipSetDevice(e->device);
	cs_engine *c = pass_ctx(e, ci);
	for (;;) {
		HostJob j;
		uint64_t k;
		{
			std::unique_lock<std::mutex> lk(hp.mu);
			hp.cv.wait(lk, [&] { return hp.quit || !hp.q_seed.empty(); });
			if (hp.quit) return;
			j = hp.q_seed.front(); hp.q_seed.pop_front();
			k = hp.parts_seen++;                       // running part number: pack slot k & 1, part events k % 4
		}
		const int rs = (int)(j.batch % PIPE_DEPTH);
		BatchState &b = hp.bs[rs];
		int rc; { std::lock_guard<std::mutex> lk(hp.mu); rc = b.rc; }
		std::string err;
		auto hipf = [&](hipError_t he, const char *what) { if (he != hipSuccess && rc == CS_OK) { (void)hipGetLastError(); rc = he == hipErrorOutOfMemory ? CS_ENOMEM : CS_EDEVICE; err = std::string(what) + ": " + hipGetErrorString(he); } };
		uint64_t nm = 0, ns = 0;
		const bool sal = j.par.want_sal != 0;
		const size_t msz = j.pk16 ? 16 : 32;
		const double t0 = e->opt.verbose ? pipe_ms() : 0.0;
		if (rc == CS_OK) {
			rc = pass_on_ctx(e, c, &j.par, j.n, j.packed ? nullptr : e->hp_in[j.in_slot].p, e->hp_inoff[j.in_slot].p, j.nb, &nm, &ns, j.packed ? reinterpret_cast<const uint4 *>(e->hp_in[j.in_slot].p) : nullptr);
			if (rc != CS_OK) err = g_err;
		}
		if (e->opt.verbose) { // (with the wall clock of the seeding thread: idle gaps between parts show which neighbour it waited for)
			const double t1 = pipe_ms();
			fprintf(stderr, "[cs_engine] batch %llu part %d/%d: %lld reads seeded on context %d in %.1f ms (from %.1f to %.1f ms)\n", (unsigned long long)j.batch, j.part + 1, j.n_parts, (long long)j.n, ci, t1 - t0, t0, t1);
		}
		{ // the input slot is free again (the reads were converted into the context's own buffers); then wait for this part's turn in the second stage
			std::unique_lock<std::mutex> lk(hp.mu);
			hp.in_free[j.in_slot] = 1; hp.cv.notify_all();
			hp.cv.wait(lk, [&] { return hp.quit || hp.pack_turn == k; });
			if (hp.quit) return;
			if (rc == CS_OK) rc = b.rc;               // (an earlier part of the batch failed meanwhile)
		}
		const int ps = (int)(k & 1);
		const uint64_t mem_base = b.mem_base, seed_base = b.seed_base;
		if (rc == CS_OK) {
			// the pack buffers of this slot were last used by the part before the previous one: its download must be over
			if (k >= 2) hipf(hipEventSynchronize(e->hp_ev_dn[(k - 2) % 4]), "waiting for a download");
			// the batch that used this pinned result slot before may still be in the caller's hands: wait until it is given back
			{ std::unique_lock<std::mutex> lk(hp.mu); hp.cv.wait(lk, [&] { return hp.quit || hp.handed < 0 || (uint64_t)hp.handed == j.batch || (uint64_t)hp.handed % PIPE_DEPTH != j.batch % PIPE_DEPTH; }); if (hp.quit) return; }
			// (from here on the result slot is this batch's) offsets: one entry per read + 1
			if (j.part == 0 && (e->hp_moff[rs].reserve((size_t)j.n_reads + 1) != CS_OK || (sal && e->hp_soff[rs].reserve((size_t)j.n_reads + 1) != CS_OK))) { rc = CS_ENOMEM; err = g_err; }
			// pinned room for the whole batch: estimated from its first part, grown (keeping what has arrived) if that was too little
			const size_t need_m = (size_t)(mem_base + nm), need_s = (size_t)(seed_base + ns);
			if (need_m * msz > e->hp_mems[rs].cap || (sal && need_s > e->hp_rlo[rs].cap)) {
				hipf(hipStreamSynchronize(e->s_down), "draining downloads before growing the result buffers");
				{ std::unique_lock<std::mutex> lk(hp.mu); hp.cv.wait(lk, [&] { return hp.quit || (hp.q_x.empty() && !hp.x_busy); }); if (hp.quit) return; } // the expander reads these buffers
				const double scale = (double)j.n_reads / (double)(j.r0 + j.n) * ((j.r0 + j.n) * 4 < j.n_reads ? 1.2 : 1.08); // (a small first part predicts the batch less well)
				if (rc == CS_OK && e->hp_mems[rs].reserve((size_t)((double)need_m * scale) * msz + 4096, true, (size_t)mem_base * msz) != CS_OK) { rc = CS_ENOMEM; err = g_err; }
				if (rc == CS_OK && sal && (e->hp_rlo[rs].reserve((size_t)((double)need_s * scale) + 512, true, (size_t)seed_base) != CS_OK ||
				                            e->hp_rhi[rs].reserve((size_t)((double)need_s * scale) + 512, true, (size_t)seed_base) != CS_OK)) { rc = CS_ENOMEM; err = g_err; }
			}
		}
		if (rc == CS_OK && (e->hp_pk_mems[ps].reserve((size_t)nm * msz + 64) != CS_OK || (sal && (e->hp_pk_rlo[ps].reserve((size_t)ns + 8) != CS_OK || e->hp_pk_rhi[ps].reserve((size_t)ns + 8) != CS_OK)))) { rc = CS_ENOMEM; err = g_err; }
		if (rc == CS_OK) {
			hipStream_t s = c->stream;                 // (the context's own stream: its next pass starts behind these kernels)
			const unsigned g = (unsigned)e->n_cu * 8;
			hipLaunchKernelGGL(shift_words_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)c->d_mem_off.p, (uint64_t)j.n + 1, mem_base, e->hp_pk_moff[ps].p);
			if (nm) {
				if (j.pk16) hipLaunchKernelGGL(pack_mems16_kernel, dim3(g), dim3(256), 0, s, (const OutMem *)c->d_mems.p, nm, (uint4 *)e->hp_pk_mems[ps].p);
				else hipf(hipMemcpyAsync(e->hp_pk_mems[ps].p, c->d_mems.p, (size_t)nm * 32, hipMemcpyDeviceToDevice, s), "copying mems");
			}
			if (sal) {
				hipLaunchKernelGGL(shift_words_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)c->d_seed_off.p, (uint64_t)j.n + 1, seed_base, e->hp_pk_soff[ps].p);
				if (ns) hipLaunchKernelGGL(pack_rbeg_kernel, dim3(g), dim3(256), 0, s, (const OutSeed *)c->d_seeds.p, ns, e->hp_pk_rlo[ps].p, e->hp_pk_rhi[ps].p);
			}
			hipf(hipGetLastError(), "pack kernels");
			hipf(hipEventRecord(e->hp_ev_pk[ps], s), "event");
			hipf(hipStreamWaitEvent(e->s_down, e->hp_ev_pk[ps], 0), "event");
			hipf(hipMemcpyAsync(e->hp_moff[rs].p + j.r0, e->hp_pk_moff[ps].p, ((size_t)j.n + 1) * 8, hipMemcpyDeviceToHost, e->s_down), "download");
			if (nm) hipf(hipMemcpyAsync(e->hp_mems[rs].p + (size_t)mem_base * msz, e->hp_pk_mems[ps].p, (size_t)nm * msz, hipMemcpyDeviceToHost, e->s_down), "download");
			if (sal) {
				hipf(hipMemcpyAsync(e->hp_soff[rs].p + j.r0, e->hp_pk_soff[ps].p, ((size_t)j.n + 1) * 8, hipMemcpyDeviceToHost, e->s_down), "download");
				if (ns) hipf(hipMemcpyAsync(e->hp_rlo[rs].p + seed_base, e->hp_pk_rlo[ps].p, (size_t)ns * 4, hipMemcpyDeviceToHost, e->s_down), "download");
				if (ns) hipf(hipMemcpyAsync(e->hp_rhi[rs].p + seed_base, e->hp_pk_rhi[ps].p, (size_t)ns, hipMemcpyDeviceToHost, e->s_down), "download");
			}
			hipf(hipEventRecord(e->hp_ev_dn[k % 4], e->s_down), "event");
			if (j.part + 1 == j.n_parts) hipf(hipEventRecord(e->hp_ev_done[rs], e->s_down), "event");
		} else {
			// a failed part still owns hp_ev_dn[k % 4] in the eyes of part k + 2: leave a recorded event behind
			(void)hipEventRecord(e->hp_ev_dn[k % 4], e->s_down);
		}
		std::lock_guard<std::mutex> lk(hp.mu);
		if (rc != CS_OK && b.rc == CS_OK) { b.rc = rc; b.err = err; }
		if (rc == CS_OK) {
			b.mem_base += nm; b.seed_base += ns; b.ctx = ci;
			if (j.expand) { XJob x = {j.batch, j.r0, j.n, mem_base, nm, seed_base, ns, e->hp_ev_dn[k % 4], j.part + 1 == j.n_parts}; hp.q_x.push_back(x); }
		}
		b.parts_queued++;
		hp.pack_turn = k + 1;
		hp.cv.notify_all();
	}
}

static void pipe_stop(cs_engine *e)
{
	if (!e->hp) return;
	HostPipe &hp = *e->hp;
	{ std::lock_guard<std::mutex> lk(hp.mu); hp.quit = true; hp.cv.notify_all(); }
	if (hp.th_up.joinable()) hp.th_up.join();
	for (auto &t : hp.th_seed) if (t.joinable()) t.join();
	if (hp.th_x.joinable()) hp.th_x.join();
	delete e->hp; e->hp = nullptr;
}
static bool host_pipe_busy(const cs_engine *e) { return e->hp && e->hp->n_submitted.load() != e->hp->n_collected.load(); }
static bool dev_pipe_busy(const cs_engine *e);
static bool pipe_busy(const cs_engine *e) { return host_pipe_busy(e) || dev_pipe_busy(e); }


static int pipe_submit(cs_engine *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases, const uint64_t *offsets, bool expand)
{
	if (!e || !par || n_reads < 0 || (n_reads > 0 && !offsets)) return fail(CS_EINVAL, "cs_engine_submit: bad argument");
	if (n_reads >= (int64_t)0xffffffffll) return fail(CS_ERANGE, "more than 2^32-1 reads in one call");
	if (par->min_seed_len < 1 || par->max_occ < 1 || par->split_width < 0) return fail(CS_EINVAL, "bad seeding parameters");
	HIP_TRY(hipSetDevice(e->device));
	if (dev_pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_submit: device batches are in flight (cs_engine_submit_device), collect them first");
	if (!e->hp) e->hp = new HostPipe();
	HostPipe &hp = *e->hp;
	if (hp.n_submitted.load() - hp.n_collected.load() >= (uint64_t)PIPE_DEPTH) return fail(CS_EINVAL, "cs_engine_submit: four batches are in flight already, collect one first");
	uint64_t n_bases = 0, max_len = 0;
	if (n_reads > 0) {
		if (offsets[0] != 0) return fail(CS_EINVAL, "offsets[0] must be 0");
		n_bases = offsets[n_reads];
		if (n_bases > 0 && !bases) return fail(CS_EINVAL, "bases is null");
		// (10 M offsets are 6 ms on one thread, in front of everything else a blocking call does: four threads)
		const int vt = n_reads >= (1 << 20) ? 4 : 1;
		uint64_t vmax[4] = {0, 0, 0, 0}; bool vbad[4] = {false, false, false, false};
		auto vrange = [&](int t) {
			uint64_t m = 0; bool bad = false;
			for (int64_t r = n_reads * t / vt, r1 = n_reads * (t + 1) / vt; r < r1; ++r) { bad |= offsets[r + 1] < offsets[r]; m = std::max(m, offsets[r + 1] - offsets[r]); }
			vmax[t] = m; vbad[t] = bad;
		};
		if (vt == 1) vrange(0);
		else { std::thread th[3]; for (int t = 1; t < vt; ++t) th[t - 1] = std::thread(vrange, t); vrange(0); for (auto &t : th) t.join(); }
		for (int t = 0; t < vt; ++t) { if (vbad[t]) return fail(CS_EINVAL, "offsets must start at 0, be non-decreasing and end at n_bases"); max_len = std::max(max_len, vmax[t]); }
		if (max_len >= 65535) return fail(CS_ERANGE, "read length exceeds the limit 65535 (MAX_READ_LEN)");
	}
	const uint64_t id = hp.n_submitted.load();
	const int rs = (int)(id % PIPE_DEPTH);
	// parts: contiguous read ranges of about pipeline_reads reads (one, if the batch is not much larger than that)
	std::vector<HostJob> parts;
	const int64_t per = e->opt.pipeline_reads > 0 ? e->opt.pipeline_reads : std::max<int64_t>(n_reads, 1);
	// Parts exist to overlap upload, seeding and download INSIDE one batch, and each part pays the fixed cost of a pass (two passes over
	// 5 M reads take ~8 ms longer than one over 10 M).  When other batches are in flight the overlap comes from them -- upload of n+1 and
	// download of n-1 beside the seeding of n -- so a batch submitted behind another one is seeded whole.  This needs THREE batches in
	// flight to pay: with two, collect(n) returns when download(n) ends, only then can batch n+2 be submitted and uploaded, and download +
	// upload (79 ms) is longer than the seeding of batch n+1 (55 ms): measured 92 ms per batch whole against 65 ms in parts.
	const bool streaming = hp.n_submitted.load() - hp.n_collected.load() >= 2 && !expand;
	// part boundaries.  A batch that has the engine to itself (a blocking call, the first batch of a stream) cannot hide the upload of its
	// first part or the download of its last one behind anything, so those two are made small (0.2 of the nominal part) and the rest is
	// cut into parts of about 0.8: 10 M reads at 5 M nominal = 1 / 4 / 4 / 1 M (measured against 1.5 / 3.5 / 3.5 / 1.5: section 8 of DESIGN.md).
	std::vector<int64_t> cut(1, 0);
	const int64_t even = std::max<int64_t>(1, (n_reads + per / 2) / per);
	if (streaming || even < 2) cut.push_back(n_reads);
	else {
		const int64_t h = std::max<int64_t>(1, std::min<int64_t>(n_reads / 4, per * 2 / 10)), rest = n_reads - 2 * h;
		const int64_t km = std::max<int64_t>(1, (rest + per * 8 / 20) / std::max<int64_t>(1, per * 8 / 10));
		cut.push_back(h);
		for (int64_t i = 1; i <= km; ++i) cut.push_back(h + rest * i / km);
		cut.push_back(n_reads);
	}
	const int64_t kparts = (int64_t)cut.size() - 1;
	size_t in_cap = 0, off_cap = 0, stage_cap = 0;
	const bool host_pack = e->opt.host_pack_threads > 0 && e->smem_mode == 1;
	for (int64_t i = 0; i < kparts; ++i) {
		HostJob j; j.batch = id; j.part = (int)i; j.n_parts = (int)kparts; j.bases = bases; j.offsets = offsets; j.n_reads = n_reads;
		j.r0 = cut[(size_t)i]; j.n = cut[(size_t)i + 1] - j.r0;
		j.b0 = n_reads ? offsets[j.r0] : 0; j.nb = n_reads ? offsets[j.r0 + j.n] - j.b0 : 0;
		j.par = *par; j.pk16 = (e->ix.seq_len >> 33) == 0 && max_len < (1u << 15); j.expand = expand;
		j.packed = host_pack;
		const size_t n_rec = (size_t)(j.nb >> 5) + (size_t)j.n;
		in_cap = std::max<size_t>(in_cap, host_pack ? (n_rec + 4) * 16 : j.nb); off_cap = std::max<size_t>(off_cap, (size_t)j.n + 1);
		if (host_pack) stage_cap = std::max<size_t>(stage_cap, n_rec + 4);
		parts.push_back(j);
	}
	{ // buffers the threads will use: sized here, while no part of this batch is in flight (earlier batches never need more than they have)
		std::unique_lock<std::mutex> lk(hp.mu);
		if (in_cap + 64 > e->hp_in[0].cap || stage_cap > e->hp_stage[0].cap || off_cap > e->hp_inoff[0].cap || off_cap > e->hp_pk_moff[0].cap || (par->want_sal && off_cap > e->hp_pk_soff[0].cap)) {
			// a reallocation frees buffers the other batch may still be using: not only while its parts are queued or being seeded
			// (the input slots are given back right after seed_device_impl), but until the seeding thread has queued the pack kernels
			// and downloads of its LAST part (parts_queued == parts_total) and those have drained (s_down below)
			auto others_queued = [&] { for (int o = 1; o < PIPE_DEPTH; ++o) { const BatchState &ob = hp.bs[(rs + o) % PIPE_DEPTH]; if (ob.parts_queued != ob.parts_total) return false; } return true; };
			hp.cv.wait(lk, [&] { return hp.quit || (hp.q_up.empty() && hp.q_seed.empty() && hp.in_free[0] && hp.in_free[1] && hp.in_free[2] && others_queued()); });
			lk.unlock();
			HIP_TRY(hipStreamSynchronize(e->stream));
			if (e->twin) HIP_TRY(hipStreamSynchronize(e->twin->stream));
			HIP_TRY(hipStreamSynchronize(e->s_down));
			for (int k = 0; k < 3; ++k) { CS_TRY(e->hp_in[k].reserve(in_cap + 64)); CS_TRY(e->hp_stage[k].reserve(stage_cap)); CS_TRY(e->hp_inoff[k].reserve(off_cap)); }
			for (int k = 0; k < 2; ++k) {
				CS_TRY(e->hp_pk_moff[k].reserve(off_cap));
				if (par->want_sal) CS_TRY(e->hp_pk_soff[k].reserve(off_cap));
			}
			lk.lock();
		}
		BatchState &b = hp.bs[rs];
		b = BatchState();
		b.id = id; b.n_reads = n_reads; b.parts_total = (int)kparts; b.pk16 = parts[0].pk16; b.sal = par->want_sal != 0; b.expand = expand; b.max_occ = par->max_occ;
		for (auto &j : parts) hp.q_up.push_back(j);
		if (e->opt.verbose > 1) fprintf(stderr, "[cs_engine] batch %llu submitted at %.1f ms in %d part(s)\n", (unsigned long long)id, pipe_ms(), (int)kparts);
		hp.n_submitted++;
		invalidate_last(e);
		if (!hp.started) {
			hp.started = true;
			hp.th_up = std::thread(pipe_upload_thread, e); hp.th_x = std::thread(pipe_expand_thread, e);
			// One seeding thread / pass context here.  The code takes two (parts are seeded by whichever thread is free and packed in order), and
			// that was measured at hg19 scale: a stream of batches 59-65 instead of 57 ms per batch, a blocking call 107 instead of 105 ms --
			// two passes that run side by side end together, their downloads queue up behind each other (the stream is within 20 % of what the
			// 246 bytes per read of results allow over PCIe), and the next uploads wait for a free slot.  The device-resident form
			// (cs_engine_submit_device) is where the second context pays: 48 instead of 55 ms per 10 M reads.
			const int n_host_ctx = 1;
			for (int ci = 0; ci < n_host_ctx && ci < n_pass_ctx(e); ++ci) hp.th_seed[ci] = std::thread(pipe_seed_thread, e, ci);
		}
		hp.cv.notify_all();
	}
	return CS_OK;
}

static int pipe_collect(cs_engine *e, cs_packed_result_t *out)
{
	if (!e || !out) return fail(CS_EINVAL, "cs_engine_collect: null argument");
	if (!e->hp || e->hp->n_submitted.load() == e->hp->n_collected.load()) return fail(CS_EINVAL, "cs_engine_collect: nothing has been submitted");
	HIP_TRY(hipSetDevice(e->device));
	HostPipe &hp = *e->hp;
	const uint64_t id = hp.n_collected.load();
	const int rs = (int)(id % PIPE_DEPTH);
	BatchState &b = hp.bs[rs];
	{
		std::unique_lock<std::mutex> lk(hp.mu);
		hp.handed = -1;                      // the result handed out by the previous collect is given back: its slot may be overwritten
		hp.cv.notify_all();
		hp.cv.wait(lk, [&] { return b.parts_queued == b.parts_total; });
	}
	int rc = b.rc; std::string err = b.err;
	const double tc0 = e->opt.verbose > 1 ? pipe_ms() : 0.0;
	if (rc == CS_OK && hipEventSynchronize(e->hp_ev_done[rs]) != hipSuccess) { rc = CS_EDEVICE; err = "waiting for the download"; (void)hipGetLastError(); }
	if (e->opt.verbose > 1) fprintf(stderr, "[cs_engine] batch %llu collected at %.1f ms: waited %.1f ms for its download after its last part was queued\n", (unsigned long long)id, pipe_ms(), pipe_ms() - tc0);
	if (rc == CS_OK && b.expand) {
		std::unique_lock<std::mutex> lk(hp.mu);
		hp.cv.wait(lk, [&] { return b.expanded || b.rc != CS_OK; });
		rc = b.rc; err = b.err;
	}
	memset(out, 0, sizeof *out);
	{
		std::lock_guard<std::mutex> lk(hp.mu);
		hp.n_collected++;
		if (rc == CS_OK) hp.handed = (long long)id;
	}
	if (rc != CS_OK) return fail(rc, err);
	out->n_reads = b.n_reads; out->n_mems = b.mem_base; out->n_seeds = b.seed_base; out->max_occ = b.max_occ;
	out->mem_format = b.pk16 ? CS_MEM_PACKED16 : CS_MEM_FULL32;
	out->mem_off = e->hp_moff[rs].p; out->mems = e->hp_mems[rs].p;
	out->seed_off = b.sal ? e->hp_soff[rs].p : nullptr; out->seed_format = CS_SEED_RBEG40;
	out->seed_rbeg_lo = b.sal ? e->hp_rlo[rs].p : nullptr; out->seed_rbeg_hi = b.sal ? e->hp_rhi[rs].p : nullptr;
	{ // cs_engine_result_digest / gather_reads work on the device-side result, which is the whole batch only if it was not cut
		std::lock_guard<std::mutex> lk(hp.mu); // (a submit on another thread invalidates it under the same lock)
		cs_engine *c = pass_ctx(e, b.ctx);
		invalidate_last(e);
		c->last.valid = b.parts_total == 1 && !pipe_busy(e); c->last.n_reads = b.n_reads; c->last.n_mems = b.mem_base; c->last.n_seeds = b.seed_base; c->last.want_sal = b.sal;
		e->last_ctx = c;
	}
	return CS_OK;
}

// ---- device batches, two in flight: cs_engine_submit_device / cs_engine_collect_device.  Batch n runs on pass context n & 1, on a thread
// of its own, so the thin tail of one pass overlaps the dense start of the next; results come back in submission order as device
// pointers into that context's buffers.
struct DevPipe {
	std::thread th[2]; std::mutex mu; std::condition_variable cv; bool quit = false;
	int n_ctx = 1;
	int state[2] = {0, 0};                // 0 idle, 1 queued, 2 running, 3 done
	struct Job { cs_params_t par; int64_t n; const uint8_t *bases; const uint64_t *off; uint64_t nb; } job[2];
	int rc[2] = {0, 0}; std::string err[2]; uint64_t nm[2] = {0, 0}, ns[2] = {0, 0};
	std::atomic<uint64_t> n_sub{0}, n_col{0};
};
static bool dev_pipe_busy(const cs_engine *e) { return e->dp && e->dp->n_sub.load() != e->dp->n_col.load(); }
static void dev_pipe_thread(cs_engine *e, int ci)
{
	DevPipe &dp = *e->dp;
	(void)hipSetDevice(e->device);
	cs_engine *c = pass_ctx(e, ci);
	for (;;) {
		DevPipe::Job j;
		{
			std::unique_lock<std::mutex> lk(dp.mu);
			dp.cv.wait(lk, [&] { return dp.quit || dp.state[ci] == 1; });
			if (dp.quit) return;
			dp.state[ci] = 2; j = dp.job[ci];
		}
		uint64_t nm = 0, ns = 0;
		const int rc = pass_on_ctx(e, c, &j.par, j.n, j.bases, j.off, j.nb, &nm, &ns, nullptr);
		std::lock_guard<std::mutex> lk(dp.mu);
		dp.rc[ci] = rc; dp.err[ci] = rc != CS_OK ? g_err : std::string(); dp.nm[ci] = nm; dp.ns[ci] = ns;
		dp.state[ci] = 3;
		dp.cv.notify_all();
	}
}
static void dev_pipe_stop(cs_engine *e)
{
	if (!e->dp) return;
	DevPipe &dp = *e->dp;
	{ std::lock_guard<std::mutex> lk(dp.mu); dp.quit = true; dp.cv.notify_all(); }
	for (auto &t : dp.th) if (t.joinable()) t.join();
	delete e->dp; e->dp = nullptr;
}
extern "C" int cs_engine_submit_device(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_bases)
{
	if (!e || !par || n_reads < 0 || (n_reads > 0 && !d_offsets) || (n_bases > 0 && !d_bases)) return fail(CS_EINVAL, "cs_engine_submit_device: bad argument");
	if (n_reads >= (int64_t)0xffffffffll) return fail(CS_ERANGE, "more than 2^32-1 reads in one call");
	if (host_pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_submit_device: host batches are in flight (cs_engine_submit), collect them first");
	HIP_TRY(hipSetDevice(e->device));
	if (!e->dp) {
		CS_TRY(twin_create(e));
		e->dp = new DevPipe();
		e->dp->n_ctx = n_pass_ctx(e);
		for (int ci = 0; ci < e->dp->n_ctx; ++ci) e->dp->th[ci] = std::thread(dev_pipe_thread, e, ci);
	}
	DevPipe &dp = *e->dp;
	if (dp.n_sub.load() - dp.n_col.load() >= (uint64_t)dp.n_ctx) return fail(CS_EINVAL, dp.n_ctx == 2 ? "cs_engine_submit_device: two batches are in flight already, collect one first" : "cs_engine_submit_device: a batch is in flight already (passes_in_flight = 1), collect it first");
	const int ci = (int)(dp.n_sub.load() % (uint64_t)dp.n_ctx);
	std::lock_guard<std::mutex> lk(dp.mu);
	invalidate_last(e);
	dp.job[ci] = {*par, n_reads, d_bases, d_offsets, n_bases};
	dp.state[ci] = 1;
	dp.n_sub++;
	dp.cv.notify_all();
	return CS_OK;
}
extern "C" int cs_engine_collect_device(cs_engine_t *e, cs_result_t *out)
{
	if (!e || !out) return fail(CS_EINVAL, "cs_engine_collect_device: null argument");
	if (!e->dp || e->dp->n_sub.load() == e->dp->n_col.load()) return fail(CS_EINVAL, "cs_engine_collect_device: nothing has been submitted");
	DevPipe &dp = *e->dp;
	const int ci = (int)(dp.n_col.load() % (uint64_t)dp.n_ctx);
	cs_engine *c = pass_ctx(e, ci);
	int rc; std::string err; uint64_t nm, ns; bool sal; int64_t n;
	{
		std::unique_lock<std::mutex> lk(dp.mu);
		dp.cv.wait(lk, [&] { return dp.state[ci] == 3; });
		rc = dp.rc[ci]; err = dp.err[ci]; nm = dp.nm[ci]; ns = dp.ns[ci]; sal = dp.job[ci].par.want_sal != 0; n = dp.job[ci].n;
		dp.state[ci] = 0;
		dp.n_col++;
	}
	memset(out, 0, sizeof *out);
	if (rc != CS_OK) return fail(rc, err);
	{ // (a submit on another thread invalidates it under the same lock)
		std::lock_guard<std::mutex> lk(dp.mu);
		invalidate_last(e);
		c->last.valid = !dev_pipe_busy(e); c->last.n_reads = n; c->last.n_mems = nm; c->last.n_seeds = ns; c->last.want_sal = sal; e->last_ctx = c;
	}
	out->n_reads = n; out->n_mems = nm; out->n_seeds = ns;
	out->mem_off = c->d_mem_off.p; out->mems = (const cs_intv_t *)c->d_mems.p;
	out->seed_off = sal ? c->d_seed_off.p : nullptr; out->seeds = sal ? (const cs_seed_t *)c->d_seeds.p : nullptr;
	return CS_OK;
}

extern "C" int cs_engine_submit(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases, const uint64_t *offsets)
{
	return pipe_submit(e, par, n_reads, bases, offsets, false);
}
extern "C" int cs_engine_collect_packed(cs_engine_t *e, cs_packed_result_t *out) { return pipe_collect(e, out); }

extern "C" int cs_engine_seed_batch_packed(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases,
                                           const uint64_t *offsets, cs_packed_result_t *out)
{
	if (!e || !par || !out) return fail(CS_EINVAL, "cs_engine_seed_batch_packed: bad argument");
	if (pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_seed_batch_packed: submitted batches are in flight, collect them first");
	CS_TRY(pipe_submit(e, par, n_reads, bases, offsets, false));
	return pipe_collect(e, out);
}

extern "C" int cs_engine_seed_batch(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases,
                                    const uint64_t *offsets, cs_result_t *out)
{
	if (!e || !par || !out) return fail(CS_EINVAL, "cs_engine_seed_batch: bad argument");
	if (pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_seed_batch: submitted batches are in flight, collect them first");
	CS_TRY(pipe_submit(e, par, n_reads, bases, offsets, true));
	cs_packed_result_t P;
	CS_TRY(pipe_collect(e, &P));
	out->n_reads = n_reads; out->n_mems = P.n_mems; out->n_seeds = P.n_seeds;
	out->mem_off = P.mem_off; out->mems = e->x_mems.p;
	out->seed_off = P.seed_off; out->seeds = par->want_sal ? e->x_seeds.p : nullptr;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ digest / gather of the last result
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void digest_kernel(const uint64_t *w, uint64_t n, unsigned long long *out)
{
	unsigned long long acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		acc += splitmix64(w[i] + i * 0x9E3779B97F4A7C15ull);
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
extern "C" int cs_engine_result_digest(cs_engine_t *e0, cs_digest_t *out)
{
	if (!e0 || !out) return fail(CS_EINVAL, "null argument");
	cs_engine *e = e0->last_ctx ? e0->last_ctx : e0;  // the pass context that holds the result; everything below runs on it
	if (!e->last.valid || pipe_busy(e0)) return fail(CS_EINVAL, "cs_engine_result_digest: no whole-batch result is held on the device (call a seed function first)");
	HIP_TRY(hipSetDevice(e->device));
	hipStream_t s = e->stream;
	HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, 4 * sizeof(unsigned long long), s));
	const uint64_t n = (uint64_t)e->last.n_reads;
	const unsigned g = (unsigned)e->n_cu * 8;
	hipLaunchKernelGGL(digest_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)e->d_mem_off.p, n + 1, e->d_ctr.p + 0);
	if (e->last.n_mems) hipLaunchKernelGGL(digest_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)e->d_mems.p, e->last.n_mems * 4, e->d_ctr.p + 1);
	if (e->last.want_sal) {
		hipLaunchKernelGGL(digest_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)e->d_seed_off.p, n + 1, e->d_ctr.p + 2);
		if (e->last.n_seeds) hipLaunchKernelGGL(digest_kernel, dim3(g), dim3(256), 0, s, (const uint64_t *)e->d_seeds.p, e->last.n_seeds * 2, e->d_ctr.p + 3);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_ctr.p, e->d_ctr.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	out->mem_off = e->h_ctr.p[0]; out->mems = e->h_ctr.p[1]; out->seed_off = e->h_ctr.p[2]; out->seeds = e->h_ctr.p[3];
	return CS_OK;
}

__global__ void sel_counts_kernel(const uint64_t *ids, int64_t n_sel, uint64_t n_reads, const uint64_t *mem_off, const uint64_t *seed_off,
                                  uint64_t *cm, uint64_t *cs, unsigned long long *bad)
{
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > n_sel) return;
	if (t == n_sel) { cm[t] = 0; if (cs) cs[t] = 0; return; }
	const uint64_t r = ids[t];
	if (r >= n_reads) { atomicAdd(bad, 1ull); cm[t] = 0; if (cs) cs[t] = 0; return; }
	cm[t] = mem_off[r + 1] - mem_off[r];
	if (cs) cs[t] = seed_off[r + 1] - seed_off[r];
}
// 16 lanes per selected read copy its mems and seeds
__global__ void sel_copy_kernel(const uint64_t *ids, int64_t n_sel, const uint64_t *mem_off, const uint64_t *seed_off, const OutMem *mems, const OutSeed *seeds,
                                const uint64_t *om, const uint64_t *os, OutMem *out_m, OutSeed *out_s)
{
	const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	const uint32_t a = threadIdx.x & 15u;
	if (t >= n_sel) return;
	const uint64_t r = ids[t];
	for (uint64_t j = a, n = om[t + 1] - om[t]; j < n; j += 16) out_m[om[t] + j] = mems[mem_off[r] + j];
	if (seeds) for (uint64_t j = a, n = os[t + 1] - os[t]; j < n; j += 16) out_s[os[t] + j] = seeds[seed_off[r] + j];
}
extern "C" int cs_engine_gather_reads(cs_engine_t *e0, int64_t n_sel, const uint64_t *read_ids, cs_result_t *out)
{
	if (!e0 || !out || n_sel < 0 || (n_sel > 0 && !read_ids)) return fail(CS_EINVAL, "cs_engine_gather_reads: bad argument");
	cs_engine *e = e0->last_ctx ? e0->last_ctx : e0;  // the pass context that holds the result; everything below runs on it
	if (!e->last.valid || pipe_busy(e0)) return fail(CS_EINVAL, "cs_engine_gather_reads: no whole-batch result is held on the device (call a seed function first)");
	HIP_TRY(hipSetDevice(e->device));
	hipStream_t s = e->stream;
	const bool sal = e->last.want_sal != 0;
	CS_TRY(e->d_sel.reserve((size_t)n_sel + 1)); CS_TRY(e->d_sel_moff.reserve((size_t)n_sel + 2)); CS_TRY(e->d_sel_soff.reserve((size_t)n_sel + 2));
	CS_TRY(e->d_tmp.reserve(((size_t)n_sel + 2) * 16 + 1024));
	uint64_t *cm = (uint64_t *)e->d_tmp.p, *cs = cm + n_sel + 1;
	if (n_sel) HIP_TRY(hipMemcpyAsync(e->d_sel.p, read_ids, (size_t)n_sel * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(e->d_ctr.p + 4, 0, sizeof(unsigned long long), s));
	hipLaunchKernelGGL(sel_counts_kernel, dim3(grid_for(n_sel + 1, 256)), dim3(256), 0, s, (const uint64_t *)e->d_sel.p, n_sel, (uint64_t)e->last.n_reads,
	                   (const uint64_t *)e->d_mem_off.p, sal ? (const uint64_t *)e->d_seed_off.p : nullptr, cm, sal ? cs : nullptr, e->d_ctr.p + 4);
	{
		size_t tb = 0;
		HIP_TRY(rocprim::exclusive_scan(nullptr, tb, cm, e->d_sel_moff.p, (uint64_t)0, (size_t)n_sel + 1, rocprim::plus<uint64_t>(), s));
		CS_TRY(e->d_tmp2.reserve(tb + 16));
		HIP_TRY(rocprim::exclusive_scan((void *)e->d_tmp2.p, tb, cm, e->d_sel_moff.p, (uint64_t)0, (size_t)n_sel + 1, rocprim::plus<uint64_t>(), s));
		if (sal) HIP_TRY(rocprim::exclusive_scan((void *)e->d_tmp2.p, tb, cs, e->d_sel_soff.p, (uint64_t)0, (size_t)n_sel + 1, rocprim::plus<uint64_t>(), s));
	}
	CS_TRY(e->h_mem_off.reserve((size_t)n_sel + 1));
	HIP_TRY(hipMemcpyAsync(e->h_mem_off.p, e->d_sel_moff.p, ((size_t)n_sel + 1) * 8, hipMemcpyDeviceToHost, s));
	if (sal) { CS_TRY(e->h_seed_off.reserve((size_t)n_sel + 1)); HIP_TRY(hipMemcpyAsync(e->h_seed_off.p, e->d_sel_soff.p, ((size_t)n_sel + 1) * 8, hipMemcpyDeviceToHost, s)); }
	HIP_TRY(hipMemcpyAsync(e->h_ctr.p + 4, e->d_ctr.p + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if (e->h_ctr.p[4]) return fail(CS_EINVAL, "cs_engine_gather_reads: read id out of range");
	const uint64_t nm = e->h_mem_off.p[n_sel], ns = sal ? e->h_seed_off.p[n_sel] : 0;
	CS_TRY(e->d_sel_mems.reserve((size_t)nm + 1)); CS_TRY(e->h_mems.reserve((size_t)nm + 1));
	if (sal) { CS_TRY(e->d_sel_seeds.reserve((size_t)ns + 1)); CS_TRY(e->h_seeds.reserve((size_t)ns + 1)); }
	if (n_sel) hipLaunchKernelGGL(sel_copy_kernel, dim3(grid_for(n_sel * 16, 256)), dim3(256), 0, s, (const uint64_t *)e->d_sel.p, n_sel, (const uint64_t *)e->d_mem_off.p,
	                              (const uint64_t *)e->d_seed_off.p, (const OutMem *)e->d_mems.p, sal ? (const OutSeed *)e->d_seeds.p : nullptr,
	                              (const uint64_t *)e->d_sel_moff.p, (const uint64_t *)e->d_sel_soff.p, e->d_sel_mems.p, e->d_sel_seeds.p);
	HIP_TRY(hipGetLastError());
	if (nm) HIP_TRY(hipMemcpyAsync(e->h_mems.p, e->d_sel_mems.p, (size_t)nm * sizeof(OutMem), hipMemcpyDeviceToHost, s));
	if (ns) HIP_TRY(hipMemcpyAsync(e->h_seeds.p, e->d_sel_seeds.p, (size_t)ns * sizeof(OutSeed), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	out->n_reads = n_sel; out->n_mems = nm; out->n_seeds = ns;
	out->mem_off = e->h_mem_off.p; out->mems = (const cs_intv_t *)e->h_mems.p;
	out->seed_off = sal ? e->h_seed_off.p : nullptr; out->seeds = sal ? (const cs_seed_t *)e->h_seeds.p : nullptr;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ index validation at the size it is used
// The index a 3.1 Gbp engine runs on is built on the GPU in the same process (index_build.hip), and the arrays the shortcuts read are derived
// from it at engine creation; the byte-for-byte comparisons with bwaidx stop at 64 Mbp.  This check is independent of how any of it was
// made: (1) the recovered 2-bit text equals the caller's genome and its reverse complement; (2) every pair of neighbouring rows of the
// full suffix array is in suffix order, decided by comparing the TEXT (end of text smallest, as the sentinel); (3) ISA[SA[r]] = r, so
// SA is a permutation; (4) the BWT character of row r is T[SA[r] - 1] and the row of suffix 0 is `primary`; (5) the sampled suffix
// array of the file equals the full one at the sampled rows.  (1)-(3) make SA THE suffix array of the given text, (4)-(5) tie the
// reference's two files to it (FM_index/bwt.c:62-96, index_main.c:152-174).
__global__ void check_text_kernel(const DevIndex ix, const uint8_t *fwd, uint64_t l_pac, unsigned long long *bad)
{
	unsigned long long c = 0;
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < l_pac; p += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t q = 2 * l_pac - 1 - p;                       // the position of base p on the reverse-complement strand
		const uint32_t f = fwd[p] & 3u;
		const uint32_t a = (ix.text2[p >> 4] >> ((p & 15) << 1)) & 3u, b = (ix.text2[q >> 4] >> ((q & 15) << 1)) & 3u;
		c += (a != f) + (b != 3u - f);
	}
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(bad, c);
}
__global__ void check_rows_kernel(const DevIndex ix, uint32_t cap, unsigned long long *out /* [0] order [1] isa [2] bwt [3] sampled SA [4] undecided (LCP beyond cap) */)
{
	unsigned long long v[5] = {0, 0, 0, 0, 0};
	const uint32_t *t2 = ix.text2;
	auto base = [&](uint64_t p) { return (t2[p >> 4] >> ((p & 15) << 1)) & 3u; };
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; r <= ix.seq_len; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t a = r == 1 ? ix.seq_len : sa_direct(ix, r - 1), b = sa_direct(ix, r);
		if (b >= ix.seq_len || a > ix.seq_len) { ++v[0]; continue; }
		const uint32_t l = text_lcp(ix, a, b, cap);
		if (l >= cap) ++v[4];
		else if (!(a + l == ix.seq_len || (b + l < ix.seq_len && base(a + l) < base(b + l)))) ++v[0];
		if (isa_direct(ix, b) != r) ++v[1];
		if (b == 0) { if (r != ix.primary) ++v[2]; }
		else {
			if (r == ix.primary) ++v[2];
			else {
				const uint64_t row = r - (r > ix.primary);
				const Block k = load_block(ix, row >> OCC_SHIFT);
				const uint32_t p = (uint32_t)row & OCC_MASK, w = p >> 5, bit = p & 31;
				const uint32_t lo = w == 0 ? k.pl.x : k.pl.y, hi = w == 0 ? k.pl.z : k.pl.w;
				if ((((lo >> bit) & 1u) | (((hi >> bit) & 1u) << 1)) != base(b - 1)) ++v[2];
			}
		}
		if ((r & ix.sa_mask) == 0 && ix.sa[r >> ix.sa_shift] != b) ++v[3];
	}
	for (int i = 0; i < 5; ++i) {
		unsigned long long c = v[i];
		for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
		if ((threadIdx.x & 63) == 0 && c) atomicAdd(out + i, c);
	}
}
extern "C" int cs_engine_check_index(cs_engine_t *e, const uint8_t *d_fwd_nt4, uint64_t l_pac, cs_index_check_t *out)
{
	if (!e || !out) return fail(CS_EINVAL, "cs_engine_check_index: null argument");
	if (pipe_busy(e)) return fail(CS_EINVAL, "cs_engine_check_index: submitted batches are in flight, collect them first");
	if (!e->ix.text2 || !(e->ix.fsa32 || e->ix.fsa64)) return fail(CS_EINVAL, "cs_engine_check_index: needs the full suffix array and the text arrays (engine options full_sa, text_mode)");
	if (d_fwd_nt4 && 2 * l_pac != e->ix.seq_len) return fail(CS_EINVAL, "cs_engine_check_index: l_pac is not half of the index length");
	HIP_TRY(hipSetDevice(e->device));
	hipStream_t s = e->stream;
	HIP_TRY(hipMemsetAsync(e->d_ctr.p, 0, 8 * sizeof(unsigned long long), s));
	const unsigned grid = (unsigned)e->n_cu * 16;
	if (d_fwd_nt4) hipLaunchKernelGGL(check_text_kernel, dim3(grid), dim3(256), 0, s, e->ix, d_fwd_nt4, l_pac, e->d_ctr.p + 5);
	hipLaunchKernelGGL(check_rows_kernel, dim3(grid), dim3(256), 0, s, e->ix, 1u << 20, e->d_ctr.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_ctr.p, e->d_ctr.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	out->rows_checked = e->ix.seq_len; out->order_violations = e->h_ctr.p[0]; out->isa_violations = e->h_ctr.p[1]; out->bwt_violations = e->h_ctr.p[2];
	out->sampled_sa_violations = e->h_ctr.p[3]; out->undecided_rows = e->h_ctr.p[4]; out->text_violations = e->h_ctr.p[5]; out->text_checked = d_fwd_nt4 ? 1 : 0;
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ primitives (tests)
template <typename In, typename Out, typename Launch>
static int run_prim(cs_engine *e, int64_t n, const In *h_in, size_t in_per, Out *h_out, size_t out_per, const uint8_t *h_flag, Launch launch)
{
	if (!e || n < 0 || (n > 0 && (!h_in || !h_out))) return fail(CS_EINVAL, "bad argument");
	if (n == 0) return CS_OK;
	HIP_TRY(hipSetDevice(e->device));
	In *d_in = nullptr; Out *d_out = nullptr; uint8_t *d_flag = nullptr;
	HIP_TRY(hipMalloc((void **)&d_in, (size_t)n * in_per * sizeof(In)));
	HIP_TRY(hipMalloc((void **)&d_out, (size_t)n * out_per * sizeof(Out)));
	if (h_flag) { HIP_TRY(hipMalloc((void **)&d_flag, (size_t)n)); HIP_TRY(hipMemcpy(d_flag, h_flag, (size_t)n, hipMemcpyHostToDevice)); }
	HIP_TRY(hipMemcpy(d_in, h_in, (size_t)n * in_per * sizeof(In), hipMemcpyHostToDevice));
	launch(d_in, d_flag, d_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	HIP_TRY(hipMemcpy(h_out, d_out, (size_t)n * out_per * sizeof(Out), hipMemcpyDeviceToHost));
	(void)hipFree(d_in); (void)hipFree(d_out); if (d_flag) (void)hipFree(d_flag);
	return CS_OK;
}

extern "C" int cs_engine_occ4(cs_engine_t *e, int64_t n, const uint64_t *k, uint64_t *cnt4)
{
	return run_prim<uint64_t, uint64_t>(e, n, k, 1, cnt4, 4, nullptr, [&](const uint64_t *di, const uint8_t *, uint64_t *dout) {
		hipLaunchKernelGGL(occ4_kernel, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, e->ix, di, dout, n);
	});
}
extern "C" int cs_engine_extend(cs_engine_t *e, int64_t n, const cs_intv_t *ik, const uint8_t *is_back, cs_intv_t *ok4)
{
	if (n > 0 && !is_back) return fail(CS_EINVAL, "is_back is null");
	return run_prim<OutMem, OutMem>(e, n, (const OutMem *)ik, 1, (OutMem *)ok4, 4, is_back, [&](const OutMem *di, const uint8_t *df, OutMem *dout) {
		hipLaunchKernelGGL(extend_kernel, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, e->ix, di, df, dout, n);
	});
}
extern "C" int cs_engine_sa(cs_engine_t *e, int64_t n, const uint64_t *k, uint64_t *sa)
{
	if (e) for (int64_t i = 0; i < n; ++i) if (k && k[i] > e->ix.seq_len) return fail(CS_EINVAL, "SA row out of range");
	return run_prim<uint64_t, uint64_t>(e, n, k, 1, sa, 1, nullptr, [&](const uint64_t *di, const uint8_t *, uint64_t *dout) {
		hipLaunchKernelGGL(sa_kernel, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, e->ix, di, dout, n);
	});
}

// ------------------------------------------------------------------------------------------------ access-shape micro-benchmark
// Dependent chains of random 64-byte Occ-block reads, one chain per lane, nothing else: the ceiling of this access shape on
// the resident index (SURVEY 8d asks for it next to the roofline).  Returns lines per second.
__global__ void random_block_chain_kernel(const DevIndex ix, uint32_t steps, uint64_t *sink)
{
	uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	for (uint32_t i = 0; i < steps; ++i) {
		uint64_t b = (k >> 11) % (2 * ix.n_blocks); // 32-byte records
		Block blk = load_block(ix, b);
		k = k * 6364136223846793005ull + (blk.cnt.x ^ blk.cnt.w ^ blk.pl.y ^ blk.pl.z) + 1442695040888963407ull;
	}
	if (k == 42) *sink = k;
}
extern "C" int cs_engine_probe_random_lines(cs_engine_t *e, int waves_per_simd, int steps, double *lines_per_sec)
{
	if (!e || !lines_per_sec || waves_per_simd < 1 || waves_per_simd > 8 || steps < 1) return fail(CS_EINVAL, "bad argument");
	HIP_TRY(hipSetDevice(e->device));
	unsigned blocks = (unsigned)(e->n_cu * waves_per_simd); // 256-thread blocks: 4 waves each => waves_per_simd blocks per CU
	HIP_TRY(hipEventRecord(e->ev[0], e->stream));
	hipLaunchKernelGGL(random_block_chain_kernel, dim3(blocks), dim3(256), 0, e->stream, e->ix, (uint32_t)steps, (uint64_t *)e->d_ctr.p + 7);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e->ev[1], e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
	*lines_per_sec = (double)blocks * 256.0 * steps / (ms * 1e-3);
	return CS_OK;
}

// ------------------------------------------------------------------------------------------------ device memory helpers
extern "C" int cs_device_alloc(cs_engine_t *e, size_t bytes, void **dptr)
{
	if (!e || !dptr) return fail(CS_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
	return CS_OK;
}
extern "C" int cs_device_free(cs_engine_t *e, void *dptr)
{
	if (!e) return fail(CS_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipFree(dptr));
	return CS_OK;
}
extern "C" int cs_device_upload(cs_engine_t *e, void *dst, const void *src, size_t bytes)
{
	if (!e || (bytes && (!dst || !src))) return fail(CS_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return CS_OK;
}
extern "C" int cs_device_download(cs_engine_t *e, void *dst, const void *src, size_t bytes)
{
	if (!e || (bytes && (!dst || !src))) return fail(CS_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return CS_OK;
}
extern "C" int cs_device_sync(cs_engine_t *e)
{
	if (!e) return fail(CS_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return CS_OK;
}
