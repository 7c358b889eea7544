// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Seed-level driver around the REAL reference (compiled from /root/reference by oracle/Makefile `make ref`).
// The reference prints only SAM, so this harness calls its seeding functions directly and dumps, per read,
// the sorted mem list and the SAL seed list, plus the reference's own counters.  Two paths are run and
// compared read by read:
//   path A (BWA-MEM, uncached)  : bwt_smem1 / bwt_seed_strategy1 (FM_index/bwt.h:123-126) driven by the three
//                                 loops of mem_collect_intv (mapping/bwamem.c:218-272; that function is static)
//   path B (CompSeed, SST cache): collect_mem_with_sst / tem_forward_sst (mapping/comp_seed.cpp:67,141) driven by
//                                 the loops of seed_and_extend (mapping/comp_seed.cpp:2255-2302), SSTs cleared
//                                 every BATCH_SIZE reads (comp_seed.cpp:2254)
// SAL follows comp_seed.cpp:2306-2347 (slot sampling, per-batch dedup, bwt_sa).
//
// Output file (little endian):
//   char magic[8]="CSGOLD1"; u64 n_reads, n_mems, n_seeds, bwt_queries, bwt_calls, sal_queries, sal_calls, n_diff_AB
//   u64 mem_off[n_reads+1];  {u64 x0,x1,x2,info} mems[n_mems]
//   u64 seed_off[n_reads+1]; {i64 rbeg; i32 qbeg; i32 len} seeds[n_seeds]
// With --prim FILE N SEED it also dumps known-answer vectors for the primitives (format below at dump_prims).

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include "bwalib/bwa.h"
#include "FM_index/bwt.h"
#include "FM_index/bntseq.h"
#include "mapping/comp_seed.h"

thread_aux_t tprof; // comp_seed.cpp:22 declares it extern

int collect_mem_with_sst(const uint8_t *seq, int len, int pivot, int min_hits, thread_aux_t &aux);
int tem_forward_sst(const mem_opt_t *opt, const uint8_t *seq, int len, int start, bwtintv_t *mem, thread_aux_t &aux);

// mem_chain (mapping/comp_seed.cpp:241) has external linkage but its return type lives in the .cpp: same layout declared here
typedef struct { size_t n, m; mem_chain_t *a; } mem_chain_v;
mem_chain_v mem_chain(const mem_opt_t *opt, const bntseq_t *bns, int len, const std::vector<bwtintv_t> &mem, const std::vector<mem_seed_t> &seed);

// the stage behind chaining (--aln): the reference's own chain filters and its batched extension driver, all with external linkage
// (mapping/comp_seed.cpp:297, 393, 1319).  mem_cache is defined in the .cpp (comp_seed.cpp:1124-1148): the same layout is declared here so
// that the harness can allocate the scratch the function expects exactly as memoryAlloc does (comp_seed.cpp:2444-2491).
#include "mapping/bandedSWA.h"
#include "mapping/macro.h"
int mem_chain_flt(const mem_opt_t *opt, int n_chn, mem_chain_t *a);
void mem_flt_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, int n_chn, mem_chain_t *a);
typedef struct {
	SeqPair *seqPairArrayAux[MAX_THREADS], *seqPairArrayLeft128[MAX_THREADS], *seqPairArrayRight128[MAX_THREADS];
	int64_t wsize[MAX_THREADS];
	int64_t wsize_buf_ref[MAX_THREADS * CACHE_LINE], wsize_buf_qer[MAX_THREADS * CACHE_LINE];
	uint8_t *seqBufLeftRef[MAX_THREADS * CACHE_LINE], *seqBufRightRef[MAX_THREADS * CACHE_LINE], *seqBufLeftQer[MAX_THREADS * CACHE_LINE], *seqBufRightQer[MAX_THREADS * CACHE_LINE];
	int32_t *lim[MAX_THREADS];
	int64_t wsize_mem[MAX_THREADS];
} mem_cache;
void mem_chain2aln_across_reads_V2(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *seq_, int nseq, mem_chain_v *chain_ar,
                                   mem_alnreg_v *av_v, mem_cache *mmc, int tid);
int mem_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, int n, mem_alnreg_t *a); // (comp_seed.cpp:629)

static bool by_info(const bwtintv_t &a, const bwtintv_t &b) { return a.info < b.info; }

// mapping/bwamem.c:218-272 replayed with the public bwt.h entry points
static void path_bwamem(const mem_opt_t *opt, const bwt_t *bwt, int len, const uint8_t *seq,
                        std::vector<bwtintv_t> &out, bwtintv_v *tmpv[2], bwtintv_v *mem1)
{
	int x = 0;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499); // float product, as bwamem.c:223
	out.clear();
	while (x < len) {
		if (seq[x] < 4) {
			x = bwt_smem1(bwt, len, seq, x, 1, mem1, tmpv);
			for (size_t i = 0; i < mem1->n; ++i) {
				bwtintv_t *p = &mem1->a[i];
				int slen = (uint32_t)p->info - (p->info >> 32);
				if (slen >= opt->min_seed_len) out.push_back(*p);
			}
		} else ++x;
	}
	size_t old_n = out.size();
	for (size_t k = 0; k < old_n; ++k) {
		bwtintv_t p = out[k];
		int start = p.info >> 32, end = (int32_t)p.info;
		if (end - start < split_len || p.x[2] > (uint64_t)opt->split_width) continue;
		bwt_smem1(bwt, len, seq, (start + end) >> 1, p.x[2] + 1, mem1, tmpv);
		for (size_t i = 0; i < mem1->n; ++i)
			if ((uint32_t)mem1->a[i].info - (mem1->a[i].info >> 32) >= (uint32_t)opt->min_seed_len)
				out.push_back(mem1->a[i]);
	}
	if (opt->max_mem_intv > 0) {
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				bwtintv_t m;
				x = bwt_seed_strategy1(bwt, len, seq, x, opt->min_seed_len, opt->max_mem_intv, &m);
				if (m.x[2] > 0) out.push_back(m);
			} else ++x;
		}
	}
	std::stable_sort(out.begin(), out.end(), by_info);
}

// mapping/comp_seed.cpp:2262-2301 replayed on one read
static void path_compseed(const mem_opt_t *opt, int len, const uint8_t *seq, thread_aux_t &aux, std::vector<bwtintv_t> &match)
{
	match.clear();
	for (int j = 0; j < len; ) {
		j = collect_mem_with_sst(seq, len, j, 1, aux);
		for (const auto &m : aux.super_mem)
			if ((int)m.info - (int)(m.info >> 32) >= opt->min_seed_len) match.push_back(m);
	}
	int old_n = (int)match.size();
	for (int j = 0; j < old_n; j++) {
		const bwtintv_t p = match[j];
		int beg = p.info >> 32, end = (int)p.info;
		if (end - beg < (int)(1.0 * opt->min_seed_len * opt->split_factor + .499) or p.x[2] > (uint64_t)opt->split_width) continue;
		collect_mem_with_sst(seq, len, (beg + end) / 2, p.x[2] + 1, aux);
		for (const auto &m : aux.super_mem)
			if ((int)m.info - (int)(m.info >> 32) >= opt->min_seed_len) match.push_back(m);
	}
	if (opt->max_mem_intv > 0) {
		for (int j = 0; j < len; ) {
			if (seq[j] < 4) {
				bwtintv_t m;
				j = tem_forward_sst(opt, seq, len, j, &m, aux);
				if (m.x[2] > 0) match.push_back(m);
			} else j++;
		}
	}
	std::sort(match.begin(), match.end(), by_info);
}

struct seed_out_t { int64_t rbeg; int32_t qbeg, len; };

static uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// Known-answer vectors for the primitives.  File:
//   char magic[8]="CSPRIM1"; u64 n_occ4, n_2occ4, n_ext, n_sa
//   occ4 : n x {u64 k; u64 cnt[4]}                              bwt_occ4   (FM_index/bwt.c:169)
//   2occ4: n x {u64 k,l; u64 cntk[4], cntl[4]}                  bwt_2occ4  (bwt.c:189)
//   ext  : n x {u64 x0,x1,x2; u64 is_back; u64 ok[4][3]}        bwt_extend (bwt.c:262)
//   sa   : n x {u64 k; u64 sa}                                  bwt_sa     (bwt.c:86)
static void dump_prims(const bwt_t *bwt, const char *fn, long n, uint64_t seed, const std::vector<bwtintv_t> &pool)
{
	FILE *fp = fopen(fn, "wb");
	if (!fp) { perror(fn); exit(1); }
	uint64_t hdr[4] = {(uint64_t)n, (uint64_t)n, (uint64_t)n, (uint64_t)n};
	fwrite("CSPRIM1", 1, 8, fp); fwrite(hdr, 8, 4, fp);
	uint64_t N = bwt->seq_len, P = bwt->primary;
	auto pick_k = [&](long i) -> uint64_t {
		switch (i) { // edge rows first
			case 0: return (uint64_t)-1; case 1: return 0; case 2: return P; case 3: return P - 1; case 4: return P + 1;
			case 5: return N; case 6: return N - 1; case 7: return 127; case 8: return 128; case 9: return 129;
			default: return splitmix(seed) % (N + 1);
		}
	};
	for (long i = 0; i < n; i++) {
		uint64_t rec[5]; rec[0] = pick_k(i);
		bwt_occ4(bwt, rec[0], rec + 1);
		fwrite(rec, 8, 5, fp);
	}
	for (long i = 0; i < n; i++) {
		uint64_t rec[10];
		uint64_t k = pick_k(i);
		uint64_t span = (i & 1) ? splitmix(seed) % 300 : splitmix(seed) % (N + 1);
		uint64_t l = (k == (uint64_t)-1) ? splitmix(seed) % (N + 1) : k + span;
		if (l > N) l = N;
		rec[0] = k; rec[1] = l;
		bwt_2occ4(bwt, k, l, rec + 2, rec + 6);
		fwrite(rec, 8, 10, fp);
	}
	for (long i = 0; i < n; i++) {
		bwtintv_t ik, ok[4];
		if (!pool.empty() && (i % 3) != 0) ik = pool[splitmix(seed) % pool.size()];
		else { int c0 = (int)(splitmix(seed) & 3); bwt_set_intv(bwt, c0, ik); } // macro evaluates its argument 4x
		uint64_t is_back = splitmix(seed) & 1;
		memset(ok, 0, sizeof(ok));
		bwt_extend(bwt, &ik, ok, (int)is_back);
		uint64_t rec[16] = {ik.x[0], ik.x[1], ik.x[2], is_back};
		for (int c = 0; c < 4; c++) for (int j = 0; j < 3; j++) rec[4 + c * 3 + j] = ok[c].x[j];
		fwrite(rec, 8, 16, fp);
	}
	for (long i = 0; i < n; i++) {
		uint64_t rec[2];
		uint64_t k = pick_k(i + 1); // k = -1 is not a valid row
		rec[0] = k; rec[1] = bwt_sa(bwt, k);
		fwrite(rec, 8, 2, fp);
	}
	fclose(fp);
}

// --time T: the reference's own seeding + SAL code (CompSeed flow: collect_mem_with_sst / tem_forward_sst with SSTs cleared per 512 reads,
// then the SAL block of comp_seed.cpp:2306-2347) on T threads, each with its own thread_aux_t, 512-read batches handed out by a counter --
// what kt_for does in mem_process_seqs (comp_seed.cpp:2541-2548), without chaining / extension / SAM.  Prints reads per second.
static int time_mode(const mem_opt_t *opt, const bwt_t *bwt, const std::vector<std::string> &reads, int T)
{
	const size_t n = reads.size(), nb = (n + BATCH_SIZE - 1) / BATCH_SIZE;
	std::atomic<size_t> next(0);
	std::atomic<uint64_t> tot_mems(0), tot_seeds(0);
	auto work = [&]() {
		thread_aux_t aux;
		aux.forward_sst = new SST(bwt); aux.backward_sst = new SST(bwt);
		std::vector<bwtintv_t> B; std::vector<std::vector<bwtintv_t>> match(BATCH_SIZE);
		uint64_t nm = 0, ns = 0;
		for (size_t b = next++; b < nb; b = next++) {
			const size_t b0 = b * BATCH_SIZE, b1 = std::min(n, b0 + (size_t)BATCH_SIZE);
			aux.forward_sst->clear(); aux.backward_sst->clear();
			for (size_t r = b0; r < b1; r++) {
				std::vector<uint8_t> seq(reads[r].size() + 1, 4);
				const int len = (int)reads[r].size();
				for (int j = 0; j < len; j++) seq[j] = nst_nt4_table[(uint8_t)reads[r][j]];
				path_compseed(opt, len, seq.data(), aux, match[r - b0]);
				nm += match[r - b0].size();
			}
			std::vector<sal_request_t> uniq; std::vector<uint64_t> slots;
			for (size_t r = b0; r < b1; r++)
				for (const auto &m : match[r - b0]) {
					uint64_t step = m.x[2] > (uint64_t)opt->max_occ ? m.x[2] / opt->max_occ : 1;
					for (uint64_t k = 0, count = 0; k < m.x[2] && count < (uint64_t)opt->max_occ; k += step, count++) { slots.push_back(m.x[0] + k); uniq.emplace_back(sal_request_t(m.x[0] + k)); }
				}
			std::sort(uniq.begin(), uniq.end());
			size_t sz = 0;
			for (size_t i = 0; i < uniq.size(); i++) if (i == 0 or uniq[i - 1].que_location != uniq[i].que_location) uniq[sz++] = uniq[i];
			uniq.resize(sz);
			for (uint64_t q : slots) {
				auto k = std::lower_bound(uniq.begin(), uniq.end(), sal_request_t(q));
				if (k->coordinate == (uint64_t)-1) k->coordinate = bwt_sa(bwt, q);
			}
			ns += slots.size();
		}
		tot_mems += nm; tot_seeds += ns;
		delete aux.forward_sst; delete aux.backward_sst;
	};
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++) th.emplace_back(work);
	for (auto &t : th) t.join();
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("{\"reads\": %zu, \"threads\": %d, \"seconds\": %.3f, \"reads_per_s\": %.1f, \"mems\": %lu, \"seeds\": %lu}\n", n, T, sec, n / sec, (unsigned long)tot_mems.load(), (unsigned long)tot_seeds.load());
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 4) {
		fprintf(stderr, "usage: ref_dump <idx prefix> <reads.txt> <out.bin> [-k INT] [-r FLOAT] [-y INT] [-c INT] [-s INT] [-B batch] [--prim FILE N SEED]\n");
		return 1;
	}
	mem_opt_t *opt = mem_opt_init();
	int batch = BATCH_SIZE;
	const char *prim_fn = 0, *chain_fn = 0, *aln_fn = 0, *dedup_fn = 0; long prim_n = 0; uint64_t prim_seed = 1; int time_threads = 0;
	for (int i = 4; i < argc; i++) {
		std::string a = argv[i];
		if (a == "-k") opt->min_seed_len = atoi(argv[++i]);
		else if (a == "-r") opt->split_factor = atof(argv[++i]);
		else if (a == "-y") opt->max_mem_intv = atol(argv[++i]);
		else if (a == "-c") opt->max_occ = atoi(argv[++i]);
		else if (a == "-s") opt->split_width = atoi(argv[++i]);
		else if (a == "-B") batch = atoi(argv[++i]);
		else if (a == "--prim") { prim_fn = argv[++i]; prim_n = atol(argv[++i]); prim_seed = strtoull(argv[++i], 0, 10); }
		else if (a == "--chains") chain_fn = argv[++i];
		else if (a == "--aln") aln_fn = argv[++i];
		else if (a == "--dedup") dedup_fn = argv[++i];   // with --aln: also the regions mem_sort_dedup_patch leaves (comp_seed.cpp:2385-2395)
		else if (a == "--time") time_threads = atoi(argv[++i]);
		else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
	}
	if (batch < 1 || batch > BATCH_SIZE) { fprintf(stderr, "batch must be in [1,%d]\n", BATCH_SIZE); return 1; }
	bwa_verbose = 1;
	bwaidx_t *idx = bwa_idx_load(argv[1], aln_fn ? BWA_IDX_ALL : chain_fn ? (BWA_IDX_BWT | BWA_IDX_BNS) : BWA_IDX_BWT);
	if (!idx) { fprintf(stderr, "cannot load index %s\n", argv[1]); return 1; }
	const bwt_t *bwt = idx->bwt;

	std::vector<std::string> reads;
	{
		FILE *fp = fopen(argv[2], "r");
		if (!fp) { perror(argv[2]); return 1; }
		std::string cur; int ch;
		while ((ch = fgetc(fp)) != EOF) { if (ch == '\n') { reads.push_back(cur); cur.clear(); } else cur.push_back((char)ch); }
		if (!cur.empty()) reads.push_back(cur);
		fclose(fp);
	}
	size_t n = reads.size();
	if (time_threads > 0) return time_mode(opt, bwt, reads, time_threads);

	thread_aux_t aux;
	aux.forward_sst = new SST(bwt);
	aux.backward_sst = new SST(bwt);
	bwtintv_v tv0 = {0, 0, 0}, tv1 = {0, 0, 0}, mem1 = {0, 0, 0};
	bwtintv_v *tmpv[2] = {&tv0, &tv1};

	std::vector<uint64_t> mem_off(n + 1, 0), seed_off(n + 1, 0);
	std::vector<bwtintv_t> mems, A, B;
	std::vector<seed_out_t> seeds;
	uint64_t n_diff = 0, sal_queries = 0, sal_calls = 0, bwt_calls = 0;

	for (size_t b0 = 0; b0 < n; b0 += batch) {
		size_t b1 = std::min(n, b0 + (size_t)batch);
		aux.forward_sst->clear(); aux.backward_sst->clear();
		for (size_t r = b0; r < b1; r++) {
			std::vector<uint8_t> seq(reads[r].size() + 1, 4);
			int len = (int)reads[r].size();
			for (int j = 0; j < len; j++) seq[j] = nst_nt4_table[(uint8_t)reads[r][j]];
			path_bwamem(opt, bwt, len, seq.data(), A, tmpv, &mem1);
			path_compseed(opt, len, seq.data(), aux, B);
			bool same = A.size() == B.size();
			for (size_t i = 0; same && i < A.size(); i++)
				same = A[i].x[0] == B[i].x[0] && A[i].x[1] == B[i].x[1] && A[i].x[2] == B[i].x[2] && A[i].info == B[i].info;
			if (!same) n_diff++;
			mems.insert(mems.end(), B.begin(), B.end());
			mem_off[r + 1] = mems.size();
		}
		bwt_calls += aux.forward_sst->bwt_call + aux.backward_sst->bwt_call;
		// SAL, comp_seed.cpp:2306-2347
		std::vector<sal_request_t> uniq;
		size_t s0 = seeds.size();
		for (size_t r = b0; r < b1; r++) {
			for (uint64_t i = mem_off[r]; i < mem_off[r + 1]; i++) {
				const bwtintv_t &m = mems[i];
				uint64_t step = m.x[2] > (uint64_t)opt->max_occ ? m.x[2] / opt->max_occ : 1;
				for (uint64_t k = 0, count = 0; k < m.x[2] && count < (uint64_t)opt->max_occ; k += step, count++) {
					seed_out_t s; s.qbeg = m.info >> 32; s.len = (int)m.info - (int)(m.info >> 32); s.rbeg = m.x[0] + k;
					seeds.push_back(s);
					uniq.emplace_back(sal_request_t(m.x[0] + k));
					sal_queries++;
				}
			}
			seed_off[r + 1] = seeds.size();
		}
		std::sort(uniq.begin(), uniq.end());
		size_t sz = 0;
		for (size_t i = 0; i < uniq.size(); i++) if (i == 0 or uniq[i - 1].que_location != uniq[i].que_location) uniq[sz++] = uniq[i];
		uniq.resize(sz);
		for (size_t i = s0; i < seeds.size(); i++) {
			auto k = std::lower_bound(uniq.begin(), uniq.end(), sal_request_t(seeds[i].rbeg));
			if (k->coordinate == (uint64_t)-1) { k->coordinate = bwt_sa(bwt, seeds[i].rbeg); sal_calls++; }
			seeds[i].rbeg = k->coordinate;
		}
	}

	FILE *fp = fopen(argv[3], "wb");
	if (!fp) { perror(argv[3]); return 1; }
	uint64_t hdr[8] = {n, mems.size(), seeds.size(), (uint64_t)aux.bwt_query_times, bwt_calls, sal_queries, sal_calls, n_diff};
	fwrite("CSGOLD1", 1, 8, fp);
	fwrite(hdr, 8, 8, fp);
	fwrite(mem_off.data(), 8, n + 1, fp);
	fwrite(mems.data(), sizeof(bwtintv_t), mems.size(), fp);
	fwrite(seed_off.data(), 8, n + 1, fp);
	fwrite(seeds.data(), sizeof(seed_out_t), seeds.size(), fp);
	fclose(fp);
	fprintf(stderr, "[ref_dump] reads=%zu mems=%zu seeds=%zu bwt_queries=%ld bwt_calls=%lu sal_queries=%lu sal_calls=%lu diffAB=%lu\n",
	        n, mems.size(), seeds.size(), aux.bwt_query_times, (unsigned long)bwt_calls, (unsigned long)sal_queries, (unsigned long)sal_calls, (unsigned long)n_diff);
	if (prim_fn) dump_prims(bwt, prim_fn, prim_n, prim_seed, mems);
	if (chain_fn) { // the reference's own mem_chain on every read's mems and seeds (comp_seed.cpp:2361): chains in its traversal order
		// file: char magic[8]="CSCHAIN1"; u64 n_reads, n_chains, n_seeds; u64 chain_off[n_reads+1];
		//       {i64 pos; i32 rid, n; f32 frac_rep; i32 is_alt} chains[]; {i64 rbeg; i32 qbeg, len} seeds[] (chain after chain)
		struct chain_out_t { int64_t pos; int32_t rid, n; float frac_rep; int32_t is_alt; };
		std::vector<uint64_t> chain_off(n + 1, 0);
		std::vector<chain_out_t> chains; std::vector<seed_out_t> cseeds;
		for (size_t r = 0; r < n; r++) {
			std::vector<bwtintv_t> match(mems.begin() + mem_off[r], mems.begin() + mem_off[r + 1]);
			std::vector<mem_seed_t> sd;
			for (uint64_t i = seed_off[r]; i < seed_off[r + 1]; i++) {
				mem_seed_t s; memset(&s, 0, sizeof s);
				s.rbeg = seeds[i].rbeg; s.qbeg = seeds[i].qbeg; s.len = s.score = seeds[i].len;
				sd.push_back(s);
			}
			mem_chain_v cv = mem_chain(opt, idx->bns, (int)reads[r].size(), match, sd);
			for (size_t c = 0; c < cv.n; c++) {
				chain_out_t o = {cv.a[c].pos, cv.a[c].rid, cv.a[c].n, cv.a[c].frac_rep, (int32_t)cv.a[c].is_alt};
				chains.push_back(o);
				for (int j = 0; j < cv.a[c].n; j++) { seed_out_t s = {cv.a[c].seeds[j].rbeg, cv.a[c].seeds[j].qbeg, cv.a[c].seeds[j].len}; cseeds.push_back(s); }
				free(cv.a[c].seeds);
			}
			free(cv.a);
			chain_off[r + 1] = chains.size();
		}
		FILE *fc = fopen(chain_fn, "wb");
		if (!fc) { perror(chain_fn); return 1; }
		uint64_t h3[3] = {n, chains.size(), cseeds.size()};
		fwrite("CSCHAIN1", 1, 8, fc); fwrite(h3, 8, 3, fc);
		fwrite(chain_off.data(), 8, n + 1, fc);
		fwrite(chains.data(), sizeof(chain_out_t), chains.size(), fc);
		fwrite(cseeds.data(), sizeof(seed_out_t), cseeds.size(), fc);
		fclose(fc);
		fprintf(stderr, "[ref_dump] chains=%zu chained seeds=%zu\n", chains.size(), cseeds.size());
	}
	if (aln_fn) { // the reference's own mem_chain -> mem_chain_flt -> mem_flt_chained_seeds -> mem_chain2aln_across_reads_V2 (comp_seed.cpp:2361-2374),
		// BATCH_SIZE reads per call as seed_and_extend hands them over.  Dumped: the chains that go INTO the extension driver (after both filters)
		// and every alignment region it leaves in av_v, purged ones (qb = qe = -1) included.
		// file: "CSALN01\0"; u64 n_reads, n_chains, n_cseeds, n_regs; u64 chain_off[n+1]; {i64 pos; i32 rid, n; f32 frac_rep; i32 is_alt} chains[];
		//       {i64 rbeg; i32 qbeg, len, score, pad} cseeds[]; u64 reg_off[n+1];
		//       {i64 rb, re; i32 qb, qe, rid, score, truesc, w, seedcov, seedlen0; f32 frac_rep; i32 chain} regs[]
		struct chain_out_t { int64_t pos; int32_t rid, n; float frac_rep; int32_t is_alt; };
		struct cseed_out_t { int64_t rbeg; int32_t qbeg, len, score, pad; };
		struct reg_out_t { int64_t rb, re; int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0; float frac_rep; int32_t chain; };
		std::vector<uint64_t> chain_off(n + 1, 0), reg_off(n + 1, 0);
		std::vector<chain_out_t> chains; std::vector<cseed_out_t> cseeds; std::vector<reg_out_t> regs;
		std::vector<uint64_t> ddp_off(n + 1, 0); std::vector<reg_out_t> ddp; // --dedup: same record, `chain` holds n_comp
		mem_cache *mmc = (mem_cache *)calloc(1, sizeof(mem_cache));
		const int64_t wsize = (int64_t)BATCH_SIZE * SEEDS_PER_READ;
		mmc->seqBufLeftRef[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64); mmc->seqBufRightRef[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_REF + MAX_LINE_LEN, 64);
		mmc->seqBufLeftQer[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64); mmc->seqBufRightQer[0] = (uint8_t *)_mm_malloc(wsize * MAX_SEQ_LEN_QER + MAX_LINE_LEN, 64);
		mmc->wsize_buf_ref[0] = wsize * MAX_SEQ_LEN_REF; mmc->wsize_buf_qer[0] = wsize * MAX_SEQ_LEN_QER;
		mmc->seqPairArrayAux[0] = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
		mmc->seqPairArrayLeft128[0] = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
		mmc->seqPairArrayRight128[0] = (SeqPair *)malloc((wsize + MAX_LINE_LEN) * sizeof(SeqPair));
		mmc->wsize[0] = wsize;
		mmc->lim[0] = (int32_t *)_mm_malloc((BATCH_SIZE + 32) * sizeof(int32_t), 64);
		for (size_t b0 = 0; b0 < n; b0 += BATCH_SIZE) {
			const int nb = (int)std::min<size_t>(BATCH_SIZE, n - b0);
			std::vector<std::vector<uint8_t>> codes((size_t)nb);
			std::vector<bseq1_t> sq((size_t)nb);
			std::vector<mem_chain_v> chain_ar((size_t)nb); std::vector<mem_alnreg_v> reg_ar((size_t)nb);
			for (int r = 0; r < nb; r++) {
				const std::string &rd = reads[b0 + r];
				codes[r].assign(rd.size() + 1, 4);
				for (size_t j = 0; j < rd.size(); j++) codes[r][j] = nst_nt4_table[(uint8_t)rd[j]];
				memset(&sq[r], 0, sizeof(bseq1_t)); sq[r].l_seq = (int)rd.size(); sq[r].seq = (char *)codes[r].data();
				std::vector<bwtintv_t> match(mems.begin() + mem_off[b0 + r], mems.begin() + mem_off[b0 + r + 1]);
				std::vector<mem_seed_t> sd;
				for (uint64_t i = seed_off[b0 + r]; i < seed_off[b0 + r + 1]; i++) {
					mem_seed_t s; memset(&s, 0, sizeof s);
					s.rbeg = seeds[i].rbeg; s.qbeg = seeds[i].qbeg; s.len = s.score = seeds[i].len;
					sd.push_back(s);
				}
				mem_chain_v chn = mem_chain(opt, idx->bns, sq[r].l_seq, match, sd);
				chn.n = mem_chain_flt(opt, chn.n, chn.a);
				mem_flt_chained_seeds(opt, idx->bns, idx->pac, sq[r].l_seq, codes[r].data(), chn.n, chn.a);
				chain_ar[r] = chn;
				kv_init(reg_ar[r]);
				for (size_t c = 0; c < chn.n; c++) {
					chain_out_t o = {chn.a[c].pos, chn.a[c].rid, chn.a[c].n, chn.a[c].frac_rep, (int32_t)chn.a[c].is_alt};
					chains.push_back(o);
					for (int j = 0; j < chn.a[c].n; j++) { const mem_seed_t &t = chn.a[c].seeds[j]; cseed_out_t s = {t.rbeg, t.qbeg, t.len, t.score, 0}; cseeds.push_back(s); }
				}
				chain_off[b0 + r + 1] = chains.size();
			}
			mem_chain2aln_across_reads_V2(opt, idx->bns, idx->pac, sq.data(), nb, chain_ar.data(), reg_ar.data(), mmc, 0);
			for (int r = 0; r < nb; r++) {
				for (size_t i = 0; i < reg_ar[r].n; i++) {
					const mem_alnreg_t &a = reg_ar[r].a[i];
					reg_out_t o = {a.rb, a.re, a.qb, a.qe, a.rid, a.score, a.truesc, a.w, a.seedcov, a.seedlen0, a.frac_rep, (int32_t)(a.c - chain_ar[r].a)};
					regs.push_back(o);
				}
				reg_off[b0 + r + 1] = regs.size();
				if (dedup_fn) { // what seed_and_extend does with a read's regions next (comp_seed.cpp:2385-2395): drop the purged ones, then mem_sort_dedup_patch
					mem_alnreg_v &rg = reg_ar[r];
					int m = 0;
					for (size_t i = 0; i < rg.n; ++i) if (rg.a[i].qe > rg.a[i].qb) { if (m != (int)i) rg.a[m++] = rg.a[i]; else ++m; }
					m = mem_sort_dedup_patch(opt, idx->bns, idx->pac, codes[r].data(), m, rg.a);
					for (int i = 0; i < m; ++i) {
						const mem_alnreg_t &a = rg.a[i];
						reg_out_t o = {a.rb, a.re, a.qb, a.qe, a.rid, a.score, a.truesc, a.w, a.seedcov, a.seedlen0, a.frac_rep, a.n_comp};
						ddp.push_back(o);
					}
					ddp_off[b0 + r + 1] = ddp.size();
				}
				for (size_t c = 0; c < chain_ar[r].n; c++) free(chain_ar[r].a[c].seeds);
				free(chain_ar[r].a); free(reg_ar[r].a);
			}
		}
		FILE *fa = fopen(aln_fn, "wb");
		if (!fa) { perror(aln_fn); return 1; }
		uint64_t h4[4] = {n, chains.size(), cseeds.size(), regs.size()};
		fwrite("CSALN01", 1, 8, fa); fwrite(h4, 8, 4, fa);
		fwrite(chain_off.data(), 8, n + 1, fa); fwrite(chains.data(), sizeof(chain_out_t), chains.size(), fa); fwrite(cseeds.data(), sizeof(cseed_out_t), cseeds.size(), fa);
		fwrite(reg_off.data(), 8, n + 1, fa); fwrite(regs.data(), sizeof(reg_out_t), regs.size(), fa);
		fclose(fa);
		if (dedup_fn) { // file: "CSDDP01\0"; u64 n_reads, n_regs; u64 reg_off[n+1]; reg records (chain = n_comp)
			FILE *fd = fopen(dedup_fn, "wb");
			if (!fd) { perror(dedup_fn); return 1; }
			uint64_t h2[2] = {n, ddp.size()};
			fwrite("CSDDP01", 1, 8, fd); fwrite(h2, 8, 2, fd); fwrite(ddp_off.data(), 8, n + 1, fd); fwrite(ddp.data(), sizeof(reg_out_t), ddp.size(), fd);
			fclose(fd);
			fprintf(stderr, "[ref_dump] dedup: regions=%zu\n", ddp.size());
		}
		size_t purged = 0;
		for (const auto &g : regs) purged += g.qb == -1 && g.qe == -1;
		fprintf(stderr, "[ref_dump] extension: chains=%zu seeds=%zu regions=%zu (purged %zu)\n", chains.size(), cseeds.size(), regs.size(), purged);
	}
	return n_diff ? 2 : 0;
}
