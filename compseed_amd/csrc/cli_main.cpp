// cli_main.cpp -- `compseed_amd_cli`: the CompSeed command line for the seeding path, on MI355X.
//
//   compseed_amd_cli [options] <FM-index prefix> <reordered reads>
//
// Same positional arguments and the same seeding flags as `CompSeed` / `bwamem` (main.cpp:146-200, 233-330):
// -t -k -r -y -c -s -K.  Everything downstream of seeding (chaining, banded SW, SAM) is out of scope, so instead of SAM
// the program writes the seeds (--dump-seeds FILE) and prints the reference's exit counters (display_profile,
// main.cpp:203-214).  Flags of the stages that are not here are accepted and ignored so existing command lines run.
// Input: one read per line (input_reorder_reads, main.cpp:36-58) or FASTQ when the first byte is '@' (main.cpp:399-406).
// Chunking follows main.cpp:54,437: a chunk ends at the first even read count that reaches -K bases
// (default 10,000,000 x n_threads).  Chunks are split into contiguous read ranges, one per GPU (--gpus N, default all).
#include "../../include/compseed_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// a chunk as the reader delivers it, and its per-GPU read ranges with offsets rebased to 0 (they must live until the chunk is collected)
struct Chunk { const uint8_t *bases = nullptr; const uint64_t *off = nullptr; int64_t n = 0; uint64_t first_id = 0; std::vector<std::vector<uint64_t>> loff; };

static void usage()
{
	fprintf(stderr,
	        "Usage: compseed_amd_cli [options] <FM-index> <Reordered Reads>\n\n"
	        "Seeding options (same meaning as CompSeed / bwamem):\n"
	        "       -t INT        host threads of the reference; sets the default chunk size (-K) [1]\n"
	        "       -k INT        minimum seed length [19]\n"
	        "       -r FLOAT      look for internal seeds inside a seed longer than {-k} * FLOAT [1.5]\n"
	        "       -y INT        seed occurrence for the 3rd round seeding [20]\n"
	        "       -c INT        skip seeds with more than INT occurrences [500]\n"
	        "       -s INT        re-seed only SMEMs with at most INT occurrences [10]\n"
	        "       -K INT        process INT input bases in each batch [10000000 x -t]\n"
	        "GPU / output options:\n"
	        "       --gpus INT    number of MI355X to shard each batch over [all visible]\n"
	        "       --devices LIST   comma-separated device ordinal of every shard instead (an ordinal may repeat)\n"
	        "       --no-sal      stop after SMEM collection (no suffix-array lookup)\n"
	        "       --dump-seeds FILE   write `M read beg end x0 x1 x2` and `S read qbeg len rbeg` lines\n"
	        "Other CompSeed flags (-w -d -D -W -m -S -P -A -B -O -E -L -U -x -p -R -H -o -j -5 -q -v -T -h -a -C -V -Y -M -I)\n"
	        "belong to stages behind seeding; they are accepted and ignored.\n");
}

int main(int argc, char **argv)
{
	setenv("GPU_MAX_HW_QUEUES", "8", 0); // before the HIP runtime starts: one engine per GPU, each with streams of its own
	cs_params_t par; cs_params_default(&par);
	int n_threads = 1, n_gpus = -1, verbose = 0; long fixed_chunk = 0; const char *dump = nullptr;
	std::vector<int> devices; // --devices: the device ordinal of every shard (an ordinal may repeat: several engines on one GPU)
	std::vector<const char *> pos;
	for (int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto need = [&]() -> const char * { if (i + 1 >= argc) { usage(); exit(1); } return argv[++i]; };
		if (a == "-k") par.min_seed_len = atoi(need());
		else if (a == "-r") par.split_factor = (float)atof(need());
		else if (a == "-y") par.max_mem_intv = (uint64_t)atol(need());
		else if (a == "-c") par.max_occ = atoi(need());
		else if (a == "-s") par.split_width = atoi(need());
		else if (a == "-t") { n_threads = atoi(need()); if (n_threads < 1) n_threads = 1; }
		else if (a == "-K") fixed_chunk = atol(need());
		else if (a == "--gpus") n_gpus = atoi(need());
		else if (a == "--devices") { for (const char *q = need(); *q;) { devices.push_back((int)strtol(q, (char **)&q, 10)); if (*q == ',') ++q; else if (*q) { usage(); return 1; } } }
		else if (a == "--no-sal") par.want_sal = 0;
		else if (a == "-v") verbose = atoi(need()) >= 4 ? 1 : 0; // bwa's verbosity levels: 4 = debugging output
		else if (a == "--dump-seeds") dump = need();
		else if (a.size() == 2 && a[0] == '-' && strchr("wdDWmABOELUxRHoThI", a[1])) (void)need(); // flags with a value, other stages
		else if (a.size() == 2 && a[0] == '-' && strchr("SPpj5qaCVYM1", a[1])) {}                    // switches, other stages
		else if (a[0] == '-' && a.size() > 1) { fprintf(stderr, "[E::main] unknown option %s\n", a.c_str()); usage(); return 1; }
		else pos.push_back(argv[i]);
	}
	if (pos.size() < 2) { usage(); return 1; }
	if (par.min_seed_len < 1) { fprintf(stderr, "[E::main] -k must be positive\n"); return 1; }

	cs_index_t *idx = nullptr;
	if (cs_index_load(pos[0], &idx)) { fprintf(stderr, "[E::main] fail to locate the index files: %s\n", cs_last_error()); return 1; }
	cs_index_view_t view; cs_index_view(idx, &view);
	int ndev = 0;
	if (cs_device_count(&ndev) || ndev < 1) { fprintf(stderr, "[E::main] no MI355X visible: %s\n", cs_last_error()); return 1; }
	if (!devices.empty()) n_gpus = (int)devices.size();
	else { if (n_gpus < 1 || n_gpus > ndev) n_gpus = ndev; for (int g = 0; g < n_gpus; ++g) devices.push_back(g); }
	for (int d : devices) if (d < 0 || d >= ndev) { fprintf(stderr, "[E::main] --devices: no device %d (%d visible)\n", d, ndev); return 1; }
	std::vector<cs_engine_t *> eng((size_t)n_gpus, nullptr);
	cs_engine_options_t eopt; cs_engine_options_default(&eopt);
	eopt.count_sal_merged = 1; // "SA Lookup: ... calls, % merged" as CompSeed prints it (main.cpp:209-210)
	eopt.verbose = verbose;
	for (int g = 0; g < n_gpus; ++g)
		if (cs_engine_create_opts(&view, devices[(size_t)g], &eopt, &eng[g])) { fprintf(stderr, "[E::main] GPU %d: %s\n", g, cs_last_error()); return 1; }

	FILE *fo = dump ? fopen(dump, "w") : nullptr;
	if (dump && !fo) { fprintf(stderr, "[E::main] cannot write %s\n", dump); return 1; }
	long chunk_bases = fixed_chunk > 0 ? fixed_chunk : 10000000L * n_threads; // main.cpp:437
	cs_reader_t *rd = nullptr;
	if (cs_reader_open(pos[1], chunk_bases, &rd)) { fprintf(stderr, "[E::main] %s\n", cs_last_error()); return 1; }

	// The reference's three-step pipeline (main.cpp:438) with the GPUs in the middle: chunk n+1 is read and submitted while chunk n
	// is being seeded; every GPU gets a contiguous read range of each chunk (neighbouring, overlapping reads stay together).
	auto read_chunk = [&](Chunk &c, uint64_t first_id) -> bool {
		if (cs_reader_next(rd, &c.bases, &c.off, &c.n)) { fprintf(stderr, "[E::process] %s\n", cs_last_error()); exit(1); }
		c.first_id = first_id;
		return c.n > 0;
	};
	auto submit = [&](Chunk &c) {
		c.loff.assign((size_t)n_gpus, std::vector<uint64_t>());
		for (int g = 0; g < n_gpus; ++g) {
			const int64_t r0 = c.n * g / n_gpus, r1 = c.n * (g + 1) / n_gpus;
			c.loff[g].resize((size_t)(r1 - r0) + 1);
			for (int64_t r = r0; r <= r1; ++r) c.loff[g][(size_t)(r - r0)] = c.off[r] - c.off[r0];
			if (cs_engine_submit(eng[g], &par, r1 - r0, c.bases + c.off[r0], c.loff[g].data())) { fprintf(stderr, "[E::main] GPU %d: %s\n", g, cs_last_error()); exit(1); }
		}
	};
	auto collect = [&](Chunk &c) {
		for (int g = 0; g < n_gpus; ++g) {
			cs_packed_result_t R;
			if (cs_engine_collect_packed(eng[g], &R)) { fprintf(stderr, "[E::main] GPU %d: %s\n", g, cs_last_error()); exit(1); }
			if (!fo) continue;
			const int64_t r0 = c.n * g / n_gpus;
			for (int64_t r = 0; r < R.n_reads; ++r) {
				const uint64_t id = c.first_id + (uint64_t)(r0 + r) + 1; // read names are running integers from 1 (main.cpp:47)
				uint64_t sd = R.seed_off ? R.seed_off[r] : 0;
				for (uint64_t m = R.mem_off[r]; m < R.mem_off[r + 1]; ++m) {
					cs_intv_t v; cs_unpack_mem(&R, m, &v);
					fprintf(fo, "M\t%lu\t%u\t%u\t%lu\t%lu\t%lu\n", (unsigned long)id, (unsigned)(v.info >> 32), (unsigned)v.info, (unsigned long)v.x0, (unsigned long)v.x1, (unsigned long)v.x2);
				}
				if (R.seed_off)
					for (uint64_t m = R.mem_off[r]; m < R.mem_off[r + 1]; ++m) {
						cs_intv_t v; cs_unpack_mem(&R, m, &v);
						const uint32_t cnt = cs_mem_seed_count(&v, R.max_occ);
						for (uint32_t k = 0; k < cnt; ++k)
							fprintf(fo, "S\t%lu\t%d\t%d\t%ld\n", (unsigned long)id, (int)(v.info >> 32), (int)((uint32_t)v.info - (uint32_t)(v.info >> 32)), (long)cs_packed_seed_rbeg(&R, sd + k));
						sd += cnt;
					}
			}
		}
	};
	Chunk ch[2];
	uint64_t n_processed = 0;
	int cur = 0;
	if (read_chunk(ch[0], 0)) {
		submit(ch[0]);
		for (;;) {
			Chunk &c = ch[cur], &nx = ch[cur ^ 1];
			const bool more = read_chunk(nx, n_processed + (uint64_t)c.n);
			if (more) submit(nx);
			collect(c);
			n_processed += (uint64_t)c.n;
			if (!more) break;
			cur ^= 1;
		}
	}
	cs_reader_close(rd);
	if (fo) fclose(fo);

	cs_stats_t tot; memset(&tot, 0, sizeof tot);
	for (int g = 0; g < n_gpus; ++g) {
		cs_stats_t st; cs_engine_stats(eng[g], &st);
		tot.reads += st.reads; tot.bwt_queries += st.bwt_queries; tot.bwt_calls += st.bwt_calls; tot.sal_queries += st.sal_queries;
		tot.sal_calls += st.sal_calls; tot.mems += st.mems; tot.seeds += st.seeds;
		tot.seed_kernel_ms = st.seed_kernel_ms > tot.seed_kernel_ms ? st.seed_kernel_ms : tot.seed_kernel_ms;
		tot.sal_kernel_ms = st.sal_kernel_ms > tot.sal_kernel_ms ? st.sal_kernel_ms : tot.sal_kernel_ms;
		tot.total_ms = st.total_ms > tot.total_ms ? st.total_ms : tot.total_ms;
		cs_engine_destroy(eng[g]);
	}
	cs_index_free(idx);
	// display_profile, main.cpp:203-214
	fprintf(stderr, "BWT-extend:  %lu queries, %lu calls, %.2f %% hit in SST\n", (unsigned long)tot.bwt_queries, (unsigned long)tot.bwt_calls,
	        tot.bwt_queries ? 100.0 * (double)(tot.bwt_queries - tot.bwt_calls) / (double)tot.bwt_queries : 0.0);
	fprintf(stderr, "SA Lookup:   %lu queries, %lu calls, %.2f %% merged\n", (unsigned long)tot.sal_queries, (unsigned long)tot.sal_calls,
	        tot.sal_queries ? 100.0 * (double)(tot.sal_queries - tot.sal_calls) / (double)tot.sal_queries : 0.0);
	fprintf(stderr, "Wall time:   BWT %.2f SAL %.2f seconds on %d GPU(s); %lu reads, %lu mems, %lu seeds\n", tot.seed_kernel_ms / 1e3,
	        tot.sal_kernel_ms / 1e3, n_gpus, (unsigned long)tot.reads, (unsigned long)tot.mems, (unsigned long)tot.seeds);
	return 0;
}
