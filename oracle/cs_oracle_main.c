/* oracle/cs_oracle_main.c -- TEST INFRASTRUCTURE ONLY: command-line driver for the CPU restatement.
 *   cs_oracle <idx prefix> <reads.txt> [-k INT] [-r FLOAT] [-y INT] [-c INT] [-s INT] [-t threads] [-m 0|1] [-B batch] [-o out.bin]
 * Writes the same CSGOLD1 layout as oracle/ref_harness.cpp so the two can be compared with cmp(1). */
#define _GNU_SOURCE
#include "cs_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: cs_oracle <idx prefix> <reads.txt> [-k -r -y -c -s -t -m -B -o]\n"); return 1; }
	cso_params_t par; cso_params_default(&par);
	int threads = 1, mode = 1, batch = 512; const char *out = NULL;
	for (int i = 3; i < argc; ++i) {
		if (!strcmp(argv[i], "-k")) par.min_seed_len = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-r")) par.split_factor = (float)atof(argv[++i]);
		else if (!strcmp(argv[i], "-y")) par.max_mem_intv = (uint64_t)atol(argv[++i]);
		else if (!strcmp(argv[i], "-c")) par.max_occ = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-s")) par.split_width = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-t")) threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-m")) mode = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-B")) batch = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-o")) out = argv[++i];
		else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
	}
	cso_index_t idx;
	int rc = cso_index_load(&idx, argv[1]);
	if (rc) { fprintf(stderr, "cannot load index %s (%d)\n", argv[1], rc); return 1; }
	FILE *fp = fopen(argv[2], "rb");
	if (!fp) { perror(argv[2]); return 1; }
	fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
	uint8_t *raw = (uint8_t *)malloc((size_t)sz + 1);
	if (fread(raw, 1, (size_t)sz, fp) != (size_t)sz) return 1;
	fclose(fp);
	size_t n = 0;
	for (long i = 0; i < sz; ++i) n += raw[i] == '\n';
	if (sz && raw[sz - 1] != '\n') ++n;
	uint64_t *off = (uint64_t *)malloc((n + 1) * 8);
	uint8_t *bases = (uint8_t *)malloc((size_t)sz + 1);
	size_t r = 0, w = 0; off[0] = 0;
	for (long i = 0; i < sz; ++i) { if (raw[i] == '\n') off[++r] = w; else bases[w++] = raw[i]; }
	if (r < n) off[++r] = w;
	free(raw);

	uint64_t *mo, *so; cso_intv_t *mems; cso_seed_t *seeds; cso_stats_t st;
	double t0 = now();
	cso_seed_batch(&idx, &par, (int64_t)n, bases, off, mode, batch, 1, threads, &mo, &mems, &so, &seeds, &st);
	double dt = now() - t0;
	fprintf(stderr, "[cs_oracle] reads=%zu mems=%lu seeds=%lu bwt_queries=%lu bwt_calls=%lu blocks=%lu sal_queries=%lu sal_calls=%lu sal_steps=%lu  %.3f s  %.0f reads/s (t=%d mode=%d)\n",
	        n, (unsigned long)st.n_mems, (unsigned long)st.n_seeds, (unsigned long)st.bwt_queries, (unsigned long)st.bwt_calls,
	        (unsigned long)st.bwt_blocks, (unsigned long)st.sal_queries, (unsigned long)st.sal_calls, (unsigned long)st.sal_steps,
	        dt, n / dt, threads, mode);
	if (out) {
		FILE *fo = fopen(out, "wb");
		uint64_t hdr[8] = {n, st.n_mems, st.n_seeds, st.bwt_queries, st.bwt_calls, st.sal_queries, st.sal_calls, 0};
		fwrite("CSGOLD1", 1, 8, fo); fwrite(hdr, 8, 8, fo);
		fwrite(mo, 8, n + 1, fo); fwrite(mems, sizeof(cso_intv_t), st.n_mems, fo);
		fwrite(so, 8, n + 1, fo); fwrite(seeds, sizeof(cso_seed_t), st.n_seeds, fo);
		fclose(fo);
	}
	cso_free(mo); cso_free(mems); cso_free(so); cso_free(seeds); free(off); free(bases);
	cso_index_free(&idx);
	return 0;
}
