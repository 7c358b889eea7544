// oracle/ref_ksw_kat.cpp -- TEST INFRASTRUCTURE (build container only).
//
// Known-answer vectors for the scalar definition of the seed extension: calls the REFERENCE's own ksw_extend2 (bwalib/ksw.c:380, linked
// from the object compiled in place) on generated pairs and writes inputs + outputs in the record format of ref_bsw_trace.cpp with
// kind = 0.  The pairs stress what the golden read sets (substitutions only, default parameters) do not: insertions and deletions of
// 1..12 bases, unrelated tails, ambiguous bases, narrow bands (w 1..40), small and disabled Z-drop, asymmetric gap penalties, h0 from 1
// to a few hundred, empty targets, queries of 1..400 bases.
//   ref_ksw_kat <out.bin> <n pairs> <seed> <match a> <mismatch b>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
#include "bwalib/ksw.h"
}

static uint64_t s_;
static uint32_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (uint32_t)(s_ >> 11); }
static int rin(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: ref_ksw_kat <out.bin> <n> <seed> <a> <b>\n"); return 1; }
	const long n = atol(argv[2]); s_ = strtoull(argv[3], 0, 10) * 0x9E3779B97F4A7C15ull + 1;
	const int a = atoi(argv[4]), b = atoi(argv[5]);
	int8_t mat[25];
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? a : -b); // what bwa_fill_scmat builds (bwalib/bwa.c:17-29)
	FILE *fp = fopen(argv[1], "wb");
	if (!fp) { perror(argv[1]); return 1; }
	fwrite("CSBSW01", 1, 8, fp); fwrite(mat, 1, 25, fp);
	// a menu of parameter sets (so that a consumer can group the records into a few dozen objects): set 0 = mem_opt_init's scoring
	struct Set { int o_del, e_del, o_ins, e_ins, zdrop, end_bonus; };
	const int N_SETS = 20;
	Set menu[N_SETS];
	menu[0] = Set{6, 1, 6, 1, 100, 5};
	for (int k = 1; k < N_SETS; ++k)
		menu[k] = Set{rin(1, 2) == 1 ? rin(1, 10) : 6, rin(1, 3) == 1 ? rin(2, 3) : 1, rin(1, 2) == 1 ? rin(1, 10) : 6, rin(1, 3) == 1 ? rin(2, 3) : 1,
		              k % 4 == 0 ? 0 : rin(1, 2) == 1 ? rin(1, 30) : 100, rin(0, 1) ? 5 : rin(0, 20)};
	for (long it = 0; it < n; ++it) {
		const int qlen = (it % 50 == 0) ? rin(150, 400) : rin(1, 140);
		std::vector<uint8_t> q((size_t)qlen), t;
		for (auto &c : q) c = (uint8_t)rin(0, 3);
		// target = the query with edits, followed by an unrelated tail (what a reference window looks like)
		const int style = rin(0, 9);
		const int p_sub = style < 3 ? 0 : style < 7 ? 30 : 8, p_gap = style < 2 ? 0 : style < 8 ? 60 : 15; // one in p_sub / p_gap positions
		for (int j = 0; j < qlen; ++j) {
			if (p_gap && rin(1, p_gap) == 1) {
				const int gl = rin(1, rin(1, 4) == 1 ? 12 : 3);
				if (rin(0, 1)) { for (int g = 0; g < gl; ++g) t.push_back((uint8_t)rin(0, 3)); } // extra target bases (deletion from the query's view)
				else { j += gl - 1; continue; }                                                 // query bases without a partner (insertion)
			}
			if (j < qlen) t.push_back(p_sub && rin(1, p_sub) == 1 ? (uint8_t)rin(0, 3) : q[(size_t)j]);
		}
		if (style == 9) { const size_t cut = t.size() / 2; t.resize(cut); }                          // the similarity stops half way
		for (int g = rin(0, 60); g > 0; --g) t.push_back((uint8_t)rin(0, 3));
		if (rin(1, 12) == 1 && !t.empty()) t[(size_t)rin(0, (int)t.size() - 1)] = 4;
		if (rin(1, 12) == 1) q[(size_t)rin(0, qlen - 1)] = 4;
		if (it % 97 == 0) t.clear();
		const int tlen = (int)t.size();
		const Set &ps = menu[rin(1, 3) == 1 ? 0 : rin(1, N_SETS - 1)];
		const int o_del = ps.o_del, e_del = ps.e_del, o_ins = ps.o_ins, e_ins = ps.e_ins, zdrop = ps.zdrop, end_bonus = ps.end_bonus;
		static const int bands[8] = {1, 3, 8, 21, 40, 100, 100, 200};
		const int w = bands[rin(0, 7)];
		const int h0 = rin(1, 4) == 1 ? rin(1, 10) : rin(19, 150) * a;
		int qle = 0, tle = 0, gtle = 0, gscore = 0, max_off = 0;
		const int score = ksw_extend2(qlen, q.data(), tlen, t.data(), 5, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, &qle, &tle, &gtle, &gscore, &max_off);
		const int32_t rec[17] = {0, w, zdrop, end_bonus, o_del, e_del, o_ins, e_ins, qlen, tlen, h0, score, qle, tle, gtle, gscore, max_off};
		fwrite(rec, 4, 17, fp); fwrite(q.data(), 1, (size_t)qlen, fp); fwrite(t.data(), 1, (size_t)tlen, fp);
	}
	fclose(fp);
	return 0;
}
