// oracle/ref_bsw_trace.cpp -- TEST INFRASTRUCTURE (build container only; never shipped in the product, never sent to the GPU box as source).
//
// Records every banded Smith-Waterman extension the REFERENCE performs.  `make -C oracle ref` links the reference's own main.cpp and
// objects into oracle/_ref/CompSeed.bswtrace with
//     -Wl,--wrap=<BandedPairWiseSW::getScores8> --wrap=<...getScores16> --wrap=<...scalarBandedSWAWrapper> --wrap=<constructor>
// so that the calls mem_chain2aln_across_reads_V2 makes (mapping/comp_seed.cpp:1319; call sites :1719,1790,1859,1942,2003,2074) pass
// through the functions below: they call the reference's real implementation (mapping/bandedSWA.cpp:242, 412..., untouched) and append,
// per pair, its inputs and the six outputs the real code produced to the file named by $CS_BSW_TRACE.  Nothing is computed here.
//
// record = 17 x int32 { kind (8 / 16 / 1 = scalar), w, zdrop, end_bonus, o_del, e_del, o_ins, e_ins, qlen, tlen, h0,
//                       score, qle, tle, gtle, gscore, max_off } + qlen query bytes + tlen target bytes (codes 0..4)
// file   = "CSBSW01\0" + 25 bytes of the scoring matrix of the first object constructed + records
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "mapping/bandedSWA.h"

namespace {
struct Par { int o_del, e_del, o_ins, e_ins, zdrop, end_bonus; };
std::mutex mu;
std::map<const void *, Par> objs;
FILE *fp = nullptr;
bool header_done = false;

FILE *out(const int8_t *mat)
{
	if (!fp) {
		const char *fn = getenv("CS_BSW_TRACE");
		if (!fn) return nullptr;
		fp = fopen(fn, "wb");
		if (!fp) { perror(fn); exit(1); }
	}
	if (!header_done && mat) { fwrite("CSBSW01", 1, 8, fp); fwrite(mat, 1, 25, fp); header_done = true; }
	return fp;
}
void record(const void *self, int kind, const SeqPair *p, int n, const uint8_t *ref, const uint8_t *qer, int w)
{
	std::lock_guard<std::mutex> lk(mu);
	FILE *f = out(nullptr);
	if (!f) return;
	const Par q = objs[self];
	for (int i = 0; i < n; ++i) {
		const SeqPair &s = p[i];
		const int32_t rec[17] = {kind, w, q.zdrop, q.end_bonus, q.o_del, q.e_del, q.o_ins, q.e_ins, s.len2, s.len1, s.h0,
		                         s.score, s.qle, s.tle, s.gtle, s.gscore, s.max_off};
		fwrite(rec, 4, 17, f);
		fwrite(qer + s.idq, 1, (size_t)s.len2, f);
		fwrite(ref + s.idr, 1, (size_t)s.len1, f);
	}
	fflush(f);
}
// $CS_BSW_TIME set: the time the reference's real functions take is summed over all calls and threads and printed at exit --
// "[bswtrace] pairs=<n> thread_seconds=<s> cells=<qlen x tlen summed>" -- the CPU baseline of tools/extend_bench.py (the reference's own
// vectorised extension code on the GPU box's host cores).
std::atomic<long long> t_ns{0}, t_pairs{0}, t_cells{0};
bool timing() { static const bool on = getenv("CS_BSW_TIME") != nullptr; return on; }
void report()
{
	fprintf(stderr, "[bswtrace] pairs=%lld thread_seconds=%.6f cells=%lld\n", t_pairs.load(), (double)t_ns.load() * 1e-9, t_cells.load());
}
void arm_report() { static std::once_flag once; std::call_once(once, [] { atexit(report); }); }
template <class F> void timed(const SeqPair *p, int n, F &&call)
{
	if (!timing()) { call(); return; }
	arm_report();
	const auto t0 = std::chrono::steady_clock::now();
	call();
	t_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
	long long cells = 0;
	for (int i = 0; i < n; ++i) cells += (long long)p[i].len1 * p[i].len2;
	t_pairs += n; t_cells += cells;
}
} // namespace

// the real functions (the reference's code) and the wrappers the linker routes the reference's calls to
#define M_GS8  "_ZN16BandedPairWiseSW10getScores8EP10dnaSeqPairPhS2_iti"
#define M_GS16 "_ZN16BandedPairWiseSW11getScores16EP10dnaSeqPairPhS2_iti"
#define M_SCAL "_ZN16BandedPairWiseSW22scalarBandedSWAWrapperEP10dnaSeqPairPhS2_iii"
#define M_CTOR "_ZN16BandedPairWiseSWC1EiiiiiiPKaaai"
extern "C" {
void real_gs8(void *, SeqPair *, uint8_t *, uint8_t *, int32_t, uint16_t, int32_t) asm("__real_" M_GS8);
void real_gs16(void *, SeqPair *, uint8_t *, uint8_t *, int32_t, uint16_t, int32_t) asm("__real_" M_GS16);
void real_scal(void *, SeqPair *, uint8_t *, uint8_t *, int, int, int32_t) asm("__real_" M_SCAL);
void real_ctor(void *, int, int, int, int, int, int, const int8_t *, int8_t, int8_t, int) asm("__real_" M_CTOR);

void wrap_gs8(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int32_t n, uint16_t nt, int32_t w) asm("__wrap_" M_GS8);
void wrap_gs16(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int32_t n, uint16_t nt, int32_t w) asm("__wrap_" M_GS16);
void wrap_scal(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int n, int nt, int32_t w) asm("__wrap_" M_SCAL);
void wrap_ctor(void *self, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus, const int8_t *mat, int8_t a, int8_t b, int nt) asm("__wrap_" M_CTOR);

void wrap_gs8(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int32_t n, uint16_t nt, int32_t w)
{
	timed(p, n, [&] { real_gs8(self, p, ref, qer, n, nt, w); });
	record(self, 8, p, n, ref, qer, w);
}
void wrap_gs16(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int32_t n, uint16_t nt, int32_t w)
{
	timed(p, n, [&] { real_gs16(self, p, ref, qer, n, nt, w); });
	record(self, 16, p, n, ref, qer, w);
}
void wrap_scal(void *self, SeqPair *p, uint8_t *ref, uint8_t *qer, int n, int nt, int32_t w)
{
	timed(p, n, [&] { real_scal(self, p, ref, qer, n, nt, w); });
	record(self, 1, p, n, ref, qer, w);
}
void wrap_ctor(void *self, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus, const int8_t *mat, int8_t a, int8_t b, int nt)
{
	real_ctor(self, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, mat, a, b, nt);
	std::lock_guard<std::mutex> lk(mu);
	const Par q = {o_del, e_del, o_ins, e_ins, zdrop, end_bonus};
	objs[self] = q;
	out(mat);
}
}
