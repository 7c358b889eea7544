/* integration/mock_engine.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for libcompseed_amd.so that implements, on the CPU and on top of the oracle (oracle/cs_oracle.c), exactly the entry
 * points integration/compseed_gpu.patch calls -- so that the PATCHED reference can be run end to end in the build container, which
 * has no GPU, and its SAM output compared with the unpatched reference's: that proves the reference-side binding (chunk packing,
 * submit in the reader step / collect in the worker step, cs_unpack_mem, seeds in mem-then-slot order) without the device.
 * The real library's results are compared with the same oracle on the GPU box (tests/test_gpu_*.py).  Never shipped, never linked
 * into the product. */
#include "compseed_amd.h"
#include "../oracle/cs_oracle.h"

#include <stdlib.h>
#include <string.h>

struct cs_engine {
	cso_index_t ix;
	struct { cs_params_t par; int64_t n; const uint8_t *bases; const uint64_t *off; } q[4];
	uint64_t n_sub, n_col;
	struct { uint64_t *mem_off, *seed_off; cs_mem16_t *mems; uint32_t *rlo; uint8_t *rhi; } res[4];
	cs_stats_t st;
};
static const char *g_err = "";
const char *cs_last_error(void) { return g_err; }
void cs_params_default(cs_params_t *p)
{
	memset(p, 0, sizeof *p);
	p->min_seed_len = 19; p->split_factor = 1.5f; p->split_width = 10; p->max_occ = 500; p->max_mem_intv = 20; p->want_sal = 1; p->sst_mode = 1;
}
int cs_host_alloc(size_t bytes, void **ptr) { *ptr = malloc(bytes ? bytes : 1); return *ptr ? CS_OK : CS_ENOMEM; }
int cs_host_free(void *ptr) { free(ptr); return CS_OK; }
int cs_engine_create(const cs_index_view_t *v, int device, cs_engine_t **out)
{
	(void)device;
	cs_engine_t *e = (cs_engine_t *)calloc(1, sizeof *e);
	if (!e) return CS_ENOMEM;
	cso_index_wrap(&e->ix, v->primary, &v->L2[1], v->bwt, v->bwt_size, v->sa, v->n_sa, v->sa_intv);
	*out = e;
	return CS_OK;
}
void cs_engine_destroy(cs_engine_t *e)
{
	if (!e) return;
	for (int k = 0; k < 3; ++k) { free(e->res[k].mem_off); free(e->res[k].seed_off); free(e->res[k].mems); free(e->res[k].rlo); free(e->res[k].rhi); }
	free(e);
}
int cs_engine_stats(const cs_engine_t *e, cs_stats_t *st) { *st = e->st; return CS_OK; }
int cs_engine_submit(cs_engine_t *e, const cs_params_t *par, int64_t n_reads, const uint8_t *bases, const uint64_t *offsets)
{
	if (e->n_sub - e->n_col >= 4) { g_err = "four batches are in flight already"; return CS_EINVAL; }
	int k = (int)(e->n_sub++ % 4);
	e->q[k].par = *par; e->q[k].n = n_reads; e->q[k].bases = bases; e->q[k].off = offsets;
	return CS_OK;
}
int cs_engine_collect_packed(cs_engine_t *e, cs_packed_result_t *out)
{
	if (e->n_sub == e->n_col) { g_err = "nothing has been submitted"; return CS_EINVAL; }
	int k = (int)(e->n_col++ % 4);
	cso_params_t op = {e->q[k].par.min_seed_len, e->q[k].par.split_factor, e->q[k].par.split_width, e->q[k].par.max_occ, e->q[k].par.max_mem_intv};
	uint64_t *mo = NULL, *so = NULL; cso_intv_t *mm = NULL; cso_seed_t *ss = NULL; cso_stats_t st;
	if (cso_seed_batch(&e->ix, &op, e->q[k].n, e->q[k].bases, e->q[k].off, 1, 512, 1, 4, &mo, &mm, &so, &ss, &st)) { g_err = "oracle failed"; return CS_EDEVICE; }
	free(e->res[k].mem_off); free(e->res[k].seed_off); free(e->res[k].mems); free(e->res[k].rlo); free(e->res[k].rhi);
	e->res[k].mem_off = mo; e->res[k].seed_off = so;
	e->res[k].mems = (cs_mem16_t *)malloc((st.n_mems + 1) * sizeof(cs_mem16_t));
	e->res[k].rlo = (uint32_t *)malloc((st.n_seeds + 1) * sizeof(uint32_t)); e->res[k].rhi = (uint8_t *)malloc(st.n_seeds + 1);
	for (uint64_t i = 0; i < st.n_mems; ++i) { /* the packing of include/compseed_amd.h (CS_MEM_PACKED16) */
		uint64_t beg = mm[i].info >> 32, end = mm[i].info & 0xffffffffull;
		e->res[k].mems[i].w0 = mm[i].x0 | (mm[i].x2 & 0x7fffffffull) << 33;
		e->res[k].mems[i].w1 = mm[i].x1 | beg << 33 | end << 48 | (mm[i].x2 >> 31) << 63;
	}
	for (uint64_t i = 0; i < st.n_seeds; ++i) { e->res[k].rlo[i] = (uint32_t)(uint64_t)ss[i].rbeg; e->res[k].rhi[i] = (uint8_t)((uint64_t)ss[i].rbeg >> 32); } /* CS_SEED_RBEG40 */
	cso_free(mm); cso_free(ss);
	memset(out, 0, sizeof *out);
	out->n_reads = e->q[k].n; out->n_mems = st.n_mems; out->n_seeds = st.n_seeds; out->mem_format = CS_MEM_PACKED16; out->max_occ = e->q[k].par.max_occ;
	out->mem_off = e->res[k].mem_off; out->mems = e->res[k].mems; out->seed_off = e->res[k].seed_off; out->seed_format = CS_SEED_RBEG40; out->seed_rbeg_lo = e->res[k].rlo; out->seed_rbeg_hi = e->res[k].rhi;
	e->st.reads += (uint64_t)e->q[k].n; e->st.mems += st.n_mems; e->st.seeds += st.n_seeds; e->st.bwt_queries += st.bwt_queries; e->st.bwt_calls += st.bwt_calls;
	return CS_OK;
}

/* the extension stage has no CPU stand-in here: the patched reference then keeps its own mem_chain2aln_across_reads_V2 */
void cs_aln_params_default(cs_aln_params_t *p) { if (p) memset(p, 0, sizeof *p); }
int cs_aligner_create(const char *prefix, int device, const cs_aln_params_t *par, cs_aligner_t **out)
{
	(void)prefix; (void)device; (void)par;
	if (out) *out = NULL;
	g_err = "the mock has no extension stage";
	return CS_EDEVICE;
}
void cs_aligner_destroy(cs_aligner_t *a) { (void)a; }
int cs_extend_chains(cs_aligner_t *a, const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases, const uint64_t *read_offsets, cs_aln_result_t *out)
{
	(void)a; (void)chains; (void)cseed_score; (void)bases; (void)read_offsets; (void)out;
	g_err = "the mock has no extension stage";
	return CS_EDEVICE;
}
/* (never reached with the mock: cs_aligner_create fails, so the patched reference keeps its own chaining, filters and de-duplication) */
int cs_chainer_create(const char *prefix, cs_chainer_t **out) { (void)prefix; if (out) *out = NULL; g_err = "the mock has no chainer"; return CS_EDEVICE; }
void cs_chainer_destroy(cs_chainer_t *c) { (void)c; }
void cs_chain_params_default(cs_chain_params_t *p) { if (p) memset(p, 0, sizeof *p); }
void cs_flt_params_default(cs_flt_params_t *p) { if (p) memset(p, 0, sizeof *p); }
void cs_dedup_params_default(cs_dedup_params_t *p) { if (p) memset(p, 0, sizeof *p); }
int cs_chain_batch(cs_chainer_t *c, const cs_chain_params_t *par, const cs_result_t *seeds, const uint64_t *read_offsets, int n_threads, cs_chain_result_t *out)
{ (void)c; (void)par; (void)seeds; (void)read_offsets; (void)n_threads; (void)out; return CS_EDEVICE; }
int cs_chain_filter(cs_chainer_t *c, const cs_flt_params_t *par, const cs_chain_result_t *in, const uint8_t *bases, const uint64_t *read_offsets, int n_threads, cs_chain_result_t *out, const int32_t **cseed_score)
{ (void)c; (void)par; (void)in; (void)bases; (void)read_offsets; (void)n_threads; (void)out; (void)cseed_score; return CS_EDEVICE; }
int cs_dedup_regions(cs_aligner_t *a, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets, cs_aln_result_t *out, const int32_t **n_comp)
{ (void)a; (void)par; (void)regs; (void)bases; (void)read_offsets; (void)out; (void)n_comp; return CS_EDEVICE; }
