// align.cpp -- the aligner object behind chaining (SURVEY 8f row 4 as a whole): cs_aligner_create / destroy / stats, the argument checks
// of cs_extend_chains (whose work -- mem_chain2aln_across_reads_V2, mapping/comp_seed.cpp:1319-2237 -- runs on the GPU: align_gpu.hip around
// extend.hip) and cs_dedup_regions (dedup.cpp).  Sequences are laid out so that NO per-pair copy is made: the query buffer holds every
// read once forward and once reversed, the target buffer every chain's reference window once forward and once reversed, and a pair is
// four numbers.  Results are the reference's regions field by field (tests/golden/aln1, flt1, ddp1; tests/test_gpu_align.py).
#include "cs_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct cs_aligner {
	cs_refseq_view ref;
	std::vector<uint8_t> pac;             // forward strand, four bases per byte, first base in the top bits (bntseq.c:236-237)
	cs_aln_params_t par{};
	cs_extender_t *ext = nullptr; int device = -1; cs_aligner_gpu *gpu = nullptr;   // gpu: the device side of cs_extend_chains (align_gpu.hip)
	std::vector<uint64_t> reg_off; std::vector<cs_alnreg_t> regs;
	std::vector<uint64_t> dd_off; std::vector<cs_alnreg_t> dd_regs; std::vector<int32_t> dd_ncomp;   // cs_dedup_regions' result
	cs_aln_stats_t st{};
};


extern "C" void cs_aln_params_default(cs_aln_params_t *p)
{
	if (!p) return;
	// mem_opt_init (comp_seed.cpp:26-58)
	p->a = 1; p->b = 4; p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1; p->pen_clip5 = p->pen_clip3 = 5; p->w = 100; p->zdrop = 100;
	p->threads = 16; p->flags = 0;
}

extern "C" int cs_aligner_create(const char *prefix, int device, const cs_aln_params_t *par, cs_aligner_t **out)
{
	if (!prefix || !out) return cs_fail_(CS_EINVAL, "cs_aligner_create: null argument");
	*out = nullptr;
	cs_aligner *A = new cs_aligner();
	if (par) A->par = *par; else cs_aln_params_default(&A->par);
	const cs_aln_params_t &o = A->par;
	if (o.a < 1 || o.b < 0 || o.e_del < 1 || o.e_ins < 1 || o.o_del < 0 || o.o_ins < 0 || o.w < 1 || o.zdrop < 0 || (o.flags & ~(CS_ALN_NO_LIGHT_PATHS | CS_ALN_PURGE_FROM_HBM))) { delete A; return cs_fail_(CS_EINVAL, "cs_aligner_create: bad scoring parameters or unknown flags"); }
	int rc = cs_load_contigs_(prefix, A->ref);
	if (rc != CS_OK) { delete A; return rc; }
	rc = cs_load_pac_(prefix, A->ref.l_pac, A->pac);
	if (rc != CS_OK) { delete A; return rc; }
	cs_ext_params_t xp;
	for (int i = 0, k = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) xp.mat[k++] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? o.a : -o.b); // bwa_fill_scmat (bwalib/bwa.c:17-29)
	xp.o_del = o.o_del; xp.e_del = o.e_del; xp.o_ins = o.o_ins; xp.e_ins = o.e_ins; xp.zdrop = o.zdrop; xp.end_bonus = o.pen_clip5; xp.flags = 0;
	// (the end bonus only enters the band limit, ksw.c:402-410; the reference builds one object per side, with pen_clip5 and pen_clip3,
	// comp_seed.cpp:1702-1708: two extenders are kept when the two differ)
	A->device = device;
	if (device >= 0) { // device -1: an aligner for the host-side passes only (cs_dedup_regions); cs_extend_chains then fails with CS_EDEVICE
		rc = cs_extender_create(device, &xp, &A->ext);
		if (rc != CS_OK) { delete A; return rc; }
	}
	*out = A;
	return CS_OK;
}
extern "C" void cs_aligner_destroy(cs_aligner_t *A)
{
	if (!A) return;
	if (A->gpu) cs_aligner_gpu_release_(A->gpu);
	if (A->ext) cs_extender_destroy(A->ext);
	delete A;
}

extern "C" int cs_extend_chains(cs_aligner_t *A, const cs_chain_result_t *chains, const int32_t *cseed_score, const uint8_t *bases,
                                const uint64_t *read_offsets, cs_aln_result_t *out)
{
	if (A && !A->ext) return cs_fail_(CS_EDEVICE, "cs_extend_chains: this aligner was created without a device (device -1): the extension runs on the GPU only");
	if (!A || !chains || !out || (chains->n_reads > 0 && (!read_offsets || !chains->chain_off)) || (chains->n_chains > 0 && (!chains->chains || !chains->cseed_off)) ||
	    (chains->n_seeds > 0 && (!chains->cseeds || !bases)))
		return cs_fail_(CS_EINVAL, "cs_extend_chains: bad argument");
	if (A->par.pen_clip5 != A->par.pen_clip3) return cs_fail_(CS_EINVAL, "cs_extend_chains: pen_clip5 != pen_clip3 is not supported yet (one extender, one end bonus)");
	for (int64_t r = 0; r < chains->n_reads; ++r) if (chains->chain_off[r + 1] < chains->chain_off[r] || chains->chain_off[r + 1] > chains->n_chains) return cs_fail_(CS_EINVAL, "cs_extend_chains: chain_off is not a CSR offset array");
	for (uint64_t c = 0; c < chains->n_chains; ++c) if (chains->cseed_off[c + 1] < chains->cseed_off[c] || chains->cseed_off[c + 1] > chains->n_seeds || (int64_t)(chains->cseed_off[c + 1] - chains->cseed_off[c]) != (int64_t)chains->chains[c].n_seeds)
		return cs_fail_(CS_EINVAL, "cs_extend_chains: cseed_off does not match the chains' seed counts");
	// everything per chain, seed and read runs on the GPU (align_gpu.hip)
	const int rc = cs_extend_chains_gpu_(&A->gpu, A->device, A->ext, A->ref, A->pac, A->par, chains, cseed_score, bases, read_offsets, A->reg_off, A->regs, A->st);
	if (rc != CS_OK) return rc;
	out->n_reads = chains->n_reads; out->n_regs = A->regs.size(); out->reg_off = A->reg_off.data(); out->regs = A->regs.data();
	return CS_OK;
}

extern "C" int cs_dedup_regions(cs_aligner_t *A, const cs_dedup_params_t *par, const cs_aln_result_t *regs, const uint8_t *bases, const uint64_t *read_offsets,
                                cs_aln_result_t *out, const int32_t **n_comp)
{
	if (!A || !par || !regs || !out || (regs->n_reads > 0 && (!regs->reg_off || !read_offsets)) || (regs->n_regs > 0 && (!regs->regs || !bases))) return cs_fail_(CS_EINVAL, "cs_dedup_regions: bad argument");
	if (regs->regs == A->dd_regs.data() && regs->n_regs) return cs_fail_(CS_EINVAL, "cs_dedup_regions: the input is this function's own previous output");
	if (par->max_chain_gap < 0 || !(par->mask_level_redun > 0.f)) return cs_fail_(CS_EINVAL, "cs_dedup_regions: bad parameters");
	const cs_aligner_core core = {&A->ref, &A->pac, &A->par};
	const int rc = cs_dedup_regions_(core, par, regs, bases, read_offsets, A->dd_off, A->dd_regs, A->dd_ncomp);
	if (rc != CS_OK) return rc;
	out->n_reads = regs->n_reads; out->n_regs = A->dd_regs.size(); out->reg_off = A->dd_off.data(); out->regs = A->dd_regs.data();
	if (n_comp) *n_comp = A->dd_ncomp.data();
	return CS_OK;
}

extern "C" int cs_aligner_stats(const cs_aligner_t *A, cs_aln_stats_t *st)
{
	if (!A || !st) return cs_fail_(CS_EINVAL, "null argument");
	*st = A->st;
	if (A->ext) { cs_ext_stats_t x; if (cs_extender_stats(A->ext, &x) == CS_OK) { st->ext_cells = x.cells; st->ext_kernel_ms = x.kernel_ms; } } // (the extender is this aligner's own)
	return CS_OK;
}
