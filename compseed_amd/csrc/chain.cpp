// chain.cpp -- seed -> chain hand-off: the first consumer of the engine's output, mem_chain of the reference
// (mapping/comp_seed.cpp:241-285 with test_and_merge :182-203, bns_intv2rid FM_index/bntseq.c:370-378, frac_rep :271-280), as host
// code over the CSR result of a whole batch.  It proves the seeds are usable by the next stage: the chains are the reference's, chain
// by chain and seed by seed (tests/test_chain.py: golden chains dumped from the reference's own mem_chain on 14 runs).
//
// What has to be reproduced besides the merge rule is the ORDER and the CHOICE of chains.  The reference keeps a read's chains in a
// B-tree keyed by the chain's first reference position (cstl/kbtree.h, at most 9 keys per node for this key type) and asks it for the
// chain with the largest key <= the seed's position.  Keys are not unique (tandem repeats give chains with equal positions), and which
// of several equal keys the search returns, and where a new equal key lands, depends on how the tree has split so far; the final
// traversal is in tree order.  So the tree is restated here with the same search and insertion rules (lower bound inside a node, the
// first equal key on the way down wins, a new key goes right behind the position the search ends at, full children are split before
// descending, the median moves up) over (position, chain id) pairs.
#include "cs_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>


namespace {
constexpr int BT = 5, BMAX = 2 * BT - 1; // kb_init(chn, 512) with a 40-byte key: t = ((512 - 4 - 8) / (8 + 40) + 1) >> 1 = 5 (cstl/kbtree.h:57-62)

struct Node { int n; bool internal; int64_t pos[BMAX]; int32_t id[BMAX]; int32_t kid[BMAX + 1]; };

struct Tree { // node 0 is the root until it splits; nodes live in a vector that is reused from read to read
	std::vector<Node> nd; int root = 0; int size = 0;
	void clear() { nd.clear(); Node r; memset(&r, 0, sizeof r); nd.push_back(r); root = 0; size = 0; }
	// index of the last key <= k inside node x, the FIRST one among equal keys (r = 0 then); -1 if all keys are larger
	static int locate(const Node &x, int64_t k, int &r)
	{
		int begin = 0, end = x.n;
		if (x.n == 0) { r = 1; return -1; }
		while (begin < end) { int mid = (begin + end) >> 1; if (x.pos[mid] < k) begin = mid + 1; else end = mid; }
		if (begin == x.n) { r = 1; return x.n - 1; }
		r = k < x.pos[begin] ? -1 : 0;
		return r < 0 ? begin - 1 : begin;
	}
	int lower(int64_t k) const // chain id of the closest key <= k as kb_intervalp reports it, or -1
	{
		int lo = -1, x = root;
		for (;;) {
			int r, i = locate(nd[x], k, r);
			if (i >= 0 && r == 0) return nd[x].id[i];
			if (i >= 0) lo = nd[x].id[i];
			if (!nd[x].internal) return lo;
			x = nd[x].kid[i + 1];
		}
	}
	void split(int x, int i, int y) // y = full child i of x: its upper half becomes a new node, its median moves up into x
	{
		Node z; memset(&z, 0, sizeof z);
		z.internal = nd[y].internal; z.n = BT - 1;
		memcpy(z.pos, nd[y].pos + BT, sizeof(int64_t) * (BT - 1)); memcpy(z.id, nd[y].id + BT, sizeof(int32_t) * (BT - 1));
		if (nd[y].internal) memcpy(z.kid, nd[y].kid + BT, sizeof(int32_t) * BT);
		nd[y].n = BT - 1;
		const int zi = (int)nd.size();
		nd.push_back(z);
		Node &X = nd[x];
		memmove(X.kid + i + 2, X.kid + i + 1, sizeof(int32_t) * (X.n - i));
		X.kid[i + 1] = zi;
		memmove(X.pos + i + 1, X.pos + i, sizeof(int64_t) * (X.n - i)); memmove(X.id + i + 1, X.id + i, sizeof(int32_t) * (X.n - i));
		X.pos[i] = nd[y].pos[BT - 1]; X.id[i] = nd[y].id[BT - 1];
		++X.n;
	}
	void put(int64_t k, int32_t id)
	{
		++size;
		if (nd[root].n == BMAX) {
			Node s; memset(&s, 0, sizeof s); s.internal = true; s.kid[0] = root;
			const int si = (int)nd.size(); nd.push_back(s);
			split(si, 0, root);
			root = si;
		}
		int x = root;
		for (;;) {
			int r, i = locate(nd[x], k, r);
			if (!nd[x].internal) {
				Node &X = nd[x];
				if (i != X.n - 1) { memmove(&X.pos[i + 2], &X.pos[i + 1], sizeof(int64_t) * (X.n - i - 1)); memmove(&X.id[i + 2], &X.id[i + 1], sizeof(int32_t) * (X.n - i - 1)); } // (i may be -1)
				X.pos[i + 1] = k; X.id[i + 1] = id; ++X.n;
				return;
			}
			++i;
			if (nd[nd[x].kid[i]].n == BMAX) {
				split(x, i, nd[x].kid[i]);
				if (k > nd[x].pos[i]) ++i;
			}
			x = nd[x].kid[i];
		}
	}
	template <class F> void traverse(int x, F &f) const
	{
		const Node &X = nd[x];
		for (int i = 0; i <= X.n; ++i) {
			if (X.internal) traverse(X.kid[i], f);
			if (i < X.n) f(X.id[i]);
		}
	}
};

// a chain while it grows: the merge rule only ever looks at its first and its last seed; the seeds themselves are a linked list through the
// read's seed indices (next_of[]), so a read's chains cost no allocation of their own
struct Chain { int64_t pos; int32_t rid, n; cs_seed_t head, tail; uint32_t first, last; };

// contig of a forward-strand position (bns_pos2rid, bntseq.c:346-362): the last contig that starts at or before it
inline int contig_of(const cs_refseq_view &R, int64_t fwd_pos)
{
	if (fwd_pos >= R.l_pac) return -1;
	return (int)(std::upper_bound(R.offset.begin(), R.offset.end(), fwd_pos) - R.offset.begin()) - 1;
}
// contig of a seed [rb, re) in the doubled coordinate system (bns_intv2rid, bntseq.c:370-378): positions on the reverse strand are
// mirrored back first (bns_depos, bntseq.h:87); -2 = bridges the forward-reverse boundary, -1 = bridges two contigs
inline int contig_of_seed(const cs_refseq_view &R, int64_t rb, int64_t re)
{
	auto mirror = [&](int64_t p) { return p < R.l_pac ? p : 2 * R.l_pac - 1 - p; };
	if (rb < R.l_pac && R.l_pac < re) return -2;
	const int first = contig_of(R, mirror(rb));
	if (re <= rb) return first;
	return contig_of(R, mirror(re - 1)) == first ? first : -1;
}
// Does seed s belong to chain c (test_and_merge, comp_seed.cpp:182-203)?  Yes without a change when it lies inside the chain's
// query and reference span; yes, appended, when it follows the chain's last seed on the same strand within the band `w` and
// closer than max_chain_gap on both axes; otherwise a new chain starts.
inline bool absorb(const cs_chain_params_t &opt, int64_t l_pac, Chain &c, const cs_seed_t &s, int seed_rid, uint32_t s_idx, std::vector<uint32_t> &next_of)
{
	if (seed_rid != c.rid) return false;
	const cs_seed_t &head = c.head, &tail = c.tail;
	const bool in_query = s.qbeg >= head.qbeg && s.qbeg + s.len <= tail.qbeg + tail.len;
	const bool in_ref = s.rbeg >= head.rbeg && s.rbeg + s.len <= tail.rbeg + tail.len;
	if (in_query && in_ref) return true;
	const bool chain_fwd = tail.rbeg < l_pac || head.rbeg < l_pac;
	if (chain_fwd && s.rbeg >= l_pac) return false;                       // never across the strands
	const int64_t dq = (int64_t)s.qbeg - tail.qbeg, dr = s.rbeg - tail.rbeg;
	const bool near_diag = dq - dr <= opt.w && dr - dq <= opt.w;
	const bool close = dq - tail.len < opt.max_chain_gap && dr - tail.len < opt.max_chain_gap;
	if (dr < 0 || !near_diag || !close) return false;
	next_of[c.last] = s_idx; c.last = s_idx; c.tail = s; ++c.n;
	return true;
}

struct ReadOut { std::vector<cs_chain_t> chains; std::vector<cs_seed_t> seeds; std::vector<uint32_t> per_read; };

void chain_range(const cs_refseq_view &R, const cs_chain_params_t &opt, const cs_result_t &S, const uint64_t *read_off, int64_t r0, int64_t r1, ReadOut &out)
{
	Tree tree; std::vector<Chain> pool; std::vector<uint32_t> next_of;
	for (int64_t r = r0; r < r1; ++r) {
		const int len = (int)(read_off[r + 1] - read_off[r]);
		uint32_t n_out = 0;
		if (len >= opt.min_seed_len) {
			tree.clear(); pool.clear();
			const uint64_t s_base = S.seed_off[r];
			next_of.assign((size_t)(S.seed_off[r + 1] - s_base), 0xffffffffu);
			for (uint64_t si = S.seed_off[r]; si < S.seed_off[r + 1]; ++si) {
				const cs_seed_t &s = S.seeds[si];
				const int rid = contig_of_seed(R, s.rbeg, s.rbeg + s.len);
				if (rid < 0) continue; // bridging two sequences or the forward-reverse boundary (comp_seed.cpp:251)
				bool add = true;
				const uint32_t li = (uint32_t)(si - s_base);
				if (tree.size) { const int lo = tree.lower(s.rbeg); if (lo >= 0 && absorb(opt, R.l_pac, pool[lo], s, rid, li, next_of)) add = false; }
				if (add) { Chain c; c.pos = s.rbeg; c.rid = rid; c.n = 1; c.head = c.tail = s; c.first = c.last = li; pool.push_back(c); tree.put(s.rbeg, (int32_t)pool.size() - 1); }
			}
			// fraction of the read covered by repetitive mems (comp_seed.cpp:271-280); the mems are sorted by interval
			int beg = 0, end = 0, l_rep = 0;
			for (uint64_t m = S.mem_off[r]; m < S.mem_off[r + 1]; ++m) {
				if (S.mems[m].x2 <= (uint64_t)opt.max_occ) continue;
				const int b = (int)(S.mems[m].info >> 32), e = (int)(uint32_t)S.mems[m].info;
				if (b > end) { l_rep += end - beg; beg = b; end = e; } else end = std::max(end, e);
			}
			l_rep += end - beg;
			const float frac = (float)l_rep / len;
			auto emit = [&](int32_t id) {
				const Chain &c = pool[id];
				cs_chain_t o; o.pos = c.pos; o.rid = c.rid; o.n_seeds = c.n; o.frac_rep = frac; o.is_alt = R.is_alt[(size_t)c.rid]; // comp_seed.cpp:259
				out.chains.push_back(o);
				for (uint32_t k = c.first; k != 0xffffffffu; k = next_of[k]) out.seeds.push_back(S.seeds[s_base + k]);
				++n_out;
			};
			if (tree.size) tree.traverse(tree.root, emit);
		}
		out.per_read.push_back(n_out);
	}
}
} // namespace

// contig table from <prefix>.ann (bns_restore_core, bntseq.c:97-140: "l_pac n_seqs seed", then per sequence "gi name [comment]" and
// "offset len n_ambs"), ALT flags from <prefix>.alt where that file exists (bns_restore, bntseq.c:178-207: the first field of every
// line that does not start with '@' names an ALT contig; a last line without a newline is not looked at)
int cs_load_contigs_(const char *prefix, cs_refseq_view &ref)
{
	FILE *fp = fopen((std::string(prefix) + ".ann").c_str(), "r");
	if (!fp) return cs_fail_(CS_EIO, std::string("cannot read ") + prefix + ".ann");
	ref = cs_refseq_view();
	long long l_pac = 0; int n_seqs = 0; unsigned seed = 0;
	bool ok = fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) == 3 && n_seqs > 0;
	ref.l_pac = l_pac;
	std::vector<char> line(1 << 16);
	std::unordered_map<std::string, int> by_name; // (the reference's hash keeps the last contig of a repeated name)
	if (ok) ok = fgets(line.data(), (int)line.size(), fp) != nullptr; // rest of the first line
	for (int i = 0; ok && i < n_seqs; ++i) {
		long long off = 0; int len = 0, n_ambs = 0; unsigned gi = 0;
		std::vector<char> name(8192);
		ok = fscanf(fp, "%u%8191s", &gi, name.data()) == 2;                               // "gi name", then the comment up to the end of the line
		ok = ok && fgets(line.data(), (int)line.size(), fp) != nullptr;
		ok = ok && fscanf(fp, "%lld%d%d", &off, &len, &n_ambs) == 3;
		if (ok) { by_name[name.data()] = i; ref.offset.push_back(off); ref.len.push_back(len); ok = fgets(line.data(), (int)line.size(), fp) != nullptr || i == n_seqs - 1; }
	}
	fclose(fp);
	if (!ok) return cs_fail_(CS_EIO, std::string(prefix) + ".ann is malformed");
	ref.is_alt.assign((size_t)n_seqs, 0);
	if ((fp = fopen((std::string(prefix) + ".alt").c_str(), "r")) != nullptr) {
		std::string field; bool in_field = true; int ch;
		while ((ch = fgetc(fp)) != EOF) {
			if (ch == '\n') {
				if (in_field || !field.empty()) { if (!field.empty() && field[0] != '@') { auto it = by_name.find(field); if (it != by_name.end()) ref.is_alt[(size_t)it->second] = 1; } }
				field.clear(); in_field = true;
			} else if (in_field && (ch == '\t' || ch == '\r')) in_field = false;
			else if (in_field) field.push_back((char)ch);
		}
		fclose(fp);
	}
	return CS_OK;
}
extern "C" int cs_chainer_create(const char *prefix, cs_chainer_t **out)
{
	if (!prefix || !out) return cs_fail_(CS_EINVAL, "cs_chainer_create: null argument");
	*out = nullptr;
	cs_chainer *c = new cs_chainer();
	const int rc = cs_load_contigs_(prefix, c->ref);
	if (rc != CS_OK) { delete c; return rc; }
	c->prefix = prefix;
	*out = c;
	return CS_OK;
}
extern "C" void cs_chainer_destroy(cs_chainer_t *c) { delete c; }

extern "C" void cs_chain_params_default(cs_chain_params_t *p)
{
	if (!p) return;
	p->w = 100; p->max_chain_gap = 10000; p->min_seed_len = 19; p->max_occ = 500; // mem_opt_init, comp_seed.cpp:26-58
}

extern "C" int cs_chain_batch(cs_chainer_t *c, const cs_chain_params_t *par, const cs_result_t *seeds, const uint64_t *read_offsets, int n_threads,
                              cs_chain_result_t *out)
{
	if (!c || !par || !seeds || !out || (seeds->n_reads > 0 && (!read_offsets || !seeds->mem_off || !seeds->seed_off))) return cs_fail_(CS_EINVAL, "cs_chain_batch: bad argument (seeds are needed: want_sal = 1)");
	const int64_t n = seeds->n_reads;
	int T = std::max(1, std::min(n_threads, 256));
	if (n < 1024) T = 1;
	// chunks of reads, handed out by a counter (cs_for_chunks_); every chunk's share goes to where the prefix sums over the chunks say
	const int64_t CH = cs_chunk_reads_(n, T), K = (n + CH - 1) / CH;
	std::vector<ReadOut> part((size_t)K);
	cs_for_chunks_(T, K, [&](int64_t k) { chain_range(c->ref, *par, *seeds, read_offsets, k * CH, std::min(n, (k + 1) * CH), part[(size_t)k]); });
	std::vector<size_t> cb((size_t)K + 1, 0), sb((size_t)K + 1, 0);
	for (int64_t k = 0; k < K; ++k) { cb[(size_t)k + 1] = cb[(size_t)k] + part[(size_t)k].chains.size(); sb[(size_t)k + 1] = sb[(size_t)k] + part[(size_t)k].seeds.size(); }
	c->chains.resize(cb[(size_t)K]); c->cseeds.resize(sb[(size_t)K]); c->chain_off.resize((size_t)n + 1); c->cseed_off.resize(cb[(size_t)K] + 1);
	c->chain_off[0] = 0; c->cseed_off[0] = 0;
	cs_for_chunks_(T, K, [&](int64_t k) {
		const ReadOut &p = part[(size_t)k];
		uint64_t co = cb[(size_t)k], so = sb[(size_t)k];
		int64_t r = k * CH;
		for (uint32_t q : p.per_read) { co += q; c->chain_off[(size_t)++r] = co; }
		for (size_t i = 0; i < p.chains.size(); ++i) { so += (uint64_t)p.chains[i].n_seeds; c->cseed_off[cb[(size_t)k] + i + 1] = so; }
		if (!p.chains.empty()) memcpy(c->chains.data() + cb[(size_t)k], p.chains.data(), p.chains.size() * sizeof(cs_chain_t));
		if (!p.seeds.empty()) memcpy(c->cseeds.data() + sb[(size_t)k], p.seeds.data(), p.seeds.size() * sizeof(cs_seed_t));
	});
	out->n_reads = n; out->n_chains = c->chains.size(); out->n_seeds = c->cseeds.size();
	out->chain_off = c->chain_off.data(); out->chains = c->chains.data(); out->cseed_off = c->cseed_off.data(); out->cseeds = c->cseeds.data();
	return CS_OK;
}
