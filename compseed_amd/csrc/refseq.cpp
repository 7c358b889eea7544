// refseq.cpp -- FASTA front end of the index builder and the .pac / .ann / .amb files (the part in front of
// and beside the suffix sort): what bns_fasta2bntseq and bns_dump do in the reference (FM_index/bntseq.c:236-330, 65-95), as
// bwa_idx_build drives them (FM_index/index_main.c:257-325).  Host code, no GPU involved; byte-identical files
// (tests/test_refseq.py compares with the fixture written by the reference's own bwaidx).
//
//   * records are FASTA (">name comment" + sequence lines) or FASTQ, plain or gzip;
//   * A/C/G/T in either case are codes 0..3; every other byte is ambiguous: it opens a "hole" (a run of the SAME byte continues
//     the hole, bntseq.c:252-266) and is replaced in the packed sequence by lrand48() & 3 after srand48(11) (bntseq.c:266,295) --
//     reproduced here with the generator's definition (48-bit LCG, a = 0x5DEECE66D, c = 0xB, seed << 16 | 0x330E, top 31 bits),
//     not with the C library's hidden state;
//   * .pac holds the forward strand, 4 bases per byte, first base in the top bits, then a 0 byte if l_pac % 4 == 0, then l_pac % 4.
#include "cs_internal.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct cs_refseq {
	struct Ann { std::string name, anno; uint64_t offset; uint32_t len, n_ambs; };
	struct Amb { uint64_t offset; uint32_t len; char amb; };
	std::vector<Ann> anns; std::vector<Amb> ambs;
	std::vector<uint8_t> codes; // forward strand, one base per byte, 0..3 (holes already replaced)
};

namespace {
struct Rand48 { // srand48 / lrand48 (POSIX): X' = (a X + c) mod 2^48, value = X >> 17
	uint64_t x;
	explicit Rand48(uint32_t seed) : x(((uint64_t)seed << 16) | 0x330Eull) {}
	uint32_t next() { x = (x * 0x5DEECE66Dull + 0xBull) & ((1ull << 48) - 1); return (uint32_t)(x >> 17); }
};
inline int nt4(unsigned char c)
{
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
struct LineReader {
	gzFile fp; std::vector<char> buf; size_t pos = 0, end = 0; bool eof = false;
	explicit LineReader(gzFile f) : fp(f), buf(1 << 20) {}
	bool line(std::string &out) // one line without its terminator; false at end of file
	{
		out.clear();
		for (;;) {
			if (pos == end) {
				if (eof) return !out.empty();
				int n = gzread(fp, buf.data(), (unsigned)buf.size());
				if (n <= 0) { eof = true; return !out.empty(); }
				pos = 0; end = (size_t)n;
			}
			const char *p = buf.data() + pos, *q = (const char *)memchr(p, '\n', end - pos);
			if (q) { out.append(p, q - p); pos = (size_t)(q - buf.data()) + 1; if (!out.empty() && out.back() == '\r') out.pop_back(); return true; }
			out.append(p, end - pos); pos = end;
		}
	}
};
} // namespace

extern "C" int cs_refseq_from_fasta(const char *path, cs_refseq_t **out)
{
	if (!path || !out) return cs_fail_(CS_EINVAL, "cs_refseq_from_fasta: null argument");
	*out = nullptr;
	gzFile fp = gzopen(path, "r");
	if (!fp) return cs_fail_(CS_EIO, std::string("cannot open ") + path);
	cs_refseq *R = new cs_refseq();
	Rand48 rng(11); // bns->seed (bntseq.c:294)
	LineReader rd(fp);
	std::string ln;
	bool have = rd.line(ln);
	while (have) {
		if (ln.empty()) { have = rd.line(ln); continue; }
		if (ln[0] != '>' && ln[0] != '@') { gzclose(fp); delete R; return cs_fail_(CS_EIO, std::string(path) + ": not FASTA / FASTQ"); }
		cs_refseq::Ann a;
		size_t sp = ln.find_first_of(" \t", 1);
		a.name = ln.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
		size_t cm = sp == std::string::npos ? std::string::npos : ln.find_first_not_of(" \t", sp);
		a.anno = cm == std::string::npos ? "(null)" : ln.substr(cm); // (kseq keeps the rest of the header line as the comment; bntseq.c:249)
		a.offset = R->codes.size(); a.len = 0; a.n_ambs = 0;
		int lasts = 0;
		size_t seq_len = 0;
		// sequence lines up to the next line that starts with '>', '@' (next record) or '+' (FASTQ qualities), as kseq_read does
		for (have = rd.line(ln); have && !(ln.size() && (ln[0] == '>' || ln[0] == '@' || ln[0] == '+')); have = rd.line(ln)) {
			for (unsigned char ch : ln) {
				if (ch == ' ' || ch == '\t') continue; // (kseq skips blanks inside sequence lines)
				int c = nt4(ch);
				if (c >= 4) {
					if (lasts == ch) ++R->ambs.back().len; // contiguous run of the same ambiguity code
					else { cs_refseq::Amb h = {a.offset + seq_len, 1, (char)ch}; R->ambs.push_back(h); ++a.n_ambs; }
					c = (int)(rng.next() & 3u);
				}
				lasts = ch;
				R->codes.push_back((uint8_t)c);
				++seq_len;
			}
		}
		if (have && ln[0] == '+') { // FASTQ: as many quality characters as bases, over as many lines as it takes
			size_t q = 0;
			for (have = rd.line(ln); have; have = rd.line(ln)) { q += ln.size(); if (q >= seq_len) { have = rd.line(ln); break; } }
		}
		if (seq_len > 0xffffffffull) { gzclose(fp); delete R; return cs_fail_(CS_ERANGE, "a sequence of 2^32 bases or more"); }
		a.len = (uint32_t)seq_len;
		R->anns.push_back(a);
	}
	gzclose(fp);
	if (R->anns.empty()) { delete R; return cs_fail_(CS_EIO, std::string(path) + ": no sequences"); }
	*out = R;
	return CS_OK;
}

extern "C" int cs_refseq_codes(const cs_refseq_t *r, const uint8_t **fwd_nt4, uint64_t *l_pac, int32_t *n_seqs, int32_t *n_holes)
{
	if (!r) return cs_fail_(CS_EINVAL, "null argument");
	if (fwd_nt4) *fwd_nt4 = r->codes.data();
	if (l_pac) *l_pac = r->codes.size();
	if (n_seqs) *n_seqs = (int32_t)r->anns.size();
	if (n_holes) *n_holes = (int32_t)r->ambs.size();
	return CS_OK;
}

extern "C" int cs_refseq_save(const cs_refseq_t *r, const char *prefix)
{
	if (!r || !prefix) return cs_fail_(CS_EINVAL, "cs_refseq_save: null argument");
	const std::string p(prefix);
	const uint64_t l_pac = r->codes.size();
	{ // .pac: forward strand only, as the second bns_fasta2bntseq call of bwa_idx_build leaves it (index_main.c:303-310; bntseq.c:314-325)
		std::vector<uint8_t> pac((l_pac >> 2) + 1, 0);
		for (uint64_t l = 0; l < l_pac; ++l) pac[l >> 2] |= (uint8_t)(r->codes[l] << ((~l & 3) << 1));
		FILE *fp = fopen((p + ".pac").c_str(), "wb");
		if (!fp) return cs_fail_(CS_EIO, "cannot write " + p + ".pac");
		size_t nb = (size_t)(l_pac >> 2) + ((l_pac & 3) == 0 ? 0 : 1);
		bool ok = fwrite(pac.data(), 1, nb, fp) == nb;
		uint8_t ct = 0;
		if (l_pac % 4 == 0) ok = fwrite(&ct, 1, 1, fp) == 1 && ok;
		ct = (uint8_t)(l_pac % 4);
		ok = fwrite(&ct, 1, 1, fp) == 1 && ok;
		ok = fclose(fp) == 0 && ok;
		if (!ok) return cs_fail_(CS_EIO, "short write on " + p + ".pac");
	}
	{ // .ann (bns_dump, bntseq.c:70-82)
		FILE *fp = fopen((p + ".ann").c_str(), "w");
		if (!fp) return cs_fail_(CS_EIO, "cannot write " + p + ".ann");
		fprintf(fp, "%lld %d %u\n", (long long)l_pac, (int)r->anns.size(), 11u);
		for (const auto &a : r->anns) {
			fprintf(fp, "%d %s", 0, a.name.c_str());
			if (!a.anno.empty()) fprintf(fp, " %s\n", a.anno.c_str()); else fprintf(fp, "\n");
			fprintf(fp, "%lld %d %d\n", (long long)a.offset, (int)a.len, (int)a.n_ambs);
		}
		if (fclose(fp) != 0) return cs_fail_(CS_EIO, "short write on " + p + ".ann");
	}
	{ // .amb (bntseq.c:84-93)
		FILE *fp = fopen((p + ".amb").c_str(), "w");
		if (!fp) return cs_fail_(CS_EIO, "cannot write " + p + ".amb");
		fprintf(fp, "%lld %d %u\n", (long long)l_pac, (int)r->anns.size(), (unsigned)r->ambs.size());
		for (const auto &h : r->ambs) fprintf(fp, "%lld %d %c\n", (long long)h.offset, (int)h.len, h.amb);
		if (fclose(fp) != 0) return cs_fail_(CS_EIO, "short write on " + p + ".amb");
	}
	return CS_OK;
}

extern "C" void cs_refseq_free(cs_refseq_t *r) { delete r; }

// bwa_idx_build (index_main.c:257-325): FASTA -> <prefix>.pac .ann .amb .bwt .sa, the suffix sort on GPU `device`
extern "C" int cs_index_build_fasta(const char *fasta, const char *prefix, int device)
{
	if (!fasta || !prefix) return cs_fail_(CS_EINVAL, "cs_index_build_fasta: null argument");
	cs_refseq_t *r = nullptr;
	int rc = cs_refseq_from_fasta(fasta, &r);
	if (rc != CS_OK) return rc;
	cs_index_t *ix = nullptr;
	rc = cs_index_build(r->codes.data(), r->codes.size(), device, &ix);
	if (rc == CS_OK) rc = cs_index_save(ix, prefix);
	if (rc == CS_OK) rc = cs_refseq_save(r, prefix);
	cs_index_free(ix);
	cs_refseq_free(r);
	return rc;
}
