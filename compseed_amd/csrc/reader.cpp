// reader.cpp -- reordered-reads ingest: the reader step of the reference's pipeline (input_reorder_reads,
// main.cpp:36-58; FASTQ sniffing main.cpp:399-419; -K chunking main.cpp:54,437), producing chunks directly in the form the engine
// takes: bases back to back in pinned host memory + offsets, ready for cs_engine_submit.  Host code.
//
// The reference reads a chunk with gzgets + strdup per read (one malloc and two copies per 150 bytes); at the engine's rate a chunk
// of 10 M reads is consumed in under 0.1 s, so the reader scans 64-MB blocks with memchr and copies every read once, into the
// buffer the upload engine reads from.  Two chunk buffers alternate: a chunk stays intact while the next one is being read, which is
// what a two-deep submit / collect pipeline needs (read chunk n+2 after chunk n has been collected).
#include "cs_internal.hpp"

#include <hip/hip_runtime.h>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Pinned { // pinned when a device is present (uploads at link speed), plain memory otherwise
	uint8_t *p = nullptr; size_t cap = 0; bool pinned = false;
	bool reserve(size_t n, size_t keep)
	{
		if (n <= cap) return true;
		size_t want = n + n / 4 + 4096;
		uint8_t *q = nullptr; bool pin = true;
		if (hipHostMalloc((void **)&q, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); q = (uint8_t *)malloc(want); pin = false; }
		if (!q) return false;
		if (p && keep) memcpy(q, p, keep);
		release();
		p = q; cap = want; pinned = pin;
		return true;
	}
	void release() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; }
};
} // namespace

struct cs_reader {
	gzFile fp = nullptr; bool fastq = false; int64_t chunk_bases = 0;
	std::vector<uint8_t> blk; size_t pos = 0, end = 0; bool eof = false;
	Pinned bases[2], offs[2]; int cur = 0;
	uint64_t n_read = 0; int line_in_rec = 0; // FASTQ: position inside the 4-line record
};

extern "C" int cs_reader_open(const char *path, int64_t chunk_bases, cs_reader_t **out)
{
	if (!path || !out || chunk_bases < 1) return cs_fail_(CS_EINVAL, "cs_reader_open: bad argument");
	*out = nullptr;
	gzFile fp = gzopen(path, "r"); // reads plain files as they are
	if (!fp) return cs_fail_(CS_EIO, std::string("fail to open file `") + path + "'.");
	gzbuffer(fp, 1 << 20);
	cs_reader *r = new cs_reader();
	r->fp = fp; r->chunk_bases = chunk_bases; r->blk.resize((size_t)64 << 20);
	int n = gzread(fp, r->blk.data(), (unsigned)r->blk.size());
	r->end = n > 0 ? (size_t)n : 0; r->eof = n <= 0;
	r->fastq = r->end > 0 && r->blk[0] == '@'; // main.cpp:399-406
	*out = r;
	return CS_OK;
}

extern "C" int cs_reader_next(cs_reader_t *r, const uint8_t **bases, const uint64_t **offsets, int64_t *n_reads)
{
	if (!r || !bases || !offsets || !n_reads) return cs_fail_(CS_EINVAL, "cs_reader_next: null argument");
	Pinned &B = r->bases[r->cur], &O = r->offs[r->cur];
	r->cur ^= 1;
	size_t nb = 0; int64_t n = 0;
	if (!B.reserve((size_t)r->chunk_bases + 65536 + 64, 0) || !O.reserve(((size_t)r->chunk_bases / 64 + 1024) * 8, 0)) return cs_fail_(CS_ENOMEM, "cs_reader_next: out of memory");
	uint64_t *off = (uint64_t *)O.p;
	off[0] = 0;
	bool done = false, partial = false; // partial: a line has begun in an earlier block and has not ended yet
	while (!done) {
		bool line_ends = false, take = !r->fastq || r->line_in_rec == 1; // FASTQ: @name / SEQUENCE / + / quality
		if (r->pos == r->end && !r->eof) {
			int got = gzread(r->fp, r->blk.data(), (unsigned)r->blk.size());
			r->pos = 0; r->end = got > 0 ? (size_t)got : 0;
			if (got <= 0) r->eof = true;
		}
		if (r->pos == r->end) { // end of input: a last line without a terminator still counts
			if (!partial) break;
			line_ends = true;
		} else {
			const uint8_t *p = r->blk.data() + r->pos;
			const uint8_t *q = (const uint8_t *)memchr(p, '\n', r->end - r->pos);
			const size_t len = q ? (size_t)(q - p) : r->end - r->pos;
			if (take) {
				if (nb + len + 64 >= B.cap) { if (!B.reserve(nb + len + 65536, nb)) return cs_fail_(CS_ENOMEM, "cs_reader_next: out of memory"); }
				memcpy(B.p + nb, p, len);
				nb += len;
			}
			r->pos += len + (q ? 1 : 0);
			partial = q == nullptr;
			line_ends = q != nullptr;
		}
		if (!line_ends) continue; // the line goes on in the next block (its first part is copied already)
		partial = false;
		if (take) {
			while (nb > off[n] && B.p[nb - 1] == '\r') --nb;
			if (nb - off[n] >= 65535) return cs_fail_(CS_ERANGE, "Read length of " + std::to_string(nb - off[n]) + " exceeds the limit 65535"); // main.cpp:83-86
			if (((size_t)n + 2) * 8 > O.cap) { if (!O.reserve(((size_t)n + 2) * 8 * 2, ((size_t)n + 1) * 8)) return cs_fail_(CS_ENOMEM, "cs_reader_next: out of memory"); off = (uint64_t *)O.p; }
			off[++n] = nb;
			if ((int64_t)nb >= r->chunk_bases && (n & 1) == 0) done = true; // main.cpp:54: a chunk ends on an even read count
		}
		if (r->fastq) r->line_in_rec = (r->line_in_rec + 1) & 3;
	}
	memset(B.p + nb, 0, 64); // (padding: the engine copies whole words)
	r->n_read += (uint64_t)n;
	*bases = B.p; *offsets = off; *n_reads = n;
	return CS_OK;
}

extern "C" void cs_reader_close(cs_reader_t *r)
{
	if (!r) return;
	if (r->fp) gzclose(r->fp);
	for (int k = 0; k < 2; ++k) { r->bases[k].release(); r->offs[k].release(); }
	delete r;
}
