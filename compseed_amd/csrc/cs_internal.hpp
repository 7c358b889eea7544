// cs_internal.hpp -- declarations shared by the translation units of libcompseed_amd.so (not part of the C ABI)
#pragma once
#include "../../include/compseed_amd.h"
#include <string>
#include <vector>

struct cs_index { // host copy of an index: the arrays behind a cs_index_view_t
	cs_index_view_t v;
	std::vector<uint32_t> bwt;
	std::vector<uint64_t> sa;
};

int cs_fail_(int code, const std::string &msg); // records the calling thread's error message, returns code

// contig table of an index (<prefix>.ann, ALT flags from <prefix>.alt): shared by the chainer and the extension driver
struct cs_refseq_view { int64_t l_pac; std::vector<int64_t> offset; std::vector<int32_t> len; std::vector<uint8_t> is_alt; };
int cs_load_contigs_(const char *prefix, cs_refseq_view &ref);

// the reads of a part as 16-byte records of 32 bases, made on host threads (host_pack.cpp; bit-identical to pack_reads_kernel's)
void cs_pack_reads_host_(const uint8_t *bases, const uint64_t *offsets, int64_t r0, int64_t n, int64_t lo, int64_t hi, void *rec_out, int threads, int force_scalar);
