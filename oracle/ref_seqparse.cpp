// TEST INFRASTRUCTURE: the reference's own sequence ingest -- its vendored bioparser (header-only, zlib) feeding
// racon::Sequence (src/sequence.cpp), compiled where they lie under /root/reference (oracle/Makefile target `seqparse`,
// output oracle/_ref/libvcseq.so).  Used to pin vechat_amd/seqio.py:read_sequences (names cut at the first white space,
// upper-casing, an all-'!' quality string dropped) and the reverse complement / reverse quality the window builder uses.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "sequence.hpp"
#include "bioparser/parser.hpp"
#include "bioparser/fasta_parser.hpp"
#include "bioparser/fastq_parser.hpp"

extern "C" {

// Writes one line per sequence: name \t data \t quality-or-* \t reverse complement \t reverse quality-or-* \n
// Returns the number of bytes needed (call again with a larger buffer if > cap), or -1 on an exception.
long vcref_parse_sequences(const char* path, int is_fastq, char* out, long cap) {
    try {
        std::unique_ptr<bioparser::Parser<racon::Sequence>> p = is_fastq
            ? bioparser::Parser<racon::Sequence>::Create<bioparser::FastqParser>(path)
            : bioparser::Parser<racon::Sequence>::Create<bioparser::FastaParser>(path);
        auto seqs = p->Parse(-1);
        std::string s;
        for (auto& q : seqs) {
            q->transmute(true, true, true);                 // keeps name and data, builds the reverse complement
            s += q->name(); s += '\t'; s += q->data(); s += '\t';
            s += q->quality().empty() ? std::string("*") : q->quality(); s += '\t';
            s += q->reverse_complement(); s += '\t';
            s += q->reverse_quality().empty() ? std::string("*") : q->reverse_quality(); s += '\n';
        }
        if ((long)s.size() <= cap && out) std::memcpy(out, s.data(), s.size());
        return (long)s.size();
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
