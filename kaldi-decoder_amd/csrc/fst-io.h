// kaldi-decoder_amd/csrc/fst-io.h -- OpenFst binary files <-> Fst.
//
// The reference's callers load H/HL/HLG with kaldifst.StdVectorFst.read(path)
// (README.md:18-20 scripts; constructor overloads python/csrc/faster-decoder.cc:34-42);
// kaldifst / OpenFst are not part of this build, so the two container formats
// those files use are read (and written) here: "vector" and "const" FSTs over
// "standard" (tropical, float) arcs.  Layout follows OpenFst's published
// fst.h / vector-fst.h / const-fst.h / symbol-table.cc.  No file written by OpenFst
// exists in this image: tests/test_fst_io.py builds vector and const files byte by
// byte from that layout (symbol tables attached, the aligned const version 1 and the
// unaligned version 2, a vector file without a state count) independently of the
// writer here, and reads them back; a file from a real OpenFst build has not been read.
#ifndef KALDI_DECODER_AMD_CSRC_FST_IO_H_
#define KALDI_DECODER_AMD_CSRC_FST_IO_H_

#include <memory>
#include <string>

#include "fst.h"

namespace kaldi_decoder {

// Throws std::runtime_error on malformed input or unsupported fst/arc types.
std::unique_ptr<Fst> ReadFst(const std::string &path);

// fst_type is "vector" or "const".
void WriteFst(const Fst &fst, const std::string &path, const std::string &fst_type = "vector");

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_FST_IO_H_
