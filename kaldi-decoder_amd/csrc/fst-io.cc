// kaldi-decoder_amd/csrc/fst-io.cc -- see fst-io.h.

#include "fst-io.h"

#include <cstring>
#include <fstream>
#include <limits>
#include <vector>

namespace kaldi_decoder {
namespace {

constexpr int32_t kFstMagicNumber = 2125659606;
constexpr int32_t kSymbolTableMagicNumber = 2125658996;
constexpr int32_t kHasIsymbols = 1, kHasOsymbols = 2, kIsAligned = 4;
constexpr int kConstAlign = 16;  // ConstFst::kStaticProperties alignment of the arrays

// The whole file in memory (one read), then a cursor over it: a 500 MB HLG is three
// memcpy-sized passes, not 10^8 four-byte stream reads.
struct Reader {
  std::vector<char> buf;
  size_t pos = 0;
  std::string path;
  explicit Reader(const std::string &p) : path(p) {
    std::ifstream in(p, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("ReadFst: cannot open " + p);
    const std::streamoff n = in.tellg();
    buf.resize(static_cast<size_t>(n));
    in.seekg(0);
    if (n > 0) in.read(buf.data(), n);
    if (!in) throw std::runtime_error("ReadFst: cannot read " + p);
  }
  bool AtEnd() const { return pos >= buf.size(); }
  const char *Take(size_t n) {
    if (n > buf.size() - pos) throw std::runtime_error("ReadFst: unexpected end of file in " + path);
    const char *p = buf.data() + pos;
    pos += n;
    return p;
  }
  template <class T> T Get() {
    T v;
    std::memcpy(&v, Take(sizeof(T)), sizeof(T));
    return v;
  }
  std::string Str() {
    const int32_t n = Get<int32_t>();
    if (n < 0 || n > (1 << 20)) throw std::runtime_error("ReadFst: bad string length in " + path);
    return std::string(Take(static_cast<size_t>(n)), static_cast<size_t>(n));
  }
  void AlignTo(int a) { Take((a - pos % a) % a); }
  void SkipSymbolTable() {  // symbol-table.cc SymbolTableImpl::Read
    if (Get<int32_t>() != kSymbolTableMagicNumber)
      throw std::runtime_error("ReadFst: bad symbol table magic in " + path);
    Str();                  // name
    Get<int64_t>();         // available_key
    const int64_t size = Get<int64_t>();
    if (size < 0) throw std::runtime_error("ReadFst: bad symbol table size in " + path);
    for (int64_t i = 0; i < size; ++i) {
      Str();
      Get<int64_t>();
    }
  }
};

// 16-byte arc records {ilabel, olabel, weight, nextstate} -> four arrays
void SplitArcs(const char *rec, size_t n, std::vector<int32_t> &il, std::vector<int32_t> &ol,
               std::vector<float> &w, std::vector<int32_t> &ns) {
  const size_t at = il.size();
  il.resize(at + n); ol.resize(at + n); w.resize(at + n); ns.resize(at + n);
  for (size_t a = 0; a < n; ++a) {
    int32_t r[4];
    std::memcpy(r, rec + 16 * a, 16);
    il[at + a] = r[0];
    ol[at + a] = r[1];
    std::memcpy(&w[at + a], &r[2], 4);
    ns[at + a] = r[3];
  }
}

struct Writer {
  std::ofstream out;
  explicit Writer(const std::string &p) : out(p, std::ios::binary) {
    if (!out) throw std::runtime_error("WriteFst: cannot open " + p);
  }
  template <class T> void Put(T v) { out.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
  void Str(const std::string &s) {
    Put<int32_t>(static_cast<int32_t>(s.size()));
    out.write(s.data(), static_cast<std::streamsize>(s.size()));
  }
  void AlignTo(int a) {
    const std::streamoff pos = out.tellp();
    for (std::streamoff i = 0; i < (a - pos % a) % a; ++i) out.put('\0');
  }
};

}  // namespace

std::unique_ptr<Fst> ReadFst(const std::string &path) {
  Reader r(path);
  // FstHeader::Read (fst.cc)
  if (r.Get<int32_t>() != kFstMagicNumber)
    throw std::runtime_error("ReadFst: " + path + " is not an OpenFst binary file (bad magic)");
  const std::string fst_type = r.Str(), arc_type = r.Str();
  const int32_t version = r.Get<int32_t>();
  const int32_t flags = r.Get<int32_t>();
  r.Get<uint64_t>();  // properties
  const int64_t start = r.Get<int64_t>();
  const int64_t num_states = r.Get<int64_t>();
  const int64_t num_arcs_hdr = r.Get<int64_t>();
  if (arc_type != "standard")
    throw std::runtime_error("ReadFst: arc type '" + arc_type + "' (only 'standard' is supported)");
  if (fst_type != "vector" && fst_type != "const")
    throw std::runtime_error("ReadFst: fst type '" + fst_type + "' (only 'vector' and 'const')");
  // (a VectorFst written to a stream that cannot seek carries num_states = -1 and
  // is read until the file ends: vector-fst.h VectorFstImpl::Read)
  const bool until_eof = fst_type == "vector" && num_states == -1;
  if (!until_eof && (num_states < 0 || num_states >= (1ll << 31)))
    throw std::runtime_error("ReadFst: bad state count in " + path);
  if (flags & kHasIsymbols) r.SkipSymbolTable();
  if (flags & kHasOsymbols) r.SkipSymbolTable();

  std::vector<int32_t> row_ptr(1, 0), il, ol, ns;
  std::vector<float> w, fin;
  int64_t n_states = until_eof ? 0 : num_states;
  if (fst_type == "vector") {
    // VectorFstImpl::Read: per state the final weight, the arc count, the arcs
    if (!until_eof) { row_ptr.reserve(static_cast<size_t>(num_states) + 1); fin.reserve(num_states); }
    for (int64_t s = 0; until_eof ? !r.AtEnd() : s < num_states; ++s) {
      fin.push_back(r.Get<float>());
      const int64_t narcs = r.Get<int64_t>();
      if (narcs < 0 || static_cast<uint64_t>(narcs) + il.size() >= (1ull << 31))
        throw std::runtime_error("ReadFst: bad arc count in " + path);
      SplitArcs(r.Take(static_cast<size_t>(narcs) * 16), static_cast<size_t>(narcs), il, ol, w, ns);
      row_ptr.push_back(static_cast<int32_t>(il.size()));
    }
    n_states = static_cast<int64_t>(fin.size());
    if (n_states >= (1ll << 31)) throw std::runtime_error("ReadFst: bad state count in " + path);
  } else {
    // ConstFstImpl::Read: state array {final, pos, narcs, niepsilons, noepsilons}, then arcs;
    // file version 1 is the aligned layout (const-fst.h kAlignedFileVersion)
    const bool aligned = (flags & kIsAligned) != 0 || version == 1;
    if (aligned) r.AlignTo(kConstAlign);
    if (num_arcs_hdr < 0 || num_arcs_hdr >= (1ll << 31))
      throw std::runtime_error("ReadFst: bad arc count in " + path);
    const char *st = r.Take(static_cast<size_t>(num_states) * 20);
    if (aligned) r.AlignTo(kConstAlign);
    SplitArcs(r.Take(static_cast<size_t>(num_arcs_hdr) * 16), static_cast<size_t>(num_arcs_hdr), il, ol, w, ns);
    row_ptr.resize(static_cast<size_t>(num_states) + 1);
    fin.resize(static_cast<size_t>(num_states));
    for (int64_t s = 0; s < num_states; ++s) {
      uint32_t rec[5];
      std::memcpy(rec, st + 20 * s, 20);
      std::memcpy(&fin[s], &rec[0], 4);
      if (rec[1] != static_cast<uint32_t>(row_ptr[s]))
        throw std::runtime_error("ReadFst: const FST arcs are not stored state by state");
      if (static_cast<uint64_t>(row_ptr[s]) + rec[2] > static_cast<uint64_t>(num_arcs_hdr))
        throw std::runtime_error("ReadFst: const FST state arc ranges exceed the arc array");
      row_ptr[s + 1] = row_ptr[s] + static_cast<int32_t>(rec[2]);
    }
  }
  const int64_t num_states_final = n_states;
  if (start >= num_states_final) throw std::runtime_error("ReadFst: start state out of range in " + path);
  for (int32_t n : ns)
    if (n < 0 || n >= num_states_final) throw std::runtime_error("ReadFst: arc nextstate out of range");
  return std::unique_ptr<Fst>(new Fst(static_cast<int32_t>(start), std::move(row_ptr), std::move(il),
                                      std::move(ol), std::move(w), std::move(ns), std::move(fin)));
}

void WriteFst(const Fst &fst, const std::string &path, const std::string &fst_type) {
  if (fst_type != "vector" && fst_type != "const")
    throw std::runtime_error("WriteFst: fst type must be 'vector' or 'const'");
  Writer wr(path);
  const int64_t S = fst.NumStates(), A = fst.NumArcs();
  const bool is_const = fst_type == "const";
  wr.Put<int32_t>(kFstMagicNumber);
  wr.Str(fst_type);
  wr.Str("standard");
  wr.Put<int32_t>(2);                              // file version of both containers
  wr.Put<int32_t>(is_const ? kIsAligned : 0);      // no symbol tables
  wr.Put<uint64_t>(0);                             // properties: unknown
  wr.Put<int64_t>(fst.Start());
  wr.Put<int64_t>(S);
  wr.Put<int64_t>(A);
  const auto &rp = fst.RowPtr();
  if (!is_const) {
    for (int64_t s = 0; s < S; ++s) {
      wr.Put<float>(fst.Final(static_cast<int32_t>(s)));
      wr.Put<int64_t>(rp[s + 1] - rp[s]);
      for (int32_t a = rp[s]; a < rp[s + 1]; ++a) {
        wr.Put<int32_t>(fst.Ilabel()[a]);
        wr.Put<int32_t>(fst.Olabel()[a]);
        wr.Put<float>(fst.Weight()[a]);
        wr.Put<int32_t>(fst.NextState()[a]);
      }
    }
    return;
  }
  wr.AlignTo(kConstAlign);
  for (int64_t s = 0; s < S; ++s) {
    uint32_t nie = 0, noe = 0;
    for (int32_t a = rp[s]; a < rp[s + 1]; ++a) {
      nie += fst.Ilabel()[a] == 0;
      noe += fst.Olabel()[a] == 0;
    }
    wr.Put<float>(fst.Final(static_cast<int32_t>(s)));
    wr.Put<uint32_t>(static_cast<uint32_t>(rp[s]));
    wr.Put<uint32_t>(static_cast<uint32_t>(rp[s + 1] - rp[s]));
    wr.Put<uint32_t>(nie);
    wr.Put<uint32_t>(noe);
  }
  wr.AlignTo(kConstAlign);
  for (int64_t a = 0; a < A; ++a) {
    wr.Put<int32_t>(fst.Ilabel()[a]);
    wr.Put<int32_t>(fst.Olabel()[a]);
    wr.Put<float>(fst.Weight()[a]);
    wr.Put<int32_t>(fst.NextState()[a]);
  }
}

}  // namespace kaldi_decoder
