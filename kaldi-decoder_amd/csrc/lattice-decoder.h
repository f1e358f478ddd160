// kaldi-decoder_amd/csrc/lattice-decoder.h -- the lattice-generating decoders, host side.
//
// LatticeSimpleDecoder mirrors the reference class
// (/root/reference/kaldi-decoder/csrc/lattice-simple-decoder.h:24-158): same config
// struct (fields, defaults, Check, ToString) and methods -- InitDecoding, Decode,
// FinalizeDecoding, GetBestPath, GetRawLattice, FinalRelativeCost, GetOptions,
// NumFramesDecoded.  LatticeFasterDecoder is the class the reference only
// declares: its config is LatticeFasterDecoderConfig verbatim
// (lattice-faster-decoder.h:23-134); its search is LatticeSimpleDecoder's lattice
// semantics with FasterDecoder's GetCutoff (max_active / min_active / beam_delta),
// and it adds AdvanceDecoding for streaming.  Both run on the GPU through the C
// ABI (kd_batch_create_lattice and friends); GetBestPath is the tropical shortest
// path through the raw lattice, computed on the host from the exported arrays.
#ifndef KALDI_DECODER_AMD_CSRC_LATTICE_DECODER_H_
#define KALDI_DECODER_AMD_CSRC_LATTICE_DECODER_H_

#include <cstdint>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "decodable-itf.h"
#include "faster-decoder.h"
#include "fst.h"

namespace kaldi_decoder {

struct LatticeSimpleDecoderConfig {
  float beam;
  float lattice_beam;
  int32_t prune_interval;
  bool determinize_lattice;  // not inspected by the decoder
  bool prune_lattice;
  float beam_ratio;
  float prune_scale;

  LatticeSimpleDecoderConfig(float beam = 16.0, float lattice_beam = 10.0,
                             int32_t prune_interval = 25, bool determinize_lattice = true,
                             bool prune_lattice = true, float beam_ratio = 0.9,
                             float prune_scale = 0.1)
      : beam(beam), lattice_beam(lattice_beam), prune_interval(prune_interval),
        determinize_lattice(determinize_lattice), prune_lattice(prune_lattice),
        beam_ratio(beam_ratio), prune_scale(prune_scale) {}

  void Check() const {  // lattice-simple-decoder.h:63-66
    if (!(beam > 0.0 && lattice_beam > 0.0 && prune_interval > 0))
      throw std::runtime_error("Check failed!\nx: beam > 0.0 && lattice_beam > 0.0 && prune_interval > 0");
  }

  std::string ToString() const {  // same text as lattice-simple-decoder.h:68-83
    std::ostringstream os;
    os << "LatticeSimpleDecoderConfig(";
    os << "beam=" << beam << ", ";
    os << "lattice_beam=" << lattice_beam << ", ";
    os << "prune_interval=" << prune_interval << ", ";
    os << "determinize_lattice=" << (determinize_lattice ? "True" : "False") << ", ";
    os << "prune_lattice=" << prune_lattice << ", ";
    os << "beam_ratio=" << beam_ratio << ", ";
    os << "prune_scale=" << prune_scale << ")";
    return os.str();
  }
};

struct LatticeFasterDecoderConfig {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float lattice_beam;
  int32_t prune_interval;
  bool determinize_lattice;
  float beam_delta;
  float hash_ratio;
  float prune_scale;
  int32_t memory_pool_tokens_block_size;
  int32_t memory_pool_links_block_size;

  LatticeFasterDecoderConfig(float beam = 16.0,
                             int32_t max_active = std::numeric_limits<int32_t>::max(),
                             int32_t min_active = 200, float lattice_beam = 10.0,
                             int32_t prune_interval = 25, bool determinize_lattice = true,
                             float beam_delta = 0.5, float hash_ratio = 2.0,
                             float prune_scale = 0.1,
                             int32_t memory_pool_tokens_block_size = 1 << 8,
                             int32_t memory_pool_links_block_size = 1 << 8)
      : beam(beam), max_active(max_active), min_active(min_active), lattice_beam(lattice_beam),
        prune_interval(prune_interval), determinize_lattice(determinize_lattice),
        beam_delta(beam_delta), hash_ratio(hash_ratio), prune_scale(prune_scale),
        memory_pool_tokens_block_size(memory_pool_tokens_block_size),
        memory_pool_links_block_size(memory_pool_links_block_size) {}

  std::string ToString() const {  // same text as lattice-faster-decoder.h:106-127
    std::ostringstream os;
    os << "LatticeFasterDecoderConfig(";
    os << "beam=" << beam << ", ";
    os << "max_active=" << max_active << ", ";
    os << "min_active=" << min_active << ", ";
    os << "lattice_beam=" << lattice_beam << ", ";
    os << "prune_interval=" << prune_interval << ", ";
    os << "determinize_lattice=" << (determinize_lattice ? "True" : "False") << ", ";
    os << "beam_delta=" << beam_delta << ", ";
    os << "hash_ratio=" << hash_ratio << ", ";
    os << "prune_scale=" << prune_scale << ", ";
    os << "memory_pool_tokens_block_size=" << memory_pool_tokens_block_size << ", ";
    os << "memory_pool_links_block_size=" << memory_pool_links_block_size << ")";
    return os.str();
  }
  // Check() is enforced by kd_batch_create_lattice (lattice-faster-decoder.h:128-133).
};

// Lattice post-processing on the output side.
// Tropical shortest path over LatticeWeight: compare graph + acoustic, then graph
// (kaldifst lattice-weight.h Compare; fst::ShortestPath is called at
// lattice-simple-decoder.cc:578 and lives in OpenFst, outside the reference tree:
// parity unpinned).  Result is a linear lattice; false / empty if no final state
// is reachable.
bool ShortestPath(const Lattice &in, Lattice *out);

// `max_batch` lattice decoders over one graph.
class BatchLatticeDecoder {
 public:
  BatchLatticeDecoder(const Fst &fst, const LatticeFasterDecoderConfig &config, int32_t max_batch,
                      const DeviceOptions &dev = DeviceOptions());
  ~BatchLatticeDecoder();
  BatchLatticeDecoder(const BatchLatticeDecoder &) = delete;
  BatchLatticeDecoder &operator=(const BatchLatticeDecoder &) = delete;

  void InitDecoding(int32_t batch);
  void AdvanceDecoding(const std::vector<DecodableInterface *> &decodables,
                       int32_t max_num_frames = -1);
  // InitDecoding + all frames + FinalizeDecoding; per utterance "a final state was reached"
  std::vector<bool> Decode(const std::vector<DecodableInterface *> &decodables);
  void FinalizeDecoding();
  std::vector<float> FinalRelativeCost() const;
  std::vector<int32_t> NumFramesDecoded() const;
  std::vector<bool> GetRawLattice(std::vector<Lattice> *out, bool use_final_probs = true,
                                  std::vector<std::vector<int32_t>> *state_frame = nullptr) const;
  std::vector<bool> GetBestPath(std::vector<Lattice> *out, bool use_final_probs = true) const;
  int32_t Batch() const { return batch_; }
  float LastKernelMs() const;
  std::vector<int64_t> Stats() const;

 private:
  const Fst &fst_;
  kd_batch *b_ = nullptr;
  int32_t max_batch_, batch_ = 0;
  std::vector<float> scratch_;
};

class LatticeFasterDecoder {
 public:
  LatticeFasterDecoder(const Fst &fst, const LatticeFasterDecoderConfig &config,
                       const DeviceOptions &dev = DeviceOptions())
      : config_(config), impl_(fst, config, 1, dev) {}
  const LatticeFasterDecoderConfig &GetOptions() const { return config_; }
  int32_t NumFramesDecoded() const { return impl_.NumFramesDecoded()[0]; }
  float FinalRelativeCost() const { return impl_.FinalRelativeCost()[0]; }
  bool Decode(DecodableInterface *d) { return impl_.Decode({d})[0]; }
  void InitDecoding() { impl_.InitDecoding(1); }
  void AdvanceDecoding(DecodableInterface *d, int32_t max_num_frames = -1) {
    impl_.AdvanceDecoding({d}, max_num_frames);
  }
  void FinalizeDecoding() { impl_.FinalizeDecoding(); }
  bool GetBestPath(Lattice *lat, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetBestPath(&v, use_final_probs)[0];
    *lat = std::move(v[0]);
    return ok;
  }
  bool GetRawLattice(Lattice *lat, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetRawLattice(&v, use_final_probs)[0];
    *lat = std::move(v[0]);
    return ok;
  }

 private:
  LatticeFasterDecoderConfig config_;
  mutable BatchLatticeDecoder impl_;
};

class LatticeSimpleDecoder {
 public:
  LatticeSimpleDecoder(const Fst &fst, const LatticeSimpleDecoderConfig &config,
                       const DeviceOptions &dev = DeviceOptions())
      : config_((config.Check(), config)), impl_(fst, ToFaster(config), 1, dev) {}
  const LatticeSimpleDecoderConfig &GetOptions() const { return config_; }
  int32_t NumFramesDecoded() const { return impl_.NumFramesDecoded()[0]; }
  float FinalRelativeCost() const { return impl_.FinalRelativeCost()[0]; }
  bool Decode(DecodableInterface *d) { return impl_.Decode({d})[0]; }
  void InitDecoding() { impl_.InitDecoding(1); }
  void FinalizeDecoding() { impl_.FinalizeDecoding(); }
  bool GetBestPath(Lattice *lat, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetBestPath(&v, use_final_probs)[0];
    *lat = std::move(v[0]);
    return ok;
  }
  bool GetRawLattice(Lattice *lat, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetRawLattice(&v, use_final_probs)[0];
    *lat = std::move(v[0]);
    return ok;
  }

 private:
  static LatticeFasterDecoderConfig ToFaster(const LatticeSimpleDecoderConfig &c) {
    // no max_active / min_active: every token is expanded, plain beam
    LatticeFasterDecoderConfig f;
    f.beam = c.beam;
    f.max_active = std::numeric_limits<int32_t>::max();
    f.min_active = 0;
    f.lattice_beam = c.lattice_beam;
    f.prune_interval = c.prune_interval;
    f.determinize_lattice = c.determinize_lattice;
    // prune_scale is not range-checked by the reference's LatticeSimpleDecoderConfig;
    // it is unused here (all pruning is exact, in FinalizeDecoding)
    f.prune_scale = (c.prune_scale > 0.0f && c.prune_scale < 1.0f) ? c.prune_scale : 0.1f;
    return f;
  }
  LatticeSimpleDecoderConfig config_;
  mutable BatchLatticeDecoder impl_;
};

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_LATTICE_DECODER_H_
