// kaldi-decoder_amd/csrc/faster-decoder.h -- FasterDecoder, host side.
//
// Same public surface as the reference class
// (/root/reference/kaldi-decoder/csrc/faster-decoder.h:24-107): options struct
// with the same fields and defaults, InitDecoding / AdvanceDecoding / Decode /
// ReachedFinal / GetBestPath / NumFramesDecoded / SetOptions.  The search itself
// runs on the GPU through the C ABI (include/kd_hip.h); this class only owns a
// kd_batch and marshals arguments.  BatchFasterDecoder is the same object for
// many utterances at once, which is how the GPU is meant to be fed.
#ifndef KALDI_DECODER_AMD_CSRC_FASTER_DECODER_H_
#define KALDI_DECODER_AMD_CSRC_FASTER_DECODER_H_

#include <cstdint>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "decodable-ctc.h"
#include "decodable-itf.h"
#include "fst.h"

namespace kaldi_decoder {

struct FasterDecoderOptions {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float beam_delta;
  float hash_ratio;

  /*implicit*/ FasterDecoderOptions(float beam = 16.0,
                                    int32_t max_active = std::numeric_limits<int32_t>::max(),
                                    int32_t min_active = 20, float beam_delta = 0.5,
                                    float hash_ratio = 2.0)
      : beam(beam), max_active(max_active), min_active(min_active), beam_delta(beam_delta),
        hash_ratio(hash_ratio) {}

  std::string ToString() const {  // same text as the reference, faster-decoder.h:51-62
    std::ostringstream os;
    os << "FasterDecoderOptions(";
    os << "beam=" << beam << ", ";
    os << "max_active=" << max_active << ", ";
    os << "min_active=" << min_active << ", ";
    os << "beam_delta=" << beam_delta << ", ";
    os << "hash_ratio=" << hash_ratio << ")";
    return os.str();
  }
};

// Device-memory sizing that has no counterpart in the reference; 0 = default.
struct DeviceOptions {
  int device = 0;
  int64_t arena_tokens = 0;
  int32_t table_slots = 0;
  int32_t max_frames = 0;
};

class BatchFasterDecoder {
 public:
  BatchFasterDecoder(const Fst &fst, const FasterDecoderOptions &config, int32_t max_batch,
                     const DeviceOptions &dev = DeviceOptions());
  // SimpleDecoder semantics (plain beam); see SimpleDecoder below
  BatchFasterDecoder(const Fst &fst, float simple_beam, int32_t max_batch, const DeviceOptions &dev);
  ~BatchFasterDecoder();
  BatchFasterDecoder(const BatchFasterDecoder &) = delete;
  BatchFasterDecoder &operator=(const BatchFasterDecoder &) = delete;

  void SetOptions(const FasterDecoderOptions &config);
  void InitDecoding(int32_t batch);
  // One decodable per utterance; all must have the same NumIndices().
  void AdvanceDecoding(const std::vector<DecodableInterface *> &decodables,
                       int32_t max_num_frames = -1);
  void Decode(const std::vector<DecodableInterface *> &decodables);
  std::vector<bool> ReachedFinal() const;
  std::vector<int32_t> NumFramesDecoded() const;
  // ok[i] is GetBestPath's return value for utterance i.
  std::vector<bool> GetBestPath(std::vector<Lattice> *out, bool use_final_probs = true) const;
  int32_t Batch() const { return batch_; }
  std::vector<int64_t> Stats() const;  // [batch][KD_NUM_STATS]
  float LastKernelMs() const;
  std::vector<float> FinalRelativeCost() const;

 private:
  const Fst &fst_;
  kd_batch *b_ = nullptr;
  FasterDecoderOptions config_;
  DeviceOptions dev_;
  int32_t max_batch_, batch_ = 0;
  std::vector<float> scratch_;  // read-out of decodables that are not matrix backed
};

class FasterDecoder {
 public:
  FasterDecoder(const Fst &fst, const FasterDecoderOptions &config,
                const DeviceOptions &dev = DeviceOptions())
      : impl_(fst, config, 1, dev) {}
  FasterDecoder(const FasterDecoder &) = delete;
  FasterDecoder &operator=(const FasterDecoder &) = delete;

  void SetOptions(const FasterDecoderOptions &config) { impl_.SetOptions(config); }
  void Decode(DecodableInterface *decodable) { impl_.Decode({decodable}); }
  bool ReachedFinal() const { return impl_.ReachedFinal()[0]; }
  bool GetBestPath(Lattice *fst_out, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetBestPath(&v, use_final_probs)[0];
    *fst_out = std::move(v[0]);
    return ok;
  }
  void InitDecoding() { impl_.InitDecoding(1); }
  void AdvanceDecoding(DecodableInterface *decodable, int32_t max_num_frames = -1) {
    impl_.AdvanceDecoding({decodable}, max_num_frames);
  }
  int32_t NumFramesDecoded() const { return impl_.NumFramesDecoded()[0]; }
  const BatchFasterDecoder &Impl() const { return impl_; }

 private:
  mutable BatchFasterDecoder impl_;
};

// SimpleDecoder (ref: /root/reference/kaldi-decoder/csrc/simple-decoder.h:24-85):
// same methods -- Decode (true if any token survived), ReachedFinal, GetBestPath,
// FinalRelativeCost, InitDecoding, AdvanceDecoding, NumFramesDecoded -- running
// the GPU search in its plain-beam mode (kd_batch_create_simple).
class SimpleDecoder {
 public:
  SimpleDecoder(const Fst &fst, float beam, const DeviceOptions &dev = DeviceOptions())
      : impl_(fst, beam, 1, dev) {}
  SimpleDecoder(const SimpleDecoder &) = delete;
  SimpleDecoder &operator=(const SimpleDecoder &) = delete;

  bool Decode(DecodableInterface *decodable) {  // simple-decoder.cc:24-28
    impl_.Decode({decodable});
    std::vector<Lattice> v;
    return impl_.GetBestPath(&v, true)[0];  // !cur_toks_.empty()
  }
  bool ReachedFinal() const { return impl_.ReachedFinal()[0]; }
  bool GetBestPath(Lattice *fst_out, bool use_final_probs = true) const {
    std::vector<Lattice> v;
    const bool ok = impl_.GetBestPath(&v, use_final_probs)[0];
    *fst_out = std::move(v[0]);
    return ok;
  }
  float FinalRelativeCost() const { return impl_.FinalRelativeCost()[0]; }
  void InitDecoding() { impl_.InitDecoding(1); }
  void AdvanceDecoding(DecodableInterface *decodable, int32_t max_num_frames = -1) {
    impl_.AdvanceDecoding({decodable}, max_num_frames);
  }
  int32_t NumFramesDecoded() const { return impl_.NumFramesDecoded()[0]; }

 private:
  mutable BatchFasterDecoder impl_;
};

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_FASTER_DECODER_H_
