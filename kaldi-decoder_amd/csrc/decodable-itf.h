// kaldi-decoder_amd/csrc/decodable-itf.h -- the Decodable boundary.
//
// Same four-method contract as the reference's DecodableInterface
// (/root/reference/kaldi-decoder/csrc/decodable-itf.h:65-102): the decoders only
// ever ask for LogLikelihood(frame, one-based index), NumFramesReady(),
// IsLastFrame(frame) and NumIndices().  The GPU search cannot call a virtual
// per arc, so a decodable that is not matrix backed (DecodableCtc) is read out
// into a [frames, NumIndices()] matrix by the host decoder before each advance.
#ifndef KALDI_DECODER_AMD_CSRC_DECODABLE_ITF_H_
#define KALDI_DECODER_AMD_CSRC_DECODABLE_ITF_H_

#include <cstdint>
#include <stdexcept>

namespace kaldi_decoder {

class DecodableInterface {
 public:
  virtual ~DecodableInterface() = default;

  // Log-likelihood of one-based `index` at zero-based `frame`; negated by the decoder.
  virtual float LogLikelihood(int32_t frame, int32_t index) = 0;

  // True for the last frame of a finished utterance.
  virtual bool IsLastFrame(int32_t frame) const = 0;

  // Frames available so far; the reference's default throws too.
  virtual int32_t NumFramesReady() const {
    throw std::runtime_error("NumFramesReady() not implemented for this decodable type.");
  }

  // Indices run from 1 to NumIndices() (0 is epsilon in the graph).
  virtual int32_t NumIndices() const = 0;
};

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_DECODABLE_ITF_H_
