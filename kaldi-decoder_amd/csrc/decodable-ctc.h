// kaldi-decoder_amd/csrc/decodable-ctc.h -- matrix-backed decodable.
//
// Mirrors DecodableCtc (/root/reference/kaldi-decoder/csrc/decodable-ctc.h:13-43):
// a row-major [num_rows, num_cols] float matrix of log-probs, `offset` = frame
// index of row 0 for chunked streaming, LogLikelihood(frame, index) =
// p[(frame - offset) * cols + index - 1].  Added here: the matrix may already
// live in GPU memory (a torch ROCm tensor), in which case the search reads it
// in place.
#ifndef KALDI_DECODER_AMD_CSRC_DECODABLE_CTC_H_
#define KALDI_DECODER_AMD_CSRC_DECODABLE_CTC_H_

#include <cassert>
#include <cstdint>
#include <vector>

#include "decodable-itf.h"

namespace kaldi_decoder {

class DecodableCtc : public DecodableInterface {
 public:
  // Copies a host matrix (the reference's FloatMatrix constructor copies too).
  static DecodableCtc Copy(const float *p, int32_t num_rows, int32_t num_cols, int32_t offset = 0) {
    DecodableCtc d(nullptr, num_rows, num_cols, offset);
    d.owned_.assign(p, p + static_cast<size_t>(num_rows) * num_cols);
    d.p_ = d.owned_.data();
    return d;
  }

  // Shares the caller's memory, which must outlive this object.  `on_device`
  // says the pointer is a HIP device pointer.
  DecodableCtc(const float *p, int32_t num_rows, int32_t num_cols, int32_t offset = 0,
               bool on_device = false, int64_t row_stride = -1)
      : p_(p), num_rows_(num_rows), num_cols_(num_cols), offset_(offset),
        row_stride_(row_stride < 0 ? num_cols : row_stride), on_device_(on_device) {}

  float LogLikelihood(int32_t frame, int32_t index) override {
    if (on_device_) throw std::runtime_error("LogLikelihood() on a device-resident DecodableCtc");
    assert(index >= 1);
    return p_[static_cast<int64_t>(frame - offset_) * row_stride_ + index - 1];
  }
  int32_t NumFramesReady() const override { return offset_ + num_rows_; }
  int32_t NumIndices() const override { return num_cols_; }
  bool IsLastFrame(int32_t frame) const override {
    assert(frame < NumFramesReady());
    return frame == NumFramesReady() - 1;
  }

  const float *Data() const { return p_; }
  int32_t NumRows() const { return num_rows_; }
  int32_t NumCols() const { return num_cols_; }
  int32_t Offset() const { return offset_; }
  int64_t RowStride() const { return row_stride_; }
  bool OnDevice() const { return on_device_; }

 private:
  std::vector<float> owned_;
  const float *p_ = nullptr;
  int32_t num_rows_, num_cols_, offset_;
  int64_t row_stride_;
  bool on_device_;
};

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_DECODABLE_CTC_H_
