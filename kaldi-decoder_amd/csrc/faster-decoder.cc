// kaldi-decoder_amd/csrc/faster-decoder.cc -- argument marshalling between the
// reference-shaped FasterDecoder API and the C ABI.  No search logic lives here.

#include "faster-decoder.h"

#include <algorithm>

namespace kaldi_decoder {

static kd_faster_options ToC(const FasterDecoderOptions &c, const DeviceOptions &d) {
  kd_faster_options o;
  kd_faster_options_default(&o);
  o.beam = c.beam;
  o.max_active = c.max_active;
  o.min_active = c.min_active;
  o.beam_delta = c.beam_delta;
  o.hash_ratio = c.hash_ratio;
  o.arena_tokens = d.arena_tokens;
  o.table_slots = d.table_slots;
  o.max_frames = d.max_frames;
  return o;
}

BatchFasterDecoder::BatchFasterDecoder(const Fst &fst, const FasterDecoderOptions &config,
                                       int32_t max_batch, const DeviceOptions &dev)
    : fst_(fst), config_(config), dev_(dev), max_batch_(max_batch) {
  const kd_faster_options o = ToC(config, dev);
  KdCheck(kd_batch_create(fst.OnDevice(dev.device), &o, max_batch, nullptr, &b_));
}

BatchFasterDecoder::BatchFasterDecoder(const Fst &fst, float simple_beam, int32_t max_batch,
                                       const DeviceOptions &dev)
    : fst_(fst), config_(simple_beam, std::numeric_limits<int32_t>::max(), 0), dev_(dev),
      max_batch_(max_batch) {
  KdCheck(kd_batch_create_simple(fst.OnDevice(dev.device), simple_beam, dev.arena_tokens,
                                 dev.table_slots, dev.max_frames, max_batch, nullptr, &b_));
}

BatchFasterDecoder::~BatchFasterDecoder() { kd_batch_destroy(b_); }

std::vector<float> BatchFasterDecoder::FinalRelativeCost() const {
  std::vector<float> r(std::max(batch_, 1));
  KdCheck(kd_batch_final_relative_cost(b_, r.data()));
  r.resize(batch_);
  return r;
}

void BatchFasterDecoder::SetOptions(const FasterDecoderOptions &config) {
  config_ = config;
  const kd_faster_options o = ToC(config, dev_);
  KdCheck(kd_batch_set_options(b_, &o));
}

void BatchFasterDecoder::InitDecoding(int32_t batch) {
  KdCheck(kd_batch_init_decoding(b_, batch));
  batch_ = batch;
}

void BatchFasterDecoder::AdvanceDecoding(const std::vector<DecodableInterface *> &decodables,
                                         int32_t max_num_frames) {
  if (batch_ == 0) {
    // not initialised: let the library produce the reference's assertion text
    const float *none = nullptr;
    int32_t zero = 0;
    KdCheck(kd_batch_advance(b_, &none, &zero, nullptr, 1, 1, KD_MEM_HOST, max_num_frames));
    return;
  }
  if (static_cast<int32_t>(decodables.size()) != batch_)
    throw std::runtime_error("AdvanceDecoding: one decodable per utterance expected");
  std::vector<int32_t> decoded(batch_);
  KdCheck(kd_batch_num_frames_decoded(b_, decoded.data()));
  std::vector<const float *> ptr(batch_);
  std::vector<int32_t> rows(batch_), off(batch_);
  const int32_t cols = decodables[0]->NumIndices();
  int64_t stride = -1;
  bool all_ctc = true, any_dev = false, all_dev = true;
  for (int i = 0; i < batch_; ++i) {
    auto *d = dynamic_cast<DecodableCtc *>(decodables[i]);
    if (!d) { all_ctc = false; all_dev = false; continue; }
    any_dev |= d->OnDevice();
    all_dev &= d->OnDevice();
    if (stride < 0) stride = d->RowStride();
    if (d->RowStride() != stride) all_ctc = false;
  }
  if (any_dev && !all_dev)
    throw std::runtime_error("AdvanceDecoding: mix of host and device decodables");
  if (all_ctc) {
    for (int i = 0; i < batch_; ++i) {
      auto *d = static_cast<DecodableCtc *>(decodables[i]);
      if (d->NumCols() != cols) throw std::runtime_error("AdvanceDecoding: NumIndices() differ");
      ptr[i] = d->Data();
      rows[i] = d->NumRows();
      off[i] = d->Offset();
    }
    KdCheck(kd_batch_advance(b_, ptr.data(), rows.data(), off.data(), cols, stride,
                             all_dev ? KD_MEM_DEVICE : KD_MEM_HOST, max_num_frames));
    return;
  }
  // Generic decodables (e.g. Python subclasses): read out the frames this call
  // will consume through the virtual interface, then hand over a plain matrix.
  size_t total = 0;
  std::vector<size_t> start(batch_);
  for (int i = 0; i < batch_; ++i) {
    if (decodables[i]->NumIndices() != cols)
      throw std::runtime_error("AdvanceDecoding: NumIndices() differ");
    const int32_t ready = decodables[i]->NumFramesReady();
    if (!(ready >= decoded[i]))
      throw std::runtime_error("Check failed: num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, decoded[i] + max_num_frames);
    rows[i] = target - decoded[i];
    off[i] = decoded[i];
    start[i] = total;
    total += static_cast<size_t>(rows[i]) * cols;
  }
  scratch_.resize(std::max<size_t>(total, 1));
  for (int i = 0; i < batch_; ++i) {
    float *dst = scratch_.data() + start[i];
    for (int32_t f = 0; f < rows[i]; ++f)
      for (int32_t k = 1; k <= cols; ++k)
        dst[static_cast<size_t>(f) * cols + k - 1] = decodables[i]->LogLikelihood(decoded[i] + f, k);
    ptr[i] = dst;
  }
  KdCheck(kd_batch_advance(b_, ptr.data(), rows.data(), off.data(), cols, cols, KD_MEM_HOST, -1));
}

// Matrix-backed decodables of one layout (all DecodableCtc, all on the host or all on the
// device, one row stride): their pointers, rows and offsets for the one-launch Decode.
static bool AllCtc(const std::vector<DecodableInterface *> &decodables, std::vector<const float *> *ptr,
                   std::vector<int32_t> *rows, std::vector<int32_t> *off, int32_t *cols,
                   int64_t *stride, bool *on_device) {
  const size_t n = decodables.size();
  if (n == 0) return false;
  ptr->resize(n); rows->resize(n); off->resize(n);
  *cols = decodables[0]->NumIndices();
  *stride = -1;
  for (size_t i = 0; i < n; ++i) {
    auto *d = dynamic_cast<DecodableCtc *>(decodables[i]);
    if (!d || d->NumCols() != *cols) return false;
    if (i == 0) { *stride = d->RowStride(); *on_device = d->OnDevice(); }
    if (d->RowStride() != *stride || d->OnDevice() != *on_device) return false;
    (*ptr)[i] = d->Data();
    (*rows)[i] = d->NumRows();
    (*off)[i] = d->Offset();
  }
  return true;
}

void BatchFasterDecoder::Decode(const std::vector<DecodableInterface *> &decodables) {
  // faster-decoder.cc:121-124.  Matrix-backed decodables: InitDecoding, every frame,
  // ReachedFinal and GetBestPath in ONE kernel launch; others go call by call.
  std::vector<const float *> ptr;
  std::vector<int32_t> rows, off;
  int32_t cols = 0;
  int64_t stride = 0;
  bool dev = false;
  if (AllCtc(decodables, &ptr, &rows, &off, &cols, &stride, &dev)) {
    KdCheck(kd_batch_decode(b_, static_cast<int32_t>(decodables.size()), ptr.data(), rows.data(),
                            off.data(), cols, stride, dev ? KD_MEM_DEVICE : KD_MEM_HOST,
                            KD_DECODE_BEST_PATH));
    batch_ = static_cast<int32_t>(decodables.size());
    return;
  }
  InitDecoding(static_cast<int32_t>(decodables.size()));
  AdvanceDecoding(decodables);
}

std::vector<bool> BatchFasterDecoder::ReachedFinal() const {
  std::vector<int32_t> r(std::max(batch_, 1));
  KdCheck(kd_batch_reached_final(b_, r.data()));
  return std::vector<bool>(r.begin(), r.begin() + batch_);
}

std::vector<int32_t> BatchFasterDecoder::NumFramesDecoded() const {
  std::vector<int32_t> r(std::max(batch_, 1), -1);
  if (batch_ == 0) return {-1};  // the reference starts at -1 (faster-decoder.cc:23)
  KdCheck(kd_batch_num_frames_decoded(b_, r.data()));
  return r;
}

std::vector<bool> BatchFasterDecoder::GetBestPath(std::vector<Lattice> *out,
                                                  bool use_final_probs) const {
  std::vector<int32_t> len(std::max(batch_, 1));
  KdCheck(kd_batch_best_path_prepare(b_, use_final_probs ? 1 : 0, len.data()));
  std::vector<int64_t> offs(batch_);
  int64_t total = 0;
  for (int i = 0; i < batch_; ++i) {
    offs[i] = total;
    total += std::max(len[i], 0);
  }
  const size_t n = static_cast<size_t>(std::max<int64_t>(total, 1));
  std::vector<int32_t> il(n), ol(n), st(n);
  std::vector<float> gc(n), ac(n), fw(2 * static_cast<size_t>(batch_));
  KdCheck(kd_batch_best_path_copy(b_, offs.data(), il.data(), ol.data(), st.data(), gc.data(),
                                  ac.data(), fw.data()));
  out->assign(batch_, Lattice());
  std::vector<bool> ok(batch_);
  for (int i = 0; i < batch_; ++i) {
    Lattice &lat = (*out)[i];
    ok[i] = len[i] >= 0;
    if (!ok[i]) continue;  // fst_out stays empty, faster-decoder.cc:386-389
    int32_t cur = lat.AddState();
    lat.SetStart(cur);
    for (int32_t k = 0; k < len[i]; ++k) {
      const size_t j = static_cast<size_t>(offs[i]) + k;
      const int32_t nxt = lat.AddState();
      lat.AddArc(cur, LatticeArc{il[j], ol[j], LatticeWeight{gc[j], ac[j]}, nxt});
      cur = nxt;
    }
    lat.SetFinal(cur, LatticeWeight{fw[2 * i], fw[2 * i + 1]});
    lat.RemoveEpsLocalLinear();  // faster-decoder.cc:422
  }
  return ok;
}

std::vector<int64_t> BatchFasterDecoder::Stats() const {
  std::vector<int64_t> s(static_cast<size_t>(std::max(batch_, 1)) * KD_NUM_STATS);
  KdCheck(kd_batch_stats(b_, s.data()));
  return s;
}

float BatchFasterDecoder::LastKernelMs() const {
  float ms = 0.f;
  KdCheck(kd_batch_last_kernel_ms(b_, &ms));
  return ms;
}

}  // namespace kaldi_decoder
