// kaldi-decoder_amd/csrc/fst.h -- the decoding graph and the lattice result type.
//
// The reference borrows an fst::Fst<fst::StdArc> from kaldifst/OpenFst and
// returns fst::VectorFst<fst::LatticeArc> (/root/reference/kaldi-decoder/csrc/
// faster-decoder.h:71-95); neither library is part of this build.  `Fst` is the
// read-only StdArc graph as CSR arrays plus its per-device GPU copy; `Lattice`
// is a small mutable LatticeArc FST holding what GetBestPath / GetRawLattice
// produce.
#ifndef KALDI_DECODER_AMD_CSRC_FST_H_
#define KALDI_DECODER_AMD_CSRC_FST_H_

#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kd_hip.h"

namespace kaldi_decoder {

inline void KdCheck(int rc) {
  if (rc != KD_OK) throw std::runtime_error(kd_last_error());
}

// A tropical-semiring transducer (StdArc): arcs of state s are
// [row_ptr[s], row_ptr[s+1]); ilabel 0 is epsilon; final cost +inf = not final.
class Fst {
 public:
  Fst(int32_t start, std::vector<int32_t> row_ptr, std::vector<int32_t> ilabel,
      std::vector<int32_t> olabel, std::vector<float> weight, std::vector<int32_t> nextstate,
      std::vector<float> final_cost)
      : start_(start), row_ptr_(std::move(row_ptr)), ilabel_(std::move(ilabel)),
        olabel_(std::move(olabel)), weight_(std::move(weight)),
        nextstate_(std::move(nextstate)), final_(std::move(final_cost)) {
    if (row_ptr_.size() != final_.size() + 1 || row_ptr_.empty())
      throw std::runtime_error("Fst: row_ptr must have num_states + 1 entries");
    const size_t na = static_cast<size_t>(row_ptr_.back());
    if (ilabel_.size() != na || olabel_.size() != na || weight_.size() != na || nextstate_.size() != na)
      throw std::runtime_error("Fst: arc arrays must have row_ptr[num_states] entries");
  }
  ~Fst() {
    for (auto &kv : device_) kd_graph_destroy(kv.second);
  }
  Fst(const Fst &) = delete;
  Fst &operator=(const Fst &) = delete;

  int32_t Start() const { return start_; }
  int32_t NumStates() const { return static_cast<int32_t>(final_.size()); }
  int64_t NumArcs() const { return row_ptr_.back(); }
  float Final(int32_t s) const { return final_[s]; }
  const std::vector<int32_t> &RowPtr() const { return row_ptr_; }
  const std::vector<int32_t> &Ilabel() const { return ilabel_; }
  const std::vector<int32_t> &Olabel() const { return olabel_; }
  const std::vector<float> &Weight() const { return weight_; }
  const std::vector<int32_t> &NextState() const { return nextstate_; }
  const std::vector<float> &FinalCosts() const { return final_; }

  // The graph's copy on `device` (uploaded on first use, shared by every decoder).
  const kd_graph *OnDevice(int device) const {
    std::lock_guard<std::mutex> lock(mu_);
    auto it = device_.find(device);
    if (it != device_.end()) return it->second;
    kd_graph *g = nullptr;
    KdCheck(kd_graph_create(device, NumStates(), start_, row_ptr_.data(), ilabel_.data(),
                            olabel_.data(), weight_.data(), nextstate_.data(), final_.data(), &g));
    device_[device] = g;
    return g;
  }

 private:
  int32_t start_;
  std::vector<int32_t> row_ptr_, ilabel_, olabel_;
  std::vector<float> weight_;
  std::vector<int32_t> nextstate_;
  std::vector<float> final_;
  mutable std::mutex mu_;
  mutable std::map<int, kd_graph *> device_;
};

// LatticeWeight = (graph cost, acoustic cost); One() = (0, 0), Zero() = (+inf, +inf).
struct LatticeWeight {
  float graph = 0.f, acoustic = 0.f;
  static LatticeWeight One() { return {0.f, 0.f}; }
  static LatticeWeight Zero() {
    const float inf = std::numeric_limits<float>::infinity();
    return {inf, inf};
  }
  bool IsZero() const {
    return graph == std::numeric_limits<float>::infinity() &&
           acoustic == std::numeric_limits<float>::infinity();
  }
};

struct LatticeArc {
  int32_t ilabel, olabel;
  LatticeWeight weight;
  int32_t nextstate;
};

// Mutable LatticeArc FST with the handful of operations the decoders and
// their callers use (AddState / SetStart / AddArc / SetFinal / iteration).
class Lattice {
 public:
  void DeleteStates() { arcs_.clear(); final_.clear(); start_ = -1; }
  int32_t AddState() {
    arcs_.emplace_back();
    final_.push_back(LatticeWeight::Zero());
    return static_cast<int32_t>(arcs_.size()) - 1;
  }
  void SetStart(int32_t s) { start_ = s; }
  void AddArc(int32_t s, const LatticeArc &a) { arcs_[s].push_back(a); }
  void SetFinal(int32_t s, LatticeWeight w) { final_[s] = w; }
  int32_t Start() const { return start_; }
  int32_t NumStates() const { return static_cast<int32_t>(arcs_.size()); }
  int64_t NumArcs() const {
    int64_t n = 0;
    for (auto &v : arcs_) n += static_cast<int64_t>(v.size());
    return n;
  }
  LatticeWeight Final(int32_t s) const { return final_[s]; }
  const std::vector<LatticeArc> &Arcs(int32_t s) const { return arcs_[s]; }
  std::vector<LatticeArc> &MutableArcs(int32_t s) { return arcs_[s]; }

  // kaldifst's GetLinearSymbolSequence: for a linear FST, the non-epsilon input
  // and output labels and the total weight (arcs x final).  False if not linear.
  bool GetLinearSymbolSequence(std::vector<int32_t> *isyms, std::vector<int32_t> *osyms,
                               LatticeWeight *total) const;

  // The linear-chain case of kaldifst's RemoveEpsLocal (called at
  // /root/reference/kaldi-decoder/csrc/faster-decoder.cc:422): an arc with
  // epsilon on both sides is folded into the arc that follows it (or, for a
  // trailing one, into the final weight).  Label sequences and the total weight
  // are unchanged.  kaldifst is not in the reference tree: parity unpinned.
  void RemoveEpsLocalLinear();

 private:
  int32_t start_ = -1;
  std::vector<std::vector<LatticeArc>> arcs_;
  std::vector<LatticeWeight> final_;
};

inline bool Lattice::GetLinearSymbolSequence(std::vector<int32_t> *isyms,
                                             std::vector<int32_t> *osyms,
                                             LatticeWeight *total) const {
  if (isyms) isyms->clear();
  if (osyms) osyms->clear();
  LatticeWeight tot = LatticeWeight::One();
  if (start_ < 0) {
    if (total) *total = LatticeWeight::Zero();
    return true;  // the empty FST counts as linear
  }
  int32_t s = start_;
  while (true) {
    const LatticeWeight f = final_[s];
    if (!f.IsZero()) {
      if (!arcs_[s].empty()) return false;
      tot.graph += f.graph;
      tot.acoustic += f.acoustic;
      if (total) *total = tot;
      return true;
    }
    if (arcs_[s].size() != 1) return false;
    const LatticeArc &a = arcs_[s][0];
    tot.graph += a.weight.graph;
    tot.acoustic += a.weight.acoustic;
    if (a.ilabel != 0 && isyms) isyms->push_back(a.ilabel);
    if (a.olabel != 0 && osyms) osyms->push_back(a.olabel);
    s = a.nextstate;
  }
}

inline void Lattice::RemoveEpsLocalLinear() {
  if (start_ < 0) return;
  std::vector<LatticeArc> path;
  int32_t s = start_;
  while (final_[s].IsZero() && arcs_[s].size() == 1) {
    path.push_back(arcs_[s][0]);
    s = arcs_[s][0].nextstate;
  }
  LatticeWeight fin = final_[s];
  std::vector<LatticeArc> kept;
  LatticeWeight carry = LatticeWeight::One();
  for (const LatticeArc &a : path) {
    if (a.ilabel == 0 && a.olabel == 0) {
      carry.graph += a.weight.graph;
      carry.acoustic += a.weight.acoustic;
    } else {
      LatticeArc b = a;
      b.weight.graph += carry.graph;
      b.weight.acoustic += carry.acoustic;
      carry = LatticeWeight::One();
      kept.push_back(b);
    }
  }
  fin.graph += carry.graph;
  fin.acoustic += carry.acoustic;
  DeleteStates();
  int32_t cur = AddState();
  SetStart(cur);
  for (LatticeArc a : kept) {
    a.nextstate = AddState();
    AddArc(cur, a);
    cur = a.nextstate;
  }
  SetFinal(cur, fin);
}

}  // namespace kaldi_decoder

#endif  // KALDI_DECODER_AMD_CSRC_FST_H_
