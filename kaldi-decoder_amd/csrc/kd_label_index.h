// kaldi-decoder_amd/csrc/kd_label_index.h -- the label index of a state's emitting arcs.
//
// The reference's emitting loop visits every arc of every token it expands
// (/root/reference/kaldi-decoder/csrc/faster-decoder.cc:204-211,
// lattice-simple-decoder.cc:375-399).  With peaky CTC posteriors 94 % of those arcs are known
// to be beyond the cutoff before they are read: all that a token which is not among the
// frame's best can still afford is the frame's likeliest label (kd_search.hip, the tier-1
// frames' second pass).  The index answers "which arcs of this state carry ilabel L" without
// reading the arcs.
//
// Layout: SearchParams::lab_idx is an array parallel to the arcs; the entries of a state's
// emitting range [arc_begin, emit_end) are the keys (ilabel << kIdxOffBits | offset of the arc
// in the range << 1 | "the next key in ascending order has the same ilabel"), one per arc,
// arranged as an implicit 5-ary search tree in breadth-first order:
// node j holds the keys at positions 4j .. 4j + 3 (16 bytes: one load), its children are the
// nodes 5j + 1 .. 5j + 5, and an in-order walk yields the keys ascending.  A lookup reads one
// node per level -- 1 level up to 4 arcs, 2 up to 24, 3 up to 124, 4 up to 624 -- against the
// 3 / 5 / 7 / 10 dependent probes of a binary search over the sorted keys (measured, round 5:
// the binary search's probes cost a tier-1 frame of 512-1023 tokens 9.8 k cycles -- the probes
// of one round miss the 32 KB L1 again by the time a lane comes back to its line).  The
// arcs themselves keep their order: arc indices break ties and leave the library in lattices.
// States of 2048 emitting arcs or more keep no index (their entries are all ones).
#ifndef KD_LABEL_INDEX_H_
#define KD_LABEL_INDEX_H_

#include <stdint.h>
#include <algorithm>

#include "kd_device.h"

namespace kd {

constexpr uint32_t kIdxNone = 0xFFFFFFFFu;
constexpr uint32_t kIdxOffMask = (1u << kIdxOffBits) - 1u;
constexpr int32_t kIdxMaxDeg = 1 << (kIdxOffBits - 1);  // ranges of this many arcs or more: no index
__host__ __device__ inline uint32_t IdxKey(uint32_t ilabel, uint32_t offset) {
  return ilabel << kIdxOffBits | offset << 1;
}
__host__ __device__ inline uint32_t IdxKeyLabel(uint32_t key) { return key >> kIdxOffBits; }
__host__ __device__ inline uint32_t IdxKeyOffset(uint32_t key) { return (key & kIdxOffMask) >> 1; }
__host__ __device__ inline bool IdxKeyMore(uint32_t key) { return key & 1u; }

// `sorted`: the n keys ascending; `out`: the n entries in tree order (an in-order walk of the
// implicit tree hands out the keys; a node whose last slots do not exist is a leaf).
inline void LabelIndexFill(const uint32_t *sorted, int32_t n, uint32_t *out, int32_t node, int32_t &next) {
  for (int32_t k = 0; k < 4; ++k) {
    if (4 * (5 * node + 1 + k) < n) LabelIndexFill(sorted, n, out, 5 * node + 1 + k, next);
    if (4 * node + k < n) out[4 * node + k] = sorted[next++];
  }
  if (4 * (5 * node + 5) < n) LabelIndexFill(sorted, n, out, 5 * node + 5, next);
}
inline void LabelIndexLayout(const uint32_t *sorted, int32_t n, uint32_t *out) {
  int32_t next = 0;
  if (n > 0) LabelIndexFill(sorted, n, out, 0, next);
}
// The index of a range of n < kIdxMaxDeg arcs with the given input labels, into out[0, n).
inline void LabelIndexBuild(const uint32_t *ilabel, int32_t n, uint32_t *out, uint32_t *scratch /* n */) {
  for (int32_t i = 0; i < n; ++i) scratch[i] = IdxKey(ilabel[i], static_cast<uint32_t>(i));
  // (keys carry their offset: equal labels keep their arc order.  A range that arcsort left
  // ascending -- the usual case -- is found so in one pass; insertion sort for the short ranges
  // that make up a graph, std::sort for the history states of a large LM: up to 2047 arcs)
  bool ascending = true;
  for (int32_t i = 1; i < n && ascending; ++i) ascending = scratch[i - 1] <= scratch[i];
  if (!ascending) {
    if (n > 32) {
      std::sort(scratch, scratch + n);
    } else {
      for (int32_t i = 1; i < n; ++i) {
        const uint32_t v = scratch[i];
        int32_t j = i;
        for (; j > 0 && scratch[j - 1] > v; --j) scratch[j] = scratch[j - 1];
        scratch[j] = v;
      }
    }
  }
  for (int32_t i = 0; i + 1 < n; ++i)
    if (IdxKeyLabel(scratch[i]) == IdxKeyLabel(scratch[i + 1])) scratch[i] |= 1u;
  LabelIndexLayout(scratch, n, out);
}

// The smallest key >= target among the `deg` keys of a range in tree order, or kIdxNone.
// `node(j)` returns the four entries at positions 4j .. 4j + 3 (those at or beyond `deg` are
// ignored here); `probes` counts the nodes read.
struct IdxNode { uint32_t k[4]; };
template <class LoadNode>
__host__ __device__ inline uint32_t LabelIndexLowerBound(LoadNode node, int32_t deg, uint32_t target,
                                                         int32_t &probes) {
  uint32_t cand = kIdxNone;
  int32_t j = 0;
  while (4 * j < deg) {
    const IdxNode nd = node(j);
    ++probes;
    const int32_t have = deg - 4 * j;  // keys of this node that exist (>= 1)
    const uint32_t k0 = nd.k[0], k1 = have > 1 ? nd.k[1] : kIdxNone, k2 = have > 2 ? nd.k[2] : kIdxNone,
                   k3 = have > 3 ? nd.k[3] : kIdxNone;
    const int32_t c = (k0 < target) + (k1 < target) + (k2 < target) + (k3 < target);
    // (ascending inside a node: key c is the node's smallest key >= target, and every key of
    // the subtree we descend into is below whatever `cand` held before)
    const uint32_t kc = c == 0 ? k0 : c == 1 ? k1 : c == 2 ? k2 : c == 3 ? k3 : kIdxNone;
    if (kc != kIdxNone) cand = kc;
    j = 5 * j + 1 + c;
  }
  return cand;
}

}  // namespace kd

#endif  // KD_LABEL_INDEX_H_
