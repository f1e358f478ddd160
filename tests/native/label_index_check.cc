// tests/native/label_index_check.cc -- host check of csrc/kd_label_index.h (compiled and run by
// tests/test_label_index.py): for every range size up to 700 and random label multisets, the
// implicit 5-ary tree's lower bound equals std::lower_bound over the sorted keys, repeated
// lower bounds enumerate exactly the arcs of a label, and a lookup reads one node per level.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define __host__
#define __device__
#include "kd_label_index.h"
using namespace kd;
int main() {
  srand(1);
  for (int n = 1; n < 700; ++n) for (int rep = 0; rep < 20; ++rep) {
    std::vector<uint32_t> keys(n), labs(n), tmp(n);
    int V = 1 + rand() % 50;
    for (int i = 0; i < n; ++i) labs[i] = (uint32_t)(1 + rand() % V);
    for (int i = 0; i < n; ++i) keys[i] = IdxKey(labs[i], i);
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i + 1 < n; ++i) if (IdxKeyLabel(keys[i]) == IdxKeyLabel(keys[i + 1])) keys[i] |= 1u;
    std::vector<uint32_t> tree(n + 4, 0x12345678u);
    LabelIndexBuild(labs.data(), n, tree.data(), tmp.data());
    for (int t = 0; t <= V + 1; ++t) {
      uint32_t target = (uint32_t)t << kIdxOffBits;
      // all matches by repeated lower bounds
      std::vector<uint32_t> got, want;
      for (auto k : keys) if ((k >> kIdxOffBits) == (uint32_t)t) want.push_back(k);
      int probes = 0;
      uint32_t c = LabelIndexLowerBound([&](int j) { IdxNode nd; for (int q = 0; q < 4; ++q) nd.k[q] = tree[4 * j + q]; return nd; }, n, target, probes);
      while (c != kIdxNone && (c >> kIdxOffBits) == (uint32_t)t) {
        got.push_back(c);
        if (!IdxKeyMore(c)) break;  // (the key says whether another arc of the label follows)
        c = LabelIndexLowerBound([&](int j) { IdxNode nd; for (int q = 0; q < 4; ++q) nd.k[q] = tree[4 * j + q]; return nd; }, n, c + 1, probes);
      }
      if (got != want) { printf("MISMATCH n=%d t=%d got %zu want %zu\n", n, t, got.size(), want.size()); return 1; }
      auto lb = std::lower_bound(keys.begin(), keys.end(), target);
      int p2 = 0;
      uint32_t c2 = LabelIndexLowerBound([&](int j) { IdxNode nd; for (int q = 0; q < 4; ++q) nd.k[q] = tree[4 * j + q]; return nd; }, n, target, p2);
      if ((lb == keys.end() ? kIdxNone : *lb) != c2) { printf("LB MISMATCH n=%d t=%d\n", n, t); return 1; }
      int lv = n <= 4 ? 1 : n <= 24 ? 2 : n <= 124 ? 3 : n <= 624 ? 4 : 5;
      if (p2 > lv) { printf("too many probes n=%d %d\n", n, p2); return 1; }
    }
  }
  puts("ok");
}
