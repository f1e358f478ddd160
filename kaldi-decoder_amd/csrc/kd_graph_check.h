// kaldi-decoder_amd/csrc/kd_graph_check.h -- host-side check of a graph's epsilon arcs,
// run once by kd_graph_create.
//
// The reference decodes any epsilon topology: ProcessNonemitting re-queues a token on strict
// improvement (faster-decoder.cc:97-113, lattice-simple-decoder.cc:122-191), which ends on
// every cycle whose total weight is positive -- and on one of total weight ZERO only because
// an equal cost is no improvement -- and loops forever on a negative one.  This library
// reproduces the positive case (the search relaxes to the same fixpoint; the lattice sweep's
// extra costs have ONE fixpoint there, reached from above here and from below there) and
// REJECTS the rest at the boundary, because on an epsilon cycle of total weight zero
//   * the canonical tie rule (equal cost: lowest arc index wins) can make two tokens of the
//     cycle each other's back-pointer, where the reference's "first wins" cannot; and
//   * PruneForwardLinks' extra costs have more than one fixpoint: the reference iterates
//     upwards from the zeros new tokens are created with (lattice-simple-decoder.cc:246-304)
//     and leaves the cycle's tokens at extra cost 0 whatever their exits cost, the sweep here
//     relaxes downwards from +infinity and finds the true distance -- another lattice.
// Rule (round 5): every epsilon cycle must have a POSITIVE total weight.  Negative arcs inside a
// positive cycle are fine (weight-pushed and LM-scaled graphs carry them): per strongly
// connected component of the epsilon subgraph the minimum cycle MEAN is computed (Karp, in
// double; components are a handful of states) and the graph is refused only if it is <= 0 --
// a cycle of total weight zero or less exists -- or not a number (a NaN weight on a cycle: the
// closure's strict-improvement argument says nothing about it).  Until round 5 any negative arc
// inside a component was refused.  HLG / HL / H graphs as icefall builds them have no epsilon
// cycles at all (the common case costs one in-degree peel over the epsilon arcs).
#ifndef KD_GRAPH_CHECK_H_
#define KD_GRAPH_CHECK_H_

#include <stdint.h>

#include <limits>
#include <string>
#include <vector>

namespace kd {

// Returns true if the graph passes; otherwise `why` names a state on the offending cycle.
inline bool EpsilonCyclesArePositive(int32_t num_states, const int32_t *row_ptr, const int32_t *ilabel,
                                     const float *weight, const int32_t *nextstate, std::string *why) {
  // 1. peel states whose epsilon in-degree is zero (Kahn): what is left lies on an epsilon
  //    cycle or downstream of one
  std::vector<int32_t> indeg(num_states, 0);
  int64_t n_eps = 0;
  for (int32_t s = 0; s < num_states; ++s)
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a)
      if (ilabel[a] == 0) { ++indeg[nextstate[a]]; ++n_eps; }
  if (n_eps == 0) return true;
  std::vector<int32_t> stack;
  std::vector<uint8_t> left(num_states, 1);
  for (int32_t s = 0; s < num_states; ++s)
    if (indeg[s] == 0) stack.push_back(s);
  int64_t removed = 0;
  while (!stack.empty()) {
    const int32_t s = stack.back();
    stack.pop_back();
    left[s] = 0;
    ++removed;
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a)
      if (ilabel[a] == 0 && --indeg[nextstate[a]] == 0) stack.push_back(nextstate[a]);
  }
  if (removed == num_states) return true;  // acyclic
  // 2. strongly connected components of what is left (Tarjan, iterative)
  std::vector<int32_t> index(num_states, -1), low(num_states, 0), comp(num_states, -1);
  std::vector<uint8_t> on_stack(num_states, 0);
  std::vector<int32_t> tstack;
  struct Frame { int32_t s, a; };
  std::vector<Frame> call;
  int32_t next_index = 0, n_comp = 0;
  std::vector<int32_t> comp_size;
  for (int32_t root = 0; root < num_states; ++root) {
    if (!left[root] || index[root] >= 0) continue;
    call.push_back({root, row_ptr[root]});
    index[root] = low[root] = next_index++;
    tstack.push_back(root);
    on_stack[root] = 1;
    while (!call.empty()) {
      Frame &f = call.back();
      const int32_t s = f.s;
      bool descended = false;
      while (f.a < row_ptr[s + 1]) {
        const int32_t a = f.a++;
        if (ilabel[a] != 0) continue;
        const int32_t d = nextstate[a];
        if (!left[d]) continue;
        if (index[d] < 0) {
          index[d] = low[d] = next_index++;
          tstack.push_back(d);
          on_stack[d] = 1;
          call.push_back({d, row_ptr[d]});
          descended = true;
          break;
        }
        if (on_stack[d] && index[d] < low[s]) low[s] = index[d];
      }
      if (descended) continue;
      if (low[s] == index[s]) {
        int32_t n = 0;
        while (true) {
          const int32_t v = tstack.back();
          tstack.pop_back();
          on_stack[v] = 0;
          comp[v] = n_comp;
          ++n;
          if (v == s) break;
        }
        comp_size.push_back(n);
        ++n_comp;
      }
      call.pop_back();
      if (!call.empty()) {
        const int32_t parent = call.back().s;
        if (low[s] < low[parent]) low[parent] = low[s];
      }
    }
  }
  // 3. per component that holds a cycle: its minimum cycle mean must be positive
  std::vector<std::vector<int32_t>> members(n_comp);
  for (int32_t s = 0; s < num_states; ++s)
    if (comp[s] >= 0) members[comp[s]].push_back(s);
  std::vector<int32_t> local(num_states, -1);
  for (int32_t c = 0; c < n_comp; ++c) {
    const std::vector<int32_t> &ms = members[c];
    const int32_t n = static_cast<int32_t>(ms.size());
    struct Edge { int32_t u, v; double w; };
    std::vector<Edge> edges;
    for (int32_t i = 0; i < n; ++i) local[ms[i]] = i;
    bool nan = false;
    for (int32_t i = 0; i < n; ++i) {
      const int32_t s = ms[i];
      for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
        if (ilabel[a] != 0 || comp[nextstate[a]] != c) continue;
        if (weight[a] != weight[a]) nan = true;
        edges.push_back({i, local[nextstate[a]], static_cast<double>(weight[a])});
      }
    }
    if (n == 1 && edges.empty()) continue;  // (a single state without a self-loop: no cycle)
    if (nan) {
      *why = "graph has an epsilon cycle through state " + std::to_string(ms[0]) +
             " with a NaN weight on it";
      return false;
    }
    double mu;  // the component's minimum cycle mean
    if (static_cast<int64_t>(n) * n > (int64_t{1} << 22)) {  // (32 MB of doubles at most; components are a handful of states)
      // A component too large for Karp's table (thousands of states on one epsilon cycle
      // structure): the conservative rule of rounds 4 -- a negative arc refuses, and so does a
      // cycle made of zero-weight arcs only (peel the zero-weight arcs' in-degrees).
      mu = 1.0;
      std::vector<int32_t> zdeg(n, 0), todo;
      for (const Edge &e : edges) {
        if (e.w < 0.0) mu = -1.0;
        if (e.w == 0.0) ++zdeg[e.v];
      }
      if (mu > 0.0) {
        std::vector<std::vector<int32_t>> zout(n);
        for (const Edge &e : edges)
          if (e.w == 0.0) zout[e.u].push_back(e.v);
        int32_t peeled = 0;
        for (int32_t i = 0; i < n; ++i)
          if (zdeg[i] == 0) todo.push_back(i);
        while (!todo.empty()) {
          const int32_t u = todo.back();
          todo.pop_back();
          ++peeled;
          for (int32_t v : zout[u])
            if (--zdeg[v] == 0) todo.push_back(v);
        }
        if (peeled != n) mu = 0.0;
      }
    } else {
      // Karp: D[k][v] = the lightest walk of exactly k arcs from member 0 to v;
      // mu = min over v of max over k < n of (D[n][v] - D[k][v]) / (n - k)
      const double inf = std::numeric_limits<double>::infinity();
      std::vector<double> D(static_cast<size_t>(n + 1) * n, inf);
      D[0] = 0.0;
      for (int32_t k = 1; k <= n; ++k)
        for (const Edge &e : edges) {
          const double via = D[static_cast<size_t>(k - 1) * n + e.u] + e.w;
          double &d = D[static_cast<size_t>(k) * n + e.v];
          if (via < d) d = via;
        }
      mu = inf;
      for (int32_t v = 0; v < n; ++v) {
        const double dn = D[static_cast<size_t>(n) * n + v];
        if (dn == inf) continue;
        double worst = -inf;
        for (int32_t k = 0; k < n; ++k) {
          const double dk = D[static_cast<size_t>(k) * n + v];
          if (dk == inf) continue;
          const double m = (dn - dk) / (n - k);
          if (m > worst) worst = m;
        }
        if (worst < mu) mu = worst;
      }
    }
    if (!(mu > 0.0)) {
      *why = std::string("graph has an epsilon cycle of total weight ") + (mu < 0.0 ? "below zero" : "zero") +
             " (through or into state " + std::to_string(ms[0]) + "): " +
             (mu < 0.0 ? "the reference's ProcessNonemitting does not terminate on a negative epsilon cycle"
                       : "its tokens' back-pointers and lattice extra costs are not uniquely defined without "
                         "the reference's container order") +
             "; give the cycle a positive weight or remove it (fstrmepsilon)";
      return false;
    }
  }
  return true;
}

}  // namespace kd

#endif  // KD_GRAPH_CHECK_H_
