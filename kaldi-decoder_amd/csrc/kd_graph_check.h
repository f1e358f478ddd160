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
// Rule: inside a strongly connected component of the epsilon subgraph (an epsilon cycle), a
// negative arc is refused (conservative: its cycle might still be positive), and so is a cycle
// made of zero-weight arcs only.  HLG / HL / H graphs as icefall builds them have no epsilon
// cycles at all (the common case costs one in-degree peel over the epsilon arcs).
#ifndef KD_GRAPH_CHECK_H_
#define KD_GRAPH_CHECK_H_

#include <stdint.h>

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
  // 3. inside a component that holds a cycle: no negative arc, no cycle of zero-weight arcs
  std::vector<int32_t> zdeg(num_states, 0);
  bool any_zero = false;
  for (int32_t s = 0; s < num_states; ++s) {
    if (comp[s] < 0) continue;
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
      if (ilabel[a] != 0) continue;
      const int32_t d = nextstate[a];
      if (comp[d] != comp[s]) continue;
      if (comp_size[comp[s]] == 1 && d != s) continue;
      // (an arc inside a non-trivial component, or a self-loop: it lies on an epsilon cycle)
      if (weight[a] < 0.0f) {
        *why = "graph has an epsilon cycle through state " + std::to_string(s) +
               " with a negative-weight arc on it: the reference's ProcessNonemitting does not "
               "terminate on a negative epsilon cycle, and positive ones with negative arcs are "
               "not supported";
        return false;
      }
      if (weight[a] == 0.0f) {
        ++zdeg[d];
        any_zero = true;
      }
    }
  }
  if (!any_zero) return true;
  // a cycle among the zero-weight arcs that stay inside a component: peel again
  std::vector<int32_t> todo;
  int64_t in_comp = 0, peeled = 0;
  for (int32_t s = 0; s < num_states; ++s) {
    if (comp[s] < 0) continue;
    ++in_comp;
    if (zdeg[s] == 0) todo.push_back(s);
  }
  while (!todo.empty()) {
    const int32_t s = todo.back();
    todo.pop_back();
    ++peeled;
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
      if (ilabel[a] != 0 || weight[a] != 0.0f) continue;
      const int32_t d = nextstate[a];
      if (comp[d] != comp[s] || (comp_size[comp[s]] == 1 && d != s)) continue;
      if (--zdeg[d] == 0) todo.push_back(d);
    }
  }
  if (peeled == in_comp) return true;
  int32_t bad = -1;
  for (int32_t s = 0; s < num_states && bad < 0; ++s)
    if (comp[s] >= 0 && zdeg[s] > 0) bad = s;
  *why = "graph has an epsilon cycle of total weight zero (through or into state " + std::to_string(bad) +
         "): its tokens' back-pointers and lattice extra costs are not uniquely defined without the "
         "reference's container order; give the cycle a positive weight or remove it (fstrmepsilon)";
  return false;
}

}  // namespace kd

#endif  // KD_GRAPH_CHECK_H_
