// Test driver (CPU, built with -fsanitize=address,undefined by tests/test_graph_check.py):
// reads graphs from stdin -- int32 num_states, int32 num_arcs, row_ptr[num_states + 1], ilabel[],
// nextstate[], float weight[] -- runs kd_graph_create's epsilon-cycle check on each and prints
// one line per graph: "1" (accepted) or "0 <message>".
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "kd_graph_check.h"

int main() {
  int32_t hdr[2];
  while (fread(hdr, sizeof(int32_t), 2, stdin) == 2) {
    const int32_t S = hdr[0], A = hdr[1];
    std::vector<int32_t> row_ptr(S + 1), ilabel(A), next(A);
    std::vector<float> weight(A);
    if (fread(row_ptr.data(), 4, S + 1, stdin) != static_cast<size_t>(S + 1)) return 2;
    if (A && fread(ilabel.data(), 4, A, stdin) != static_cast<size_t>(A)) return 2;
    if (A && fread(next.data(), 4, A, stdin) != static_cast<size_t>(A)) return 2;
    if (A && fread(weight.data(), 4, A, stdin) != static_cast<size_t>(A)) return 2;
    std::string why;
    const bool ok = kd::EpsilonCyclesArePositive(S, row_ptr.data(), ilabel.data(), weight.data(), next.data(), &why);
    printf("%d %s\n", ok ? 1 : 0, why.c_str());
  }
  return 0;
}
