// kaldi-decoder_amd/csrc/kd_path.hip -- ReachedFinal and GetBestPath on the
// device (ref: /root/reference/kaldi-decoder/csrc/faster-decoder.cc:347-424).
//
// One 256-thread workgroup per utterance.  Prepare: scan the last frame's
// arena block for the best token (cost, or cost + Final(state) when a final
// state is active), then follow the back-pointers once to get the path length.
// Fill: follow them again and write the arcs in path order into the caller's
// packed arrays.  Ties between equal totals go to the lowest state id (the
// reference takes the first in HashList order, which no parallel search can
// reproduce; see DESIGN.md "Order dependence").

#include "kd_device.h"

namespace kd {
namespace {

constexpr int kPathBlock = 256;
constexpr unsigned long long kKeyInf = 0xFFF0000000000000ull;

__global__ __launch_bounds__(kPathBlock) void PathPrepareKernel(const PathParams p) {
  __shared__ unsigned long long s_best;
  __shared__ unsigned int s_state;
  __shared__ int s_final;
  __shared__ long long s_tok;
  __shared__ unsigned long long s_bestc, s_bestwf;
  const int tid = threadIdx.x, utt = blockIdx.x;
  const int f = p.frames_decoded[utt];
  if (tid == 0) { s_best = ~0ull; s_state = 0xFFFFFFFFu; s_final = 0; s_tok = -1; s_bestc = ~0ull; s_bestwf = ~0ull; }
  __syncthreads();
  long long base = 0;
  int n = 0;
  if (f >= 0) {
    const long long fi = static_cast<long long>(utt) * (p.max_frames + 1) + f;
    base = p.frame_base[fi];
    n = p.frame_count[fi];
  }
  // ReachedFinal (:347-354); FinalRelativeCost's two minima (simple-decoder.cc:83-91)
  int any_final = 0;
  for (int i = tid; i < n; i += kPathBlock) {
    const double c = p.tok_cost[base + i];
    const float fc = p.states[p.tok_state[base + i]].final_cost;
    if (CostKey(c) != kKeyInf && !(fc == __int_as_float(0x7F800000))) any_final = 1;
    atomicMin(&s_bestc, CostKey(c));
    atomicMin(&s_bestwf, CostKey(c + fc));
  }
  if (any_final) s_final = 1;
  __syncthreads();
  const bool is_final = s_final != 0;
  // best token (:367-384)
  unsigned long long mk = ~0ull;
  for (int i = tid; i < n; i += kPathBlock) {
    double c = p.tok_cost[base + i];
    if (is_final) {
      c = c + p.states[p.tok_state[base + i]].final_cost;
      if (CostKey(c) == kKeyInf) continue;  // this_cost != infinity
    }
    const unsigned long long k = CostKey(c);
    mk = k < mk ? k : mk;
  }
  if (mk != ~0ull) atomicMin(&s_best, mk);
  __syncthreads();
  const unsigned long long best = s_best;
  if (best != ~0ull) {
    for (int i = tid; i < n; i += kPathBlock) {
      double c = p.tok_cost[base + i];
      if (is_final) c = c + p.states[p.tok_state[base + i]].final_cost;
      if (CostKey(c) == best) atomicMin(&s_state, static_cast<unsigned int>(p.tok_state[base + i]));
    }
  }
  __syncthreads();
  if (best != ~0ull) {
    for (int i = tid; i < n; i += kPathBlock)
      if (static_cast<unsigned int>(p.tok_state[base + i]) == s_state) s_tok = base + i;
  }
  __syncthreads();
  if (tid == 0) {
    const long long bt = s_tok;
    int len = -1;
    float fw = 0.f;
    if (bt >= 0) {
      len = 0;
      if (!p.no_backptr)
        for (long long t = bt; t >= 0 && p.tok_arc[t] != static_cast<int32_t>(kArcNone); t = p.tok_prev[t]) len++;
      if (is_final && p.use_final_probs) fw = p.states[p.tok_state[bt]].final_cost;  // :416-421
    }
    {
      float rel = __int_as_float(0x7F800000);  // no tokens: infinity (:82)
      if (n > 0) {
        rel = static_cast<float>(KeyCost(s_bestwf) - KeyCost(s_bestc));
        if (rel != rel) rel = __int_as_float(0x7F800000);  // NaN (:93-97)
      }
      p.final_rel[utt] = rel;
    }
    p.reached_final[utt] = is_final ? 1 : 0;
    p.best_tok[utt] = bt;
    p.path_len[utt] = len;
    p.final_weight[2 * utt] = fw;
    p.final_weight[2 * utt + 1] = 0.f;
  }
}

__global__ void PathFillKernel(const PathParams p) {
  const int utt = blockIdx.x * blockDim.x + threadIdx.x;
  if (utt >= p.batch) return;
  const int len = p.path_len[utt];
  if (len <= 0) return;
  const long long off = p.path_offset[utt];
  int pos = len - 1;
  for (long long t = p.best_tok[utt]; pos >= 0; --pos) {
    const int32_t a = p.tok_arc[t];
    const long long pr = p.tok_prev[t];
    const ArcRec arc = p.arcs[a];
    // :394-396: tot = float(cost - prev.cost); ac = tot - graph.  SimpleDecoder keeps
    // the arc's own (graph, acoustic) pair instead (simple-decoder.h:93-96).
    const float tot_cost = p.tok_cost[t] - (pr >= 0 ? p.tok_cost[pr] : 0.0);
    const float graph_cost = arc.weight;
    const float ac_cost = p.simple ? p.tok_ac[t] : tot_cost - graph_cost;
    p.o_ilabel[off + pos] = arc.ilabel;
    p.o_olabel[off + pos] = arc.olabel;
    p.o_state[off + pos] = static_cast<int32_t>(arc.next & kStateMask);
    p.o_graph[off + pos] = graph_cost;
    p.o_ac[off + pos] = ac_cost;
    t = pr;
  }
}

}  // namespace

hipError_t LaunchPathPrepare(const PathParams &p, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(PathPrepareKernel, dim3(batch), dim3(kPathBlock), 0, stream, p);
  return hipGetLastError();
}

hipError_t LaunchPathFill(const PathParams &p, int batch, hipStream_t stream) {
  // one lane per utterance; the walk is a dependent-load chain, so lanes of
  // different utterances overlap their latencies inside a wave
  hipLaunchKernelGGL(PathFillKernel, dim3((batch + 63) / 64), dim3(64), 0, stream, p);
  return hipGetLastError();
}

}  // namespace kd
