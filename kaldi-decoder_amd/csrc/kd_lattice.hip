// kaldi-decoder_amd/csrc/kd_lattice.hip -- lattice pruning and raw-lattice export
// on the device (ref: /root/reference/kaldi-decoder/csrc/lattice-simple-decoder.cc).
//
//   LatticePruneKernel  ComputeFinalCosts (:522-560), PruneForwardLinksFinal
//                       (:425-520) and the backward sweep of FinalizeDecoding
//                       (:407-420) = PruneForwardLinks with delta 0 (:228-305) +
//                       PruneTokensForFrame (:310-334) for every frame
//   LatticeCountKernel  sizes of GetRawLattice's output (:584-657) and the state
//   LatticeFillKernel   numbering, then the arcs / final weights themselves
//
// One 1024-thread workgroup per utterance walks its frames backwards.  A frame's
// links are processed link-parallel: link_extra = next.extra + ((tok.tot + ac +
// graph) - next.tot), links above lattice_beam are excised (dst = -1), the
// survivors min-reduce into their source token's extra cost with ds_min_u32 on an
// order-preserving float key held in LDS (one word per token of the frame).
// Epsilon links point inside the frame, so the frame is iterated to its fixpoint
// (from below: extras start at 0 and only grow, which makes early excision
// safe).  Forward links were recorded unpruned during the search -- 288 GB of HBM
// make the reference's periodic PruneActiveTokens (:198-223) unnecessary, and it
// never removes anything the final pass keeps.

#include "kd_device.h"

namespace kd {
namespace {

constexpr int kLBlock = 1024;
constexpr int kLWaves = kLBlock / 64;
constexpr int kLdsToks = 36864;  // extra-cost keys of one frame kept in LDS (144 KiB)
constexpr uint32_t kFInfKey = 0xFF800000u;  // FloatKey(+inf)

#define KD_G __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ KD_G T *ToG(T *p) { return (KD_G T *)p; }
template <class T>
__device__ __forceinline__ T *ToFlat(KD_G T *p) { return (T *)p; }

struct LShared {
  uint32_t ne[kLdsToks];
  int32_t red_i[kLWaves];
  uint32_t red_u[kLWaves];
  uint32_t best_key, bestwf_key;
  int32_t any_final, changed, count, start_id, empty_frame;
  int32_t n_states, n_arcs, n_finals;
};

__device__ __forceinline__ float Inf() { return __int_as_float(0x7F800000); }

__device__ __forceinline__ int BlockExclScanI(LShared &s, int v, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s.red_i[wave] = x;
  __syncthreads();
  if (wave == 0) {
    int w = lane < kLWaves ? s.red_i[lane] : 0, xs = w;
#pragma unroll
    for (int d = 1; d < kLWaves; d <<= 1) {
      int y = __shfl_up(xs, d, 64);
      if (lane >= d) xs += y;
    }
    if (lane < kLWaves) s.red_i[lane] = xs - w;
    if (lane == kLWaves - 1) s.count = xs;
  }
  __syncthreads();
  const int off = s.red_i[wave] + x - v;
  total = s.count;
  __syncthreads();
  return off;
}

// Applies one link block to the frame's new-extra keys.
__device__ __forceinline__ void ApplyLinks(const LatticeParams &p, LShared &s, KD_G uint32_t *ne,
                                           bool ne_in_lds, long long lb, int ln, long long base) {
  for (int i = threadIdx.x; i < ln; i += kLBlock) {
    const long long at = lb + i;
    const int32_t dst = ToG(p.link_dst)[at];
    if (dst < 0) continue;  // already excised
    const int32_t src = ToG(p.link_src)[at];
    const float tok_tot = static_cast<float>(ToG(p.tok_cost)[src]);
    const float next_tot = static_cast<float>(ToG(p.tok_cost)[dst]);
    const float next_extra = ToG(p.tok_extra)[dst];
    // lattice-simple-decoder.cc:257-260
    float link_extra_cost =
        next_extra + ((tok_tot + ToG(p.link_ac)[at] + ToG(p.link_graph)[at]) - next_tot);
    if (link_extra_cost > p.lattice_beam) {  // :264 excise
      ToG(p.link_dst)[at] = -1;
      continue;
    }
    if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
    const uint32_t k = FloatKey(link_extra_cost);
    const long long li = src - base;
    if (ne_in_lds) atomicMin(&s.ne[li], k);
    else atomicMin(ToFlat(&ne[li]), k);
  }
}

__global__ __launch_bounds__(kLBlock) void LatticePruneKernel(const LatticeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LShared &s = *reinterpret_cast<LShared *>(smem);
  const int tid = threadIdx.x, utt = blockIdx.x;
  const int F = ToG(p.frames_decoded)[utt];
  if (F < 0) return;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  KD_G uint32_t *gne = ToG(p.scratch) + utt * p.scratch_stride;

  // ComputeFinalCosts (:522-560) over the last frame's tokens
  const long long baseF = ToG(p.frame_base)[fb + F];
  const int nF = ToG(p.frame_count)[fb + F];
  if (tid == 0) { s.best_key = kFInfKey; s.bestwf_key = kFInfKey; s.any_final = 0; }
  __syncthreads();
  {
    uint32_t bk = kFInfKey, bwk = kFInfKey;
    int af = 0;
    for (int i = tid; i < nF; i += kLBlock) {
      const float cost = static_cast<float>(ToG(p.tok_cost)[baseF + i]);
      const float final_cost = ToG(p.states)[ToG(p.tok_state)[baseF + i]].final_cost;
      const float cost_with_final = cost + final_cost;
      const uint32_t k1 = FloatKey(cost), k2 = FloatKey(cost_with_final);
      bk = k1 < bk ? k1 : bk;
      bwk = k2 < bwk ? k2 : bwk;
      af |= final_cost != Inf();
    }
    if (bk != kFInfKey) atomicMin(&s.best_key, bk);
    if (bwk != kFInfKey) atomicMin(&s.bestwf_key, bwk);
    if (af) s.any_final = 1;
  }
  __syncthreads();
  const float best_cost = KeyFloat(s.best_key), best_wf = KeyFloat(s.bestwf_key);
  const bool any_final = s.any_final != 0;
  const float final_best_cost = best_wf != Inf() ? best_wf : best_cost;  // :553-559
  if (tid == 0) {
    ToG(p.final_relative_cost)[utt] =
        (best_cost == Inf() && best_wf == Inf()) ? Inf() : best_wf - best_cost;  // :544-552
    ToG(p.final_best_cost)[utt] = final_best_cost;
    ToG(p.any_final)[utt] = any_final ? 1 : 0;
  }
  if (p.stats_only) return;

  for (int f = F; f >= 0; --f) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    const bool last = f == F;
    const bool ne_in_lds = n <= kLdsToks;
    const long long eb = last ? 0 : ToG(p.link_emit_base)[fb + f];
    const int en = last ? 0 : ToG(p.link_emit_count)[fb + f];
    const long long pb = ToG(p.link_eps_base)[fb + f];
    const int pn = ToG(p.link_eps_count)[fb + f];
    while (true) {
      // start value: +inf, or for the last frame the token's own final term (:467)
      for (int i = tid; i < n; i += kLBlock) {
        uint32_t k = kFInfKey;
        if (last) {
          const float tot = static_cast<float>(ToG(p.tok_cost)[base + i]);
          const float fc =
              any_final ? ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost : 0.0f;
          k = FloatKey(tot + fc - final_best_cost);
        }
        if (ne_in_lds) s.ne[i] = k; else gne[i] = k;
      }
      if (tid == 0) s.changed = 0;
      __syncthreads();
      ApplyLinks(p, s, gne, ne_in_lds, eb, en, base);
      ApplyLinks(p, s, gne, ne_in_lds, pb, pn, base);
      __syncthreads();
      int ch = 0;
      for (int i = tid; i < n; i += kLBlock) {
        float v = KeyFloat(ne_in_lds ? s.ne[i]
                                     : __hip_atomic_load(ToFlat(&gne[i]), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT));
        if (last && v > p.lattice_beam) v = Inf();  // :507-509
        if (v != ToG(p.tok_extra)[base + i]) {
          ToG(p.tok_extra)[base + i] = v;
          ch = 1;
        }
      }
      if (ch) s.changed = 1;
      __syncthreads();
      const int changed = s.changed;
      __syncthreads();
      // without epsilon links nothing in this frame depends on this frame
      if (!changed || pn == 0) break;
    }
  }
}

__device__ __forceinline__ bool Alive(const LatticeParams &p, long long g) {
  return !p.finalized || ToG(p.tok_extra)[g] != Inf();
}

__global__ __launch_bounds__(kLBlock) void LatticeCountKernel(const LatticeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LShared &s = *reinterpret_cast<LShared *>(smem);
  const int tid = threadIdx.x, utt = blockIdx.x;
  const int F = ToG(p.frames_decoded)[utt];
  KD_G int32_t *sz = ToG(p.sizes) + 4 * utt;
  if (tid == 0) { s.n_states = 0; s.n_arcs = 0; s.n_finals = 0; s.start_id = -1; s.empty_frame = 0; }
  __syncthreads();
  if (F < 1) {  // KALDI_DECODER_ASSERT(num_frames > 0), :610
    if (tid == 0) { sz[0] = 0; sz[1] = 0; sz[2] = 0; sz[3] = -1; }
    return;
  }
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  const bool any_final = ToG(p.any_final)[utt] != 0;
  int next_id = 0;
  for (int f = 0; f <= F; ++f) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    int frame_alive = 0;
    for (int c0 = 0; c0 < n; c0 += kLBlock) {
      const int i = c0 + tid;
      const bool a = i < n && Alive(p, base + i);
      int tot;
      const int off = BlockExclScanI(s, a ? 1 : 0, tot);
      if (i < n) {
        const int id = a ? next_id + off : -1;
        ToG(p.tok_newid)[base + i] = id;
        if (a) {
          if (f == 0 && ToG(p.tok_state)[base + i] == p.start) s.start_id = id;
          if (f == F) {
            const float fc = ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost;
            if (!(p.use_final_probs && any_final) || fc != Inf()) atomicAdd(&s.n_finals, 1);
          }
        }
      }
      next_id += tot;
      frame_alive += tot;
    }
    if (frame_alive == 0 && tid == 0) s.empty_frame = 1;  // :615-619
    auto count_links = [&](long long lb, int ln) {
      int c = 0;
      for (int i = tid; i < ln; i += kLBlock) c += ToG(p.link_dst)[lb + i] >= 0;
      if (c) atomicAdd(&s.n_arcs, c);
    };
    if (f < F) count_links(ToG(p.link_emit_base)[fb + f], ToG(p.link_emit_count)[fb + f]);
    count_links(ToG(p.link_eps_base)[fb + f], ToG(p.link_eps_count)[fb + f]);
  }
  __syncthreads();
  if (tid == 0) {
    if (s.empty_frame) { sz[0] = 0; sz[1] = 0; sz[2] = 0; sz[3] = -1; }
    else { sz[0] = next_id; sz[1] = s.n_arcs; sz[2] = s.n_finals; sz[3] = s.start_id; }
  }
}

__global__ __launch_bounds__(kLBlock) void LatticeFillKernel(const LatticeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LShared &s = *reinterpret_cast<LShared *>(smem);
  const int tid = threadIdx.x, utt = blockIdx.x;
  const int F = ToG(p.frames_decoded)[utt];
  if (F < 1 || ToG(p.sizes)[4 * utt] == 0) return;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  const bool any_final = ToG(p.any_final)[utt] != 0;
  const long long so = ToG(p.out_state_off)[utt], ao = ToG(p.out_arc_off)[utt],
                  fo = ToG(p.out_final_off)[utt];
  if (tid == 0) { s.n_arcs = 0; s.n_finals = 0; }
  __syncthreads();
  for (int f = 0; f <= F; ++f) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    for (int i = tid; i < n; i += kLBlock) {
      const int id = ToG(p.tok_newid)[base + i];
      if (id < 0) continue;
      const int32_t st = ToG(p.tok_state)[base + i];
      ToG(p.o_state_frame)[so + id] = f;
      ToG(p.o_state_graph)[so + id] = st;
      if (f == F) {
        const float fc = ToG(p.states)[st].final_cost;
        if (p.use_final_probs && any_final) {  // :645-652
          if (fc != Inf()) {
            const int k = atomicAdd(&s.n_finals, 1);
            ToG(p.o_final_state)[fo + k] = id;
            ToG(p.o_final_graph)[fo + k] = fc;
          }
        } else {
          const int k = atomicAdd(&s.n_finals, 1);
          ToG(p.o_final_state)[fo + k] = id;
          ToG(p.o_final_graph)[fo + k] = 0.0f;
        }
      }
    }
    auto fill_links = [&](long long lb, int ln) {
      for (int i = tid; i < ln; i += kLBlock) {
        const long long at = lb + i;
        const int32_t dst = ToG(p.link_dst)[at];
        if (dst < 0) continue;
        const int k = atomicAdd(&s.n_arcs, 1);
        const int32_t arc = ToG(p.link_arc)[at];
        ToG(p.o_src)[ao + k] = ToG(p.tok_newid)[ToG(p.link_src)[at]];
        ToG(p.o_dst)[ao + k] = ToG(p.tok_newid)[dst];
        ToG(p.o_ilabel)[ao + k] = ToG(p.arcs)[arc].ilabel;
        ToG(p.o_olabel)[ao + k] = ToG(p.arcs)[arc].olabel;
        ToG(p.o_graph)[ao + k] = ToG(p.link_graph)[at];
        ToG(p.o_ac)[ao + k] = ToG(p.link_ac)[at];
      }
    };
    if (f < F) fill_links(ToG(p.link_emit_base)[fb + f], ToG(p.link_emit_count)[fb + f]);
    fill_links(ToG(p.link_eps_base)[fb + f], ToG(p.link_eps_count)[fb + f]);
  }
}

hipError_t SetLds(const void *fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                             static_cast<int>(sizeof(LShared)));
}

}  // namespace

hipError_t LaunchLatticePrune(const LatticeParams &p, int batch, hipStream_t stream) {
  hipError_t e = SetLds(reinterpret_cast<const void *>(LatticePruneKernel));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(LatticePruneKernel, dim3(batch), dim3(kLBlock), sizeof(LShared), stream, p);
  return hipGetLastError();
}

hipError_t LaunchLatticeCount(const LatticeParams &p, int batch, hipStream_t stream) {
  hipError_t e = SetLds(reinterpret_cast<const void *>(LatticeCountKernel));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(LatticeCountKernel, dim3(batch), dim3(kLBlock), sizeof(LShared), stream, p);
  return hipGetLastError();
}

hipError_t LaunchLatticeFill(const LatticeParams &p, int batch, hipStream_t stream) {
  hipError_t e = SetLds(reinterpret_cast<const void *>(LatticeFillKernel));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(LatticeFillKernel, dim3(batch), dim3(kLBlock), sizeof(LShared), stream, p);
  return hipGetLastError();
}

}  // namespace kd
