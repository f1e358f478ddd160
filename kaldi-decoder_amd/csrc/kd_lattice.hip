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
// One workgroup (KD_LBLOCK threads) per utterance walks its frames backwards.  A frame's
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

// Geometry knobs.  Measured on MI355X (B = 128, six batches in flight): 256-thread
// workgroups with 48 KiB of LDS (three per CU) take 3.4 ms instead of 1.5 ms per
// batch for the pruning pass and leave the throughput unchanged, so the shipped
// configuration is one 1024-thread workgroup per CU; the export kernels only ask
// for the LDS they use.
#ifndef KD_LBLOCK
#define KD_LBLOCK 1024
#endif
#ifndef KD_LLDS_TOKS
#define KD_LLDS_TOKS 36864
#endif
#ifndef KD_LEPS_REGS
#define KD_LEPS_REGS 4
#endif
constexpr int kLBlock = KD_LBLOCK;
constexpr int kLWaves = kLBlock / 64;
constexpr int kLdsToks = KD_LLDS_TOKS;  // extra-cost keys of one frame kept in LDS (144 KiB)
constexpr int kFwToks = 4096;  // export kernels: per-frame words kept in LDS (16 KiB)
constexpr uint32_t kFInfKey = 0xFF800000u;  // FloatKey(+inf)
constexpr int kEpsRegs = KD_LEPS_REGS;  // epsilon links of a frame a lane keeps in registers

#define KD_G __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ KD_G T *ToG(T *p) { return (KD_G T *)p; }
template <class T>
__device__ __forceinline__ T *ToFlat(KD_G T *p) { return (T *)p; }

template <int N>
struct LSharedT {
  static constexpr int kWords = N;
  uint32_t ne[N];
  int32_t red_i[kLWaves];
  uint32_t red_u[kLWaves];
  uint32_t best_key, bestwf_key;
  int32_t any_final, changed, count, start_id, empty_frame;
  int32_t n_states, n_arcs, n_finals;
};
typedef LSharedT<kLdsToks> LShared;  // pruning
typedef LSharedT<kFwToks> CShared;   // export

__device__ __forceinline__ float Inf() { return __int_as_float(0x7F800000); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// every outstanding global load and store of the wave; in the LDS-resident frames of
// the pruning pass the waves exchange data through LDS alone, and the link records
// requested for later steps (and the stores of finished values) stay in flight.
__device__ __forceinline__ void LdsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int BlockExclScanI(CShared &s, int v, int &total) {
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

// Applies one link block to the frame's new-extra keys (frames that do not fit the
// LDS-resident path: thousands of tokens and links).  Four links per lane at a time:
// their records, then the four destination extras (a gather), are requested together
// (every lane issues every load -- a lane without a link reads a valid dummy --
// because hipcc waits for a conditional load before it issues the next).
__device__ __forceinline__ void ApplyLinks(const LatticeParams &p, LShared &s, KD_G uint32_t *ne,
                                           bool ne_in_lds, long long lb, int ln, long long base) {
  constexpr int U = 4;
  for (int i0 = threadIdx.x; i0 < ln; i0 += U * kLBlock) {
    int32_t dst[U], src[U];
    float delta[U], next_extra[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kLBlock;
      const long long at = lb + (i < ln ? i : i0);
      dst[u] = ToG(p.link_dst)[at];
      src[u] = ToG(p.link_src)[at];
      delta[u] = ToG(p.link_delta)[at];
      if (i >= ln) dst[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) next_extra[u] = ToG(p.tok_extra)[dst[u] < 0 ? base : dst[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (dst[u] < 0) continue;  // already excised (or no link)
      // lattice-simple-decoder.cc:257-260; (tok.tot + ac + graph) - next.tot was
      // taken when the link was made (both costs were final then)
      float link_extra_cost = next_extra[u] + delta[u];
      if (link_extra_cost > p.lattice_beam) {  // :264 excise
        ToG(p.link_dst)[lb + i0 + u * kLBlock] = -1;
        continue;
      }
      if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
      const uint32_t k = FloatKey(link_extra_cost);
      const long long li = src[u] - base;
      if (ne_in_lds) atomicMin(&s.ne[li], k);
      else atomicMin(ToFlat(&ne[li]), k);
    }
  }
}

__global__ __launch_bounds__(kLBlock) void LatticePruneKernel(const LatticeParams p) {
  __shared__ LShared s;  // (static: see kd_search.hip's kd_smem)
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

  const uint32_t *next_b = nullptr;  // the next frame's final extras, while they are in LDS
  long long next_base = 0;
  // frame descriptors are fetched one frame ahead
  struct Desc { long long base, eb, pb; int n, en, pn; };
  auto describe = [&](int f) {
    Desc d;
    d.base = ToG(p.frame_base)[fb + f];
    d.n = ToG(p.frame_count)[fb + f];
    d.eb = f == F ? 0 : ToG(p.link_emit_base)[fb + f];
    d.en = f == F ? 0 : ToG(p.link_emit_count)[fb + f];
    d.pb = ToG(p.link_eps_base)[fb + f];
    d.pn = ToG(p.link_eps_count)[fb + f];
    return d;
  };
  Desc ahead = describe(F);
  for (int f = F; f >= 0; --f) {
    const Desc d = ahead;
    if (f > 0) ahead = describe(f - 1);
    const long long base = d.base;
    const int n = d.n;
    const bool last = f == F;
    const long long eb = d.eb;
    const int en = d.en;
    const long long pb = d.pb;
    const int pn = d.pn;
    if (3 * n <= kLdsToks / 2 && pn <= kEpsRegs * kLBlock) {
      // The frame's fixpoint entirely in LDS.  Emitting links end in the next
      // frame, whose extras are final: their contribution (A) is computed once.
      // Epsilon links are fetched once into registers; each round recomputes
      // C = min(A, epsilon links over the current estimate B) and B := C, as the
      // loop below does through global memory.  Frames alternate between the two
      // halves of the key array, so the next frame's extras are still there.
      uint32_t *half = s.ne + (f & 1) * (kLdsToks / 2);
      uint32_t *A = half, *B = half + n, *C = half + 2 * n;
      for (int i = tid; i < n; i += kLBlock) {
        uint32_t k = kFInfKey;
        if (last) {  // the token's own final term (:467)
          const float tot = static_cast<float>(ToG(p.tok_cost)[base + i]);
          const float fc =
              any_final ? ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost : 0.0f;
          k = FloatKey(tot + fc - final_best_cost);
        }
        A[i] = k;
        B[i] = FloatKey(0.0f);  // tok_extra as the search left it
      }
      // this lane's epsilon links: source, destination (frame-relative) and the
      // cost difference of :257-260
      // (every lane issues every load -- a lane without a link reads the arena's first
      // record -- so the epsilon links and the first round of emitting links are in
      // flight together; loads behind `if (i < pn)` complete one after the other)
      int e_src[kEpsRegs], e_dst[kEpsRegs];
      float e_delta[kEpsRegs];
#pragma unroll
      for (int r = 0; r < kEpsRegs; ++r) {
        const int i = tid + r * kLBlock;
        const long long at = i < pn ? pb + i : 0;
        e_dst[r] = ToG(p.link_dst)[at];
        e_src[r] = ToG(p.link_src)[at];
        e_delta[r] = ToG(p.link_delta)[at];
      }
      const long long at0 = tid < en ? eb + tid : 0;
      const int32_t m_dst0 = ToG(p.link_dst)[at0], m_src0 = ToG(p.link_src)[at0];
      const float m_delta0 = ToG(p.link_delta)[at0];
#pragma unroll
      for (int r = 0; r < kEpsRegs; ++r) {
        const int i = tid + r * kLBlock;
        if (i < pn && e_dst[r] >= 0) {
          e_src[r] = static_cast<int>(e_src[r] - base);
          e_dst[r] = static_cast<int>(e_dst[r] - base);
        } else {
          e_src[r] = -1;
          e_dst[r] = -1;
          e_delta[r] = 0.0f;
        }
      }
      LdsBarrier();
      // emitting links, into A
      for (int i = tid; i < en; i += kLBlock) {
        const long long at = eb + i;
        const bool first = i == tid;  // (the first round was requested above)
        const int32_t dst = first ? m_dst0 : ToG(p.link_dst)[at];
        const int32_t src = first ? m_src0 : ToG(p.link_src)[at];
        const float delta = first ? m_delta0 : ToG(p.link_delta)[at];
        if (dst < 0) continue;  // already excised
        const float next_extra = next_b ? KeyFloat(next_b[dst - next_base]) : ToG(p.tok_extra)[dst];
        float link_extra_cost = next_extra + delta;  // :257-260
        if (link_extra_cost > p.lattice_beam) {  // :264 excise
          ToG(p.link_dst)[at] = -1;
          continue;
        }
        if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
        atomicMin(&A[src - base], FloatKey(link_extra_cost));
      }
      LdsBarrier();
      while (true) {
        for (int i = tid; i < n; i += kLBlock) C[i] = A[i];
        if (tid == 0) s.changed = 0;
        LdsBarrier();
#pragma unroll
        for (int r = 0; r < kEpsRegs; ++r) {
          if (e_src[r] < 0) continue;
          float link_extra_cost = KeyFloat(B[e_dst[r]]) + e_delta[r];
          if (link_extra_cost > p.lattice_beam) {  // :264 excise
            ToG(p.link_dst)[pb + tid + r * kLBlock] = -1;
            e_src[r] = -1;
            continue;
          }
          if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
          atomicMin(&C[e_src[r]], FloatKey(link_extra_cost));
        }
        LdsBarrier();
        int ch = 0;
        for (int i = tid; i < n; i += kLBlock) {
          float v = KeyFloat(C[i]);
          if (last && v > p.lattice_beam) v = Inf();  // :507-509
          const uint32_t k = FloatKey(v);
          if (k != B[i]) { B[i] = k; ch = 1; }
        }
        if (ch) s.changed = 1;
        LdsBarrier();
        const int changed = s.changed;
        LdsBarrier();
        // without epsilon links nothing in this frame depends on this frame
        if (!changed || pn == 0) break;
      }
      for (int i = tid; i < n; i += kLBlock) ToG(p.tok_extra)[base + i] = KeyFloat(B[i]);
      next_b = B;
      next_base = base;
      // (a later frame that reads these extras from global memory -- the path below --
      // does so behind a full barrier of its own)
      LdsBarrier();
      continue;
    }
    next_b = nullptr;  // this frame's extras go to global memory only
    const bool ne_in_lds = n <= kLdsToks;
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
      for (int i0 = tid; i0 < n; i0 += 4 * kLBlock) {  // (four tokens per lane in flight)
        float was[4];
        uint32_t now[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kLBlock < n ? i0 + u * kLBlock : i0;
          was[u] = ToG(p.tok_extra)[base + i];
          now[u] = ne_in_lds ? s.ne[i]
                             : __hip_atomic_load(ToFlat(&gne[i]), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u * kLBlock >= n) continue;
          float v = KeyFloat(now[u]);
          if (last && v > p.lattice_beam) v = Inf();  // :507-509
          if (v != was[u]) {
            ToG(p.tok_extra)[base + i0 + u * kLBlock] = v;
            ch = 1;
          }
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

// GetRawLattice numbers the live tokens frame by frame (:611-622).  Frames do not
// depend on one another here, so the waves of the workgroup take frames
// independently (a wave needs no barrier to rank the tokens of its frame: ballot
// and popcount), and the links are one flat, frame-less range.
//
// Per-frame counts / id offsets live in LDS (the prune kernel's key array), or in
// the utterance's global scratch when there are more frames than that.
__device__ __forceinline__ uint32_t *FrameWords(const LatticeParams &p, CShared &s, int utt, int F) {
  return F + 1 <= kFwToks ? s.ne : ToFlat(ToG(p.scratch) + utt * p.scratch_stride);
}

// Calls visit(frame f, token i, alive, rank among the frame's live tokens so far)
// for every token of the frames this wave owns; returns nothing.  kT tokens per
// lane are fetched before any is ranked.
template <class Visit>
__device__ __forceinline__ void ForWaveFrames(const LatticeParams &p, int utt, int F, Visit visit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  constexpr int kT = 4;
  for (int f = wave; f <= F; f += kLWaves) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    int before = 0;
    for (int c0 = 0; c0 < n; c0 += 64 * kT) {
      bool a[kT];
#pragma unroll
      for (int u = 0; u < kT; ++u) {
        const int i = c0 + u * 64 + lane;
        a[u] = i < n && Alive(p, base + i);
      }
#pragma unroll
      for (int u = 0; u < kT; ++u) {
        const int i = c0 + u * 64 + lane;
        const unsigned long long m = __ballot(a[u]);
        const int rank = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
        if (i < n) visit(f, base + i, a[u], rank);
        before += __popcll(m);
      }
    }
    visit(f, -1ll, false, before);  // end of frame: `before` live tokens
  }
}

__global__ __launch_bounds__(kLBlock) void LatticeCountKernel(const LatticeParams p) {
  __shared__ CShared s;
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
  uint32_t *fw = FrameWords(p, s, utt, F);
  // live tokens per frame
  ForWaveFrames(p, utt, F, [&](int f, long long g, bool, int rank) {
    if (g < 0 && (threadIdx.x & 63) == 0) {
      fw[f] = static_cast<uint32_t>(rank);
      if (rank == 0) s.empty_frame = 1;  // :615-619
    }
  });
  __syncthreads();
  // first id of each frame (exclusive scan over the frames)
  int carry = 0;
  for (int f0 = 0; f0 <= F; f0 += kLBlock) {
    const int f = f0 + tid;
    const int v = f <= F ? static_cast<int>(fw[f]) : 0;
    int tot;
    const int off = BlockExclScanI(s, v, tot);
    if (f <= F) fw[f] = static_cast<uint32_t>(carry + off);
    carry += tot;
  }
  __syncthreads();
  // ids, the start state, the final states
  ForWaveFrames(p, utt, F, [&](int f, long long g, bool alive, int rank) {
    if (g < 0) return;
    const int id = alive ? static_cast<int>(fw[f]) + rank : -1;
    ToG(p.tok_newid)[g] = id;
    if (!alive) return;
    if (f == 0 && ToG(p.tok_state)[g] == p.start) s.start_id = id;
    if (f == F) {
      const float fc = ToG(p.states)[ToG(p.tok_state)[g]].final_cost;
      if (!(p.use_final_probs && any_final) || fc != Inf()) atomicAdd(&s.n_finals, 1);
    }
  });
  // surviving links: one flat range per utterance (frame 0's epsilon links come
  // first, frame F's epsilon links last)
  {
    const long long l0 = ToG(p.link_eps_base)[fb];
    const long long l1 = ToG(p.link_eps_base)[fb + F] + ToG(p.link_eps_count)[fb + F];
    int c = 0;
    for (long long at = l0 + tid; at < l1; at += kLBlock) c += ToG(p.link_dst)[at] >= 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((tid & 63) == 0 && c) atomicAdd(&s.n_arcs, c);
  }
  __syncthreads();
  if (tid == 0) {
    if (s.empty_frame) { sz[0] = 0; sz[1] = 0; sz[2] = 0; sz[3] = -1; }
    else { sz[0] = carry; sz[1] = s.n_arcs; sz[2] = s.n_finals; sz[3] = s.start_id; }
  }
}

__global__ __launch_bounds__(kLBlock) void LatticeFillKernel(const LatticeParams p) {
  __shared__ CShared s;
  const int tid = threadIdx.x, utt = blockIdx.x;
  const int F = ToG(p.frames_decoded)[utt];
  if (F < 1 || ToG(p.sizes)[4 * utt] == 0) return;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  const bool any_final = ToG(p.any_final)[utt] != 0;
  const long long so = ToG(p.out_state_off)[utt], ao = ToG(p.out_arc_off)[utt],
                  fo = ToG(p.out_final_off)[utt];
  if (tid == 0) { s.n_arcs = 0; s.n_finals = 0; }
  __syncthreads();
  // states (and final weights): frames are independent, one wave each
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int f = wave; f <= F; f += kLWaves) {
      const long long base = ToG(p.frame_base)[fb + f];
      const int n = ToG(p.frame_count)[fb + f];
      for (int i = lane; i < n; i += 64) {
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
    }
  }
  // arcs: the utterance's links as one flat range, four per lane in flight
  {
    const long long l0 = ToG(p.link_eps_base)[fb];
    const long long l1 = ToG(p.link_eps_base)[fb + F] + ToG(p.link_eps_count)[fb + F];
    constexpr int kT = 4;
    for (long long a0 = l0 + tid; a0 < l1; a0 += kT * kLBlock) {
      int32_t dst[kT];
#pragma unroll
      for (int u = 0; u < kT; ++u) {
        const long long at = a0 + u * kLBlock;
        dst[u] = at < l1 ? ToG(p.link_dst)[at] : -1;
      }
#pragma unroll
      for (int u = 0; u < kT; ++u) {
        if (dst[u] < 0) continue;
        const long long at = a0 + u * kLBlock;
        const int k = atomicAdd(&s.n_arcs, 1);
        const int32_t arc = ToG(p.link_arc)[at];
        ToG(p.o_src)[ao + k] = ToG(p.tok_newid)[ToG(p.link_src)[at]];
        ToG(p.o_dst)[ao + k] = ToG(p.tok_newid)[dst[u]];
        ToG(p.o_ilabel)[ao + k] = ToG(p.arcs)[arc].ilabel;
        ToG(p.o_olabel)[ao + k] = ToG(p.arcs)[arc].olabel;
        ToG(p.o_graph)[ao + k] = ToG(p.link_graph)[at];
        ToG(p.o_ac)[ao + k] = ToG(p.link_ac)[at];
      }
    }
  }
}

}  // namespace

hipError_t LaunchLatticePrune(const LatticeParams &p, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(LatticePruneKernel, dim3(batch), dim3(kLBlock), 0, stream, p);
  return hipGetLastError();
}

hipError_t LaunchLatticeCount(const LatticeParams &p, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(LatticeCountKernel, dim3(batch), dim3(kLBlock), 0, stream, p);
  return hipGetLastError();
}

hipError_t LaunchLatticeFill(const LatticeParams &p, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(LatticeFillKernel, dim3(batch), dim3(kLBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace kd
