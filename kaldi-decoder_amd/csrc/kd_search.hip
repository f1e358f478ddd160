// kaldi-decoder_amd/csrc/kd_search.hip -- the per-frame token-passing search
// (FasterDecoder::ProcessEmitting / ProcessNonemitting / GetCutoff and the
// HashList they run on) as one persistent gfx950 kernel.
//
// Replaces, from /root/reference/kaldi-decoder/csrc/:
//   faster-decoder.cc:155-241  ProcessEmitting
//   faster-decoder.cc:59-119   ProcessNonemitting
//   faster-decoder.cc:244-336  GetCutoff (beam / max_active / min_active)
//   faster-decoder.cc:42-56    InitDecoding
//   hash-list-inl.h:26-173     HashList (-> open-addressed token table in LDS)
//   faster-decoder.h:110-156   Token (-> append-only arena record)
//
// Mapping: one 1024-thread workgroup owns one utterance for the whole launch
// and loops over its frames; utterances never interact, so the only
// synchronisation is s_barrier inside the workgroup and a frame costs no
// kernel launch.  Per frame:
//   A  cutoff      min / counts / radix-select over the previous frontier
//   B  emitting    arc-parallel expansion: a block-wide prefix sum over the
//                  out-degrees of the surviving tokens spreads their arcs
//                  over the lanes (16-byte arc records, consecutive lanes read
//                  consecutive arcs), the log-likelihood row sits in LDS, and
//                  each candidate does one 64-bit ds_min on the destination
//                  state's slot of the LDS token table
//   C  epsilon     same machinery over epsilon arc ranges, iterated to a
//                  fixpoint with a worklist of improved slots
//   D  compaction  live slots -> one contiguous arena block (state, cost,
//                  prev, arc); the table is cleared for the next frame
// The table holds 8192 slots (key, 64-bit cost key, winning arc) = 128 KiB of
// the CU's 160 KiB LDS.  A frame that would overfill it is re-run against a
// larger table in global memory (same code, template parameter LDS=false).
//
// Semantics are the reference's, made order-free (DESIGN.md "Order
// dependence"): costs are double and use the reference's expressions, the
// emitting cutoff is the value the reference's running cutoff ends at
// (the running value is used only to skip work early), strict improvement
// replaces, and ties pick the lowest arc index.  The oracle's policy 1 states
// the same semantics on the CPU and the two are compared bit for bit.

#include <float.h>
#include <limits.h>

#include "kd_device.h"
#include "kd_hip.h"

namespace kd {
namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kLog2Slots = 13;
constexpr int kSlots = 1 << kLog2Slots;
constexpr int kSlotsPerThread = kSlots / kBlock;
constexpr int kLdsFillLimit = kSlots * 3 / 4;
constexpr int kVmax = 2048;  // log-likelihood row staged in LDS when V <= kVmax
constexpr int kU = 2;        // arcs per lane per batch
constexpr unsigned long long kCostInf = 0xFFF0000000000000ull;  // CostKey(+inf)

struct Shared {
  // token table (LDS mode); in spill mode `cost` is reused as the worklist bitmask
  unsigned long long cost[kSlots];
  uint32_t key[kSlots];
  uint32_t arc[kSlots];
  // expansion scratch: one entry per item of the current chunk
  double it_cost[kBlock];
  int32_t it_scan[kBlock];
  int32_t it_abeg[kBlock];
  uint32_t it_src[kBlock];
  float ll[kVmax];
  uint32_t bitmask[kSlots / 32];
  int32_t hist[256];
  unsigned long long red64[kWaves];
  int32_t red32[kWaves];
  unsigned long long min_new;
  long long base_out;
  int32_t scan_total;
  int32_t fill;
  int32_t overflow;
  int32_t wl_count[2];
  int32_t sel_bin, sel_k;
  int32_t stop;
};
static_assert(sizeof(Shared) <= 163840, "LDS budget");

// Pointers that arrive inside the by-value parameter struct are generic
// ("flat") to the compiler; casting them to the global address space at the use
// site turns flat_load/flat_atomic into global_load/global_atomic.
#define KD_G __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ KD_G T *ToG(T *p) { return (KD_G T *)p; }
template <class T>
__device__ __forceinline__ T *ToFlat(KD_G T *p) { return (T *)p; }

typedef int v4i __attribute__((ext_vector_type(4)));
// 16-byte records are fetched with one global_load_dwordx4.
__device__ __forceinline__ StateRec LoadState(const StateRec *base, uint32_t i) {
  const v4i v = reinterpret_cast<const KD_G v4i *>(ToG(base))[i];
  StateRec r;
  r.arc_begin = v.x; r.emit_end = v.y; r.arc_end = v.z; r.final_cost = __int_as_float(v.w);
  return r;
}
__device__ __forceinline__ ArcRec LoadArcRec(const ArcRec *base, uint32_t i) {
  const v4i v = reinterpret_cast<const KD_G v4i *>(ToG(base))[i];
  ArcRec r;
  r.ilabel = v.x; r.olabel = v.y; r.weight = __int_as_float(v.z); r.next = static_cast<uint32_t>(v.w);
  return r;
}

struct Counters {
  long long tokens_expanded = 0, arcs_scanned = 0, arcs_kept = 0, eps_arcs = 0, eps_exp = 0;
};

// ---- block-wide primitives (wave64 shuffles + one LDS hop) -----------------
__device__ __forceinline__ int BlockExclScan(Shared &s, int v, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s.red32[wave] = x;
  __syncthreads();
  if (wave == 0) {
    int w = lane < kWaves ? s.red32[lane] : 0, xs = w;
#pragma unroll
    for (int d = 1; d < kWaves; d <<= 1) {
      int y = __shfl_up(xs, d, 64);
      if (lane >= d) xs += y;
    }
    if (lane < kWaves) s.red32[lane] = xs - w;
    if (lane == kWaves - 1) s.scan_total = xs;
  }
  __syncthreads();
  int off = s.red32[wave] + x - v;
  total = s.scan_total;
  __syncthreads();
  return off;
}

__device__ __forceinline__ int BlockSum(Shared &s, int v) {
  int total;
  BlockExclScan(s, v, total);
  return total;
}

__device__ __forceinline__ unsigned long long BlockMinU64(Shared &s, unsigned long long v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    unsigned long long y = __shfl_xor(v, d, 64);
    v = y < v ? y : v;
  }
  if (lane == 0) s.red64[wave] = v;
  __syncthreads();
  unsigned long long r = s.red64[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) r = s.red64[w] < r ? s.red64[w] : r;
  __syncthreads();
  return r;
}

// ---- table access: LDS (ds_* ops) or global memory (L2 atomics) --------------
template <bool LDS>
struct Table {
  Shared &s;
  KD_G uint32_t *gkey;
  KD_G unsigned long long *gcost;
  KD_G uint32_t *garc;
  KD_G uint32_t *src;   // always global
  KD_G uint32_t *gidx;
  KD_G uint32_t *gused;
  uint32_t *bitmask;
  int log2_slots;
  uint32_t mask;

  __device__ __forceinline__ uint32_t *KeyP(uint32_t h) const { if constexpr (LDS) return &s.key[h]; else return ToFlat(&gkey[h]); }
  __device__ __forceinline__ unsigned long long *CostP(uint32_t h) const { if constexpr (LDS) return &s.cost[h]; else return ToFlat(&gcost[h]); }
  __device__ __forceinline__ uint32_t *ArcP(uint32_t h) const { if constexpr (LDS) return &s.arc[h]; else return ToFlat(&garc[h]); }

  // Global-table words are updated by L2 atomics, which do not refresh this
  // CU's L1: read them with agent-scope (sc1) loads.
  __device__ __forceinline__ uint32_t LoadKey(uint32_t h) const {
    if constexpr (LDS) return s.key[h];
    else return __hip_atomic_load(ToFlat(&gkey[h]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ unsigned long long LoadCost(uint32_t h) const {
    if constexpr (LDS) return s.cost[h];
    else return __hip_atomic_load(ToFlat(&gcost[h]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ uint32_t LoadArc(uint32_t h) const {
    if constexpr (LDS) return s.arc[h];
    else return __hip_atomic_load(ToFlat(&garc[h]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void StoreCost(uint32_t h, unsigned long long v) const {
    if constexpr (LDS) s.cost[h] = v;
    else __hip_atomic_store(ToFlat(&gcost[h]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void StoreArc(uint32_t h, uint32_t v) const {
    if constexpr (LDS) s.arc[h] = v;
    else __hip_atomic_store(ToFlat(&garc[h]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ uint32_t Hash(uint32_t state) const {
    return (state * 2654435761u) >> (32 - log2_slots);
  }

  // Finds or claims the slot of `k` (state | has-eps flag).  Returns kKeyEmpty
  // when the table is over its fill limit or the probe sequence is exhausted.
  __device__ __forceinline__ uint32_t Claim(uint32_t k) const {
    uint32_t h = Hash(k & kStateMask);
    const uint32_t limit = LDS ? kLdsFillLimit : (3u << (log2_slots - 2));
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      uint32_t cur = LoadKey(h);
      if (cur == k) return h;
      if (cur == kKeyEmpty) {
        uint32_t old = atomicCAS(KeyP(h), kKeyEmpty, k);
        if (old == kKeyEmpty) {
          int pos = atomicAdd(&s.fill, 1);
          if constexpr (!LDS) {
            if (static_cast<uint32_t>(pos) <= mask) gused[pos] = h;
          }
          if (static_cast<uint32_t>(pos) >= limit) s.overflow = 1;
          return h;
        }
        if (old == k) return h;
      }
      h = (h + 1) & mask;
    }
    s.overflow = 1;
    return kKeyEmpty;
  }
};

struct Cand {
  uint32_t h, arc, src;
  unsigned long long nk;
};

struct FrameCtx {
  const SearchParams &p;
  Shared &s;
  int utt;
  long long base_in;
  int n_in;
  const KD_G float *ll_row;  // global row (used when V > kVmax)
  bool ll_in_lds;
  double weight_cutoff;  // tokens with cost < this are expanded
  float adaptive_beam;
  double cutoff;  // epsilon-closure cutoff == final emitting cutoff
  KD_G uint32_t *wl[2];
};

// Expands the emitting (EPS=false) arcs of the input frontier, or the epsilon
// (EPS=true) arcs of the tokens on worklist `wl_in`, into the table.
// Returns false if the table overflowed (the caller re-runs the frame).
template <bool LDS, bool EPS>
__device__ __forceinline__ bool Expand(const FrameCtx &c, const Table<LDS> &t, int n_items,
                                       const KD_G uint32_t *wl_in, KD_G uint32_t *wl_out,
                                       int wl_out_idx,
                                       Counters &cnt) {
  const SearchParams &p = c.p;
  Shared &s = c.s;
  const int tid = threadIdx.x;
  Cand prev[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) prev[u].h = kKeyEmpty;

  for (int chunk = 0; chunk < n_items; chunk += kBlock) {
    const int i = chunk + tid;
    int deg = 0, abeg = 0;
    double cst = 0.0;
    uint32_t srcid = 0;
    if (i < n_items) {
      uint32_t st;
      bool ok;
      if constexpr (!EPS) {
        st = static_cast<uint32_t>(ToG(p.tok_state)[c.base_in + i]);
        cst = ToG(p.tok_cost)[c.base_in + i];
        srcid = static_cast<uint32_t>(i);
        ok = cst < c.weight_cutoff;  // faster-decoder.cc:202
      } else {
        const uint32_t hs = wl_in[i];
        st = t.LoadKey(hs) & kStateMask;
        cst = KeyCost(t.LoadCost(hs));
        srcid = hs | kSrcEps;
        ok = !(cst > c.cutoff);  // faster-decoder.cc:74
      }
      if (ok) {
        const StateRec r = LoadState(p.states, st);
        abeg = EPS ? r.emit_end : r.arc_begin;
        deg = EPS ? r.arc_end - r.emit_end : r.emit_end - r.arc_begin;
        if (EPS) cnt.eps_exp += 1; else cnt.tokens_expanded += 1;
      }
    }
    int total;
    const int off = BlockExclScan(s, deg, total);
    s.it_scan[tid] = off;
    s.it_abeg[tid] = abeg;
    s.it_cost[tid] = cst;
    s.it_src[tid] = srcid;
    __syncthreads();
    if (tid == 0) { if (EPS) cnt.eps_arcs += total; else cnt.arcs_scanned += total; }

    for (int base = 0; base < total; base += kBlock * kU) {
      // back-pointer sources of the previous batch's winners (see below)
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (prev[u].h != kKeyEmpty && t.LoadArc(prev[u].h) == prev[u].arc) t.src[prev[u].h] = prev[u].src;

      double runcut;
      unsigned long long seen_min = 0;
      if constexpr (EPS) {
        runcut = c.cutoff;
      } else {
        seen_min = s.min_new;
        runcut = KeyCost(seen_min) + static_cast<double>(c.adaptive_beam);  // :215-217
      }
      Cand cur[kU];
      // phase 1: relax.  One 64-bit atomic min per surviving candidate; a
      // strict improvement invalidates the slot's winning arc.
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        cur[u].h = kKeyEmpty;
        const int j = base + u * kBlock + tid;
        if (j < total) {
          int lo = 0, hi = kBlock - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s.it_scan[mid] <= j) lo = mid; else hi = mid - 1;
          }
          const uint32_t aidx = static_cast<uint32_t>(s.it_abeg[lo] + (j - s.it_scan[lo]));
          const ArcRec a = LoadArcRec(p.arcs, aidx);
          const double csrc = s.it_cost[lo];
          double nc;
          bool keep;
          if constexpr (EPS) {
            nc = csrc + a.weight;      // faster-decoder.h:134
            keep = !(nc > runcut);     // faster-decoder.cc:92
          } else {
            const float ll = c.ll_in_lds ? s.ll[a.ilabel - 1] : c.ll_row[a.ilabel - 1];
            const float ac_cost = -1 * ll;          // :209
            nc = a.weight + csrc + ac_cost;         // :210 (== Token cost, faster-decoder.h:126)
            keep = nc < runcut;                     // :211
          }
          if (keep) {
            const uint32_t h = t.Claim(a.next);
            if (h != kKeyEmpty) {
              const unsigned long long nk = CostKey(nc);
              const unsigned long long old = atomicMin(t.CostP(h), nk);
              if (old > nk) {
                t.StoreArc(h, kArcNone);
                if constexpr (EPS) {
                  if (a.next & kHasEps) {  // re-queue on improvement, :97-113
                    const uint32_t bit = 1u << (h & 31);
                    if (!(atomicOr(&t.bitmask[h >> 5], bit) & bit))
                      wl_out[atomicAdd(&s.wl_count[wl_out_idx], 1)] = h;
                  }
                }
              }
              if (old >= nk) {
                cur[u].h = h;
                cur[u].arc = aidx;
                cur[u].src = s.it_src[lo];
                cur[u].nk = nk;
              }
              if constexpr (!EPS) {
                cnt.arcs_kept += 1;
                if (nk < seen_min) { atomicMin(&s.min_new, nk); seen_min = nk; }
              }
            }
          }
        }
      }
      __syncthreads();
      const int overflowed = s.overflow;  // only phase 1 writes it
      // phase 2: candidates that still hold the slot's cost compete for the
      // lowest arc index (deterministic tie-break).
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (cur[u].h != kKeyEmpty) {
          if (t.LoadCost(cur[u].h) == cur[u].nk) atomicMin(t.ArcP(cur[u].h), cur[u].arc);
          else cur[u].h = kKeyEmpty;
        }
      }
      __syncthreads();
      // phase 3 (deferred to the top of the next batch, after this barrier):
      // the lane whose arc won writes the back-pointer source.
#pragma unroll
      for (int u = 0; u < kU; ++u) prev[u] = cur[u];
      if (overflowed) return false;
    }
  }
#pragma unroll
  for (int u = 0; u < kU; ++u)
    if (prev[u].h != kKeyEmpty && t.LoadArc(prev[u].h) == prev[u].arc) t.src[prev[u].h] = prev[u].src;
  __syncthreads();
  return true;
}

// k-th smallest (0-based) of float(cost) over the input frontier: 4-pass
// byte-wise radix select on order-preserving keys (std::nth_element in
// faster-decoder.cc:298-321; tmp_array_ is std::vector<float>, :278).
__device__ __forceinline__ float SelectKth(const FrameCtx &c, int k) {
  Shared &s = c.s;
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t prefix = 0, pmask = 0;
  int kk = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) s.hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < c.n_in; i += kBlock) {
      const uint32_t key = FloatKey(static_cast<float>(ToG(c.p.tok_cost)[c.base_in + i]));
      if ((key & pmask) == prefix) atomicAdd(&s.hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int c0 = s.hist[4 * lane], c1 = s.hist[4 * lane + 1], c2 = s.hist[4 * lane + 2],
                c3 = s.hist[4 * lane + 3];
      const int tot = c0 + c1 + c2 + c3;
      int x = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
      }
      const int excl = x - tot;
      if (kk >= excl && kk < x) {
        int r = kk - excl, b = 0;
        if (r >= c0) { r -= c0; b = 1; if (r >= c1) { r -= c1; b = 2; if (r >= c2) { r -= c2; b = 3; } } }
        s.sel_bin = 4 * lane + b;
        s.sel_k = r;
      }
    }
    __syncthreads();
    prefix |= static_cast<uint32_t>(s.sel_bin) << shift;
    pmask |= 255u << shift;
    kk = s.sel_k;
    __syncthreads();
  }
  return KeyFloat(prefix);
}

// GetCutoff, faster-decoder.cc:244-336, by value.
__device__ __forceinline__ void GetCutoff(FrameCtx &c) {
  const SearchParams &p = c.p;
  Shared &s = c.s;
  const int tid = threadIdx.x;
  unsigned long long mk = kCostDead;
  for (int i = tid; i < c.n_in; i += kBlock) {
    const unsigned long long k = CostKey(ToG(p.tok_cost)[c.base_in + i]);
    mk = k < mk ? k : mk;
  }
  mk = BlockMinU64(s, mk);
  const double best_cost = KeyCost(mk);
  const double beam_cutoff = best_cost + p.beam;
  if (p.max_active == INT_MAX && p.min_active == 0) {  // :249-272
    c.weight_cutoff = beam_cutoff;
    c.adaptive_beam = p.beam;
    return;
  }
  int n_lt = 0, n_le = 0;
  for (int i = tid; i < c.n_in; i += kBlock) {
    const double v = static_cast<float>(ToG(p.tok_cost)[c.base_in + i]);  // narrowed, :278
    n_lt += v < beam_cutoff;
    n_le += v <= beam_cutoff;
  }
  n_lt = BlockSum(s, n_lt);
  n_le = BlockSum(s, n_le);
  // tmp_array_[k] < x  <=>  #{v < x} >= k+1 ;  tmp_array_[k] > x  <=>  #{v <= x} <= k
  if (c.n_in > p.max_active && static_cast<long long>(n_lt) >= static_cast<long long>(p.max_active) + 1) {
    const double max_active_cutoff = SelectKth(c, p.max_active);  // :297-309
    c.adaptive_beam = max_active_cutoff - best_cost + p.beam_delta;
    c.weight_cutoff = max_active_cutoff;
    return;
  }
  double min_active_cutoff = beam_cutoff;  // "not looser" unless proven below
  if (c.n_in > p.min_active) {             // :311-323
    if (p.min_active == 0) min_active_cutoff = best_cost;
    else if (n_le <= p.min_active) min_active_cutoff = SelectKth(c, p.min_active);
  } else {
    min_active_cutoff = __longlong_as_double(0x7FF0000000000000ll);
  }
  if (min_active_cutoff > beam_cutoff) {  // :325-335
    c.adaptive_beam = min_active_cutoff - best_cost + p.beam_delta;
    c.weight_cutoff = min_active_cutoff;
  } else {
    c.adaptive_beam = p.beam;
    c.weight_cutoff = beam_cutoff;
  }
}

template <bool LDS>
__device__ __forceinline__ void ClearTable(const Table<LDS> &t, Shared &s, int n_used) {
  const int tid = threadIdx.x;
  if constexpr (LDS) {
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
      const int h = k * kBlock + tid;
      s.key[h] = kKeyEmpty;
      s.cost[h] = kCostDead;
      s.arc[h] = kArcNone;
    }
  } else {
    for (int i = tid; i < n_used; i += kBlock) {
      const uint32_t h = t.gused[i];
      t.gkey[h] = kKeyEmpty;
      t.gcost[h] = kCostDead;
      t.garc[h] = kArcNone;
    }
  }
  if (tid == 0) { s.fill = 0; s.overflow = 0; }
  __syncthreads();
}

// One frame (frame >= 0) or InitDecoding (frame == -1).  Returns 0 on success,
// 1 if the table overflowed (nothing committed; table left dirty), or a
// KD_UTT_* bit on a fatal capacity error.
template <bool LDS>
__device__ __forceinline__ uint32_t RunFrame(FrameCtx &c, const Table<LDS> &t, int frame,
                                             Counters &total_cnt, int &n_out_ret) {
  const SearchParams &p = c.p;
  Shared &s = c.s;
  const int tid = threadIdx.x;
  Counters cnt;

  if (tid == 0) { s.min_new = kCostInf; s.wl_count[0] = 0; s.wl_count[1] = 0; }
  if (frame < 0) {
    if (tid == 0) {  // faster-decoder.cc:49-51: dummy arc into the start state, cost 0
      const StateRec r = LoadState(p.states, p.start);
      const uint32_t k = static_cast<uint32_t>(p.start) | (r.arc_end > r.emit_end ? kHasEps : 0u);
      const uint32_t h = t.Hash(static_cast<uint32_t>(p.start));
      *t.KeyP(h) = k;
      t.StoreCost(h, CostKey(0.0));
      t.StoreArc(h, kArcNone);
      t.src[h] = kSrcNone;
      s.fill = 1;
      if constexpr (!LDS) t.gused[0] = h;
    }
    c.cutoff = static_cast<double>(FLT_MAX);  // :53
    __syncthreads();
  } else {
    GetCutoff(c);
    __syncthreads();
    if (!Expand<LDS, false>(c, t, c.n_in, nullptr, nullptr, 0, cnt)) return 1;
    c.cutoff = KeyCost(s.min_new) + static_cast<double>(c.adaptive_beam);  // value of :240
  }

  // Drop tokens the final cutoff excludes (:211 with the final cutoff), seed
  // the epsilon worklist with every surviving token that has epsilon arcs.
  const unsigned long long cutkey = CostKey(c.cutoff);
  const int n_used = s.fill;
  {
    auto visit = [&](uint32_t h) {
      const uint32_t k = t.LoadKey(h);
      if (k == kKeyEmpty) return;
      const unsigned long long ck = t.LoadCost(h);
      if (ck == kCostDead) return;
      if (frame >= 0 && ck >= cutkey) { t.StoreCost(h, kCostDead); return; }
      if (k & kHasEps) c.wl[0][atomicAdd(&s.wl_count[0], 1)] = h;
    };
    if constexpr (LDS) {
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) visit(k * kBlock + tid);
    } else {
      for (int i = tid; i < n_used; i += kBlock) visit(t.gused[i]);
    }
  }
  __syncthreads();

  // ProcessNonemitting to a fixpoint (:59-119)
  int cur = 0;
  while (true) {
    const int n_items = s.wl_count[cur];
    if (n_items == 0) break;
    const int words = LDS ? kSlots / 32 : (1 << (t.log2_slots - 5));
    for (int w = tid; w < words; w += kBlock) t.bitmask[w] = 0;
    if (tid == 0) s.wl_count[cur ^ 1] = 0;
    __syncthreads();
    if (!Expand<LDS, true>(c, t, n_items, c.wl[cur], c.wl[cur ^ 1], cur ^ 1, cnt)) return 1;
    cur ^= 1;
  }
  if (s.overflow) return 1;

  // Compaction: live slots -> arena block `frame + 1`.
  const int n_slots_used = s.fill;
  int my_live = 0;
  uint32_t live_bits = 0;
  if constexpr (LDS) {
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
      const int h = k * kBlock + tid;
      if (s.key[h] != kKeyEmpty && s.cost[h] != kCostDead) { live_bits |= 1u << k; my_live++; }
    }
  } else {
    for (int i = tid; i < n_slots_used; i += kBlock)
      if (t.LoadCost(t.gused[i]) != kCostDead) my_live++;
  }
  int n_out;
  const int off = BlockExclScan(s, my_live, n_out);
  if (tid == 0) {
    const unsigned long long b =
        atomicAdd(ToFlat(ToG(p.arena_top)), static_cast<unsigned long long>(n_out));
    s.base_out = (b + n_out > p.arena_cap) ? -1 : static_cast<long long>(b);
  }
  __syncthreads();
  const long long base_out = s.base_out;
  if (base_out < 0) return KD_UTT_ARENA_FULL;
  {
    int idx = off;
    if constexpr (LDS) {
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        if (live_bits & (1u << k)) {
          const int h = k * kBlock + tid;
          const long long g = base_out + idx;
          ToG(p.tok_state)[g] = static_cast<int32_t>(s.key[h] & kStateMask);
          ToG(p.tok_cost)[g] = KeyCost(s.cost[h]);
          const uint32_t a = s.arc[h];
          ToG(p.tok_arc)[g] = static_cast<int32_t>(a);
          if (a == kArcNone) live_bits |= 1u << (16 + k);  // start token
          s.arc[h] = static_cast<uint32_t>(idx);
          idx++;
        }
      }
    } else {
      for (int i = tid; i < n_slots_used; i += kBlock) {
        const uint32_t h = t.gused[i];
        const unsigned long long ck = t.LoadCost(h);
        if (ck != kCostDead) {
          const long long g = base_out + idx;
          ToG(p.tok_state)[g] = static_cast<int32_t>(t.LoadKey(h) & kStateMask);
          ToG(p.tok_cost)[g] = KeyCost(ck);
          ToG(p.tok_arc)[g] = static_cast<int32_t>(t.LoadArc(h));
          t.gidx[h] = static_cast<uint32_t>(idx);
          idx++;
        }
      }
    }
  }
  __syncthreads();
  {
    int idx = off;
    auto prev_of = [&](uint32_t h, bool is_start) -> int32_t {
      if (is_start) return -1;
      const uint32_t sr = t.src[h];
      if (sr & kSrcEps) {
        const uint32_t hs = sr & ~kSrcEps;
        const uint32_t si = LDS ? s.arc[hs] : t.gidx[hs];
        return static_cast<int32_t>(base_out + si);
      }
      return static_cast<int32_t>(c.base_in + sr);
    };
    if constexpr (LDS) {
#pragma unroll
      for (int k = 0; k < kSlotsPerThread; ++k) {
        if (live_bits & (1u << k)) {
          ToG(p.tok_prev)[base_out + idx] = prev_of(k * kBlock + tid, (live_bits >> (16 + k)) & 1u);
          idx++;
        }
      }
    } else {
      for (int i = tid; i < n_slots_used; i += kBlock) {
        const uint32_t h = t.gused[i];
        if (t.LoadCost(h) != kCostDead) {
          const long long g = base_out + idx;
          ToG(p.tok_prev)[g] = prev_of(h, ToG(p.tok_arc)[g] == static_cast<int32_t>(kArcNone));
          idx++;
        }
      }
    }
  }
  if (tid == 0) {
    const long long fi = static_cast<long long>(c.utt) * (p.max_frames + 1) + (frame + 1);
    ToG(p.frame_base)[fi] = base_out;
    ToG(p.frame_count)[fi] = n_out;
  }
  __syncthreads();
  total_cnt.tokens_expanded += cnt.tokens_expanded;
  total_cnt.arcs_scanned += cnt.arcs_scanned;
  total_cnt.arcs_kept += cnt.arcs_kept;
  total_cnt.eps_arcs += cnt.eps_arcs;
  total_cnt.eps_exp += cnt.eps_exp;
  n_out_ret = n_out;
  return 0;
}

__device__ __forceinline__ long long BlockSumLL(Shared &s, long long v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) s.red64[wave] = static_cast<unsigned long long>(v);
  __syncthreads();
  long long r = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) r += static_cast<long long>(s.red64[w]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kBlock) void SearchKernel(const SearchParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Shared &s = *reinterpret_cast<Shared *>(smem);
  const int tid = threadIdx.x, utt = blockIdx.x;
  const long long gslots = 1ll << p.g_log2_slots;

  Table<true> tl{s, nullptr, nullptr, nullptr, ToG(p.g_src) + utt * gslots, nullptr, nullptr,
                 s.bitmask, kLog2Slots, kSlots - 1};
  Table<false> tg{s, ToG(p.g_key) + utt * gslots, ToG(p.g_cost) + utt * gslots,
                  ToG(p.g_arc) + utt * gslots, ToG(p.g_src) + utt * gslots,
                  ToG(p.g_idx) + utt * gslots, ToG(p.g_used) + utt * gslots,
                  reinterpret_cast<uint32_t *>(s.cost), p.g_log2_slots,
                  static_cast<uint32_t>(gslots - 1)};
  FrameCtx c{p, s, utt};
  c.wl[0] = ToG(p.g_wl) + (2ll * utt) * gslots;
  c.wl[1] = ToG(p.g_wl) + (2ll * utt + 1) * gslots;
  c.ll_in_lds = p.num_cols <= kVmax;

  if (tid == 0) s.stop = 0;
  ClearTable<true>(tl, s, 0);

  UttDesc ud;
  ud.ll = ToG(p.utt)[utt].ll;
  ud.offset = ToG(p.utt)[utt].offset;
  ud.target = ToG(p.utt)[utt].target;
  Counters cnt;
  long long st_frames = 0, st_tokens_in = 0, st_tokens_out = 0, st_max = 0, st_spill = 0;
  uint32_t err = ToG(p.status)[utt];
  int frame = p.do_init ? -1 : ToG(p.frames_decoded)[utt];
  const int target = p.do_init ? 0 : ud.target;

  while (err == 0 && frame < target) {
    if (frame >= 0) {
      const long long fi = static_cast<long long>(utt) * (p.max_frames + 1) + frame;
      c.base_in = ToG(p.frame_base)[fi];
      c.n_in = ToG(p.frame_count)[fi];
      c.ll_row = ToG(ud.ll) + static_cast<long long>(frame - ud.offset) * p.row_stride;
      if (c.ll_in_lds)
        for (int v = tid; v < p.num_cols; v += kBlock) s.ll[v] = c.ll_row[v];
      __syncthreads();
    } else {
      c.base_in = 0;
      c.n_in = 0;
    }
    int n_out = 0;
    uint32_t r = 1;
    // a frontier this large cannot fit the LDS table after expansion
    if (c.n_in <= kLdsFillLimit) r = RunFrame<true>(c, tl, frame, cnt, n_out);
    if (r == 1) {
      __syncthreads();
      ClearTable<true>(tl, s, 0);  // also resets fill/overflow; s.cost becomes the spill bitmask
      r = RunFrame<false>(c, tg, frame, cnt, n_out);
      const int used = s.fill < static_cast<int>(gslots) ? s.fill : static_cast<int>(gslots);
      __syncthreads();
      ClearTable<false>(tg, s, used);
      ClearTable<true>(tl, s, 0);
      st_spill++;
      if (r == 1) r = KD_UTT_TABLE_FULL;
    } else {
      ClearTable<true>(tl, s, 0);
    }
    if (r != 0) { err |= r; break; }
    if (frame >= 0) {
      st_frames++;
      st_tokens_in += c.n_in;
    }
    st_tokens_out += n_out;
    st_max = n_out > st_max ? n_out : st_max;
    frame++;
  }

  const long long exp_sum = BlockSumLL(s, cnt.tokens_expanded);
  const long long kept_sum = BlockSumLL(s, cnt.arcs_kept);
  const long long eps_exp_sum = BlockSumLL(s, cnt.eps_exp);
  if (tid == 0) {
    ToG(p.frames_decoded)[utt] = frame < 0 ? -1 : frame;
    ToG(p.status)[utt] = err;
    KD_G long long *st = ToG(p.stats) + static_cast<long long>(utt) * KD_NUM_STATS;
    st[KD_STAT_FRAMES] += st_frames;
    st[KD_STAT_TOKENS_IN] += st_tokens_in;
    st[KD_STAT_TOKENS_EXPANDED] += exp_sum;
    st[KD_STAT_ARCS_SCANNED] += cnt.arcs_scanned;
    st[KD_STAT_ARCS_KEPT] += kept_sum;
    st[KD_STAT_TOKENS_OUT] += st_tokens_out;
    st[KD_STAT_EPS_ARCS_SCANNED] += cnt.eps_arcs;
    st[KD_STAT_EPS_EXPANSIONS] += eps_exp_sum;
    st[KD_STAT_MAX_TOKENS] = st_max > st[KD_STAT_MAX_TOKENS] ? st_max : st[KD_STAT_MAX_TOKENS];
    st[KD_STAT_SPILL_FRAMES] += st_spill;
  }
}

}  // namespace

size_t SearchLdsBytes() { return sizeof(Shared); }

hipError_t LaunchSearch(const SearchParams &p, int batch, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(SearchKernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(sizeof(Shared)));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(SearchKernel, dim3(batch), dim3(kBlock), sizeof(Shared), stream, p);
  return hipGetLastError();
}

}  // namespace kd
