// kaldi-decoder_amd/csrc/kd_lattice_tail.h -- lattice pruning and raw-lattice export
// (ref: /root/reference/kaldi-decoder/csrc/lattice-simple-decoder.cc), run by the search
// kernel's workgroup in the tail of an utterance.  Part of kd_search.hip (included inside
// its namespace, after the LDS block, ToG / KD_G, CParams, LdsBarrier and kBlock).
//
//   FinalCosts   ComputeFinalCosts (:522-560), FinalRelativeCost (:562-572)
//   Prune        PruneForwardLinksFinal (:425-520) and the backward sweep of
//                FinalizeDecoding (:407-420) = PruneForwardLinks with delta 0 (:228-305) +
//                PruneTokensForFrame (:310-334) for every frame; PruneActiveTokens
//                (:198-223), the same sweep between two frames of the search; small
//                frames are swept a window at a time (SweepWindow)
//   Collect      what PruneActiveTokens leaves of old frames moves to the archive
//   Export       GetRawLattice (:584-657): sizes, state numbering, then states, final
//                weights and arcs into ONE contiguous region of the batch's output pool
//
// Why here and not in kernels of their own (round 1): the workgroup that finishes an
// utterance already owns a CU and 160 KiB of LDS; separate pruning / export launches
// could not start while search workgroups held all the LDS, cost two extra launches, two
// host round trips and a prefix sum on the host per batch.
//
// A frame's links are processed link-parallel: link_extra = next.extra + ((tok.tot + ac +
// graph) - next.tot), links above lattice_beam are excised (dst = -1), the survivors
// min-reduce into their source token's extra cost with ds_min_u32 on an order-preserving
// float key held in LDS (one word per token of the frame).  Epsilon links point inside
// the frame, so the frame is iterated to its fixpoint (frames swept one by one: from below,
// extras start at 0 and only grow, which makes early excision safe; windows of small frames:
// from above, links excised when the keys are final -- SweepWindow says why both give the
// same bits).
//
// State ids are frame-major like the reference's.  Inside a frame, and for arcs, the order
// is the arena order in the general export (ranks: ballot + popcount, a prefix sum over the
// frames) and the order in which the sweep listed the survivors in the staged export
// (atomic list positions, an atomic counter per frame): lattices are equal as sets from
// run to run, not as arrays.

// (epsilon links of a frame a lane keeps in registers on the frame-by-frame LDS path; one
// is enough for frames of up to 768 epsilon links -- the bench workload's largest has 282 --
// and leaves the windows more registers: 4 measured + 5 % on the sweeps' cycles)
#ifndef KD_LEPS_REGS
#define KD_LEPS_REGS 1
#endif
// Extra-cost keys of one frame kept in LDS (144 KiB of 160 with one workgroup per CU): the
// part of the search's LDS block below the frontier's state array.  The periodic sweep
// runs BETWEEN two frames of the search, when the block already holds the next frame's
// log-likelihood row, the frame counters and the launch's statistics: the keys end where
// `it_src` begins, and the few words behind them stay below `ll` (asserted).
constexpr int kLdsToks = static_cast<int>(offsetof(Shared, it_src) / 4) / 2048 * 2048;
constexpr uint32_t kFInfKey = 0xFF800000u;  // FloatKey(+inf)
constexpr int kEpsRegs = KD_LEPS_REGS;  // epsilon links of a frame a lane keeps in registers

// A WINDOW of the backward sweep: consecutive frames whose links fit the lanes' registers
// (kWinRegs each) and whose tokens fit half the key array, swept together (SweepWindow).
// (link slots per lane: every round walks all of a lane's slots, so a window of fewer,
// cheaper rounds beats a larger one -- measured on the bench workload at the default
// prune_interval: 8 slots 10.5 k cycles per utterance-frame of periodic pruning, 4: 8.7 k,
// 3: 8.5 k, 2: 8.4 k, 1: 9.0 k, 12: 12.0 k)
#ifndef KD_WIN_REGS
#define KD_WIN_REGS 3
#endif
#ifndef KD_WIN_MAX
#define KD_WIN_MAX 64
#endif
constexpr int kWinMax = KD_WIN_MAX;  // frames per window, at most (one lane of a wave reads a frame's descriptor)
static_assert(kWinMax >= 1 && kWinMax <= 64, "a lane of one wave per frame of the window");
constexpr int kWinRegs = KD_WIN_REGS;
constexpr int kWinLinks = kWinRegs * kBlock;
constexpr int kWinToks = kLdsToks / 2;
struct WinShared {
  // frame f - j of the window that starts at frame f (all kWinMax read, the first W used)
  long long base[kWinMax], eb[kWinMax], pb[kWinMax];
  int32_t n[kWinMax], en[kWinMax], pn[kWinMax];
  int32_t tok_off[kWinMax + 1];      // where the frame's tokens start in the window's key array
  int32_t seg_off[2 * kWinMax + 1];  // ... and its emitting (2j) and epsilon (2j + 1) links in link order
  int32_t fchanged[kWinMax];         // periodic: an extra cost of the frame moved by more than delta
  int32_t W, flag[3];
};

struct TailShared {
  uint32_t ne[kLdsToks];
  WinShared win;
  int32_t red_i[kWaves];
  uint32_t best_key, bestwf_key;
  int32_t any_final, changed, count, start_id, empty_frame, frame_changed;
  uint32_t alloc_err;
  int32_t n_finals;
  int32_t n_tok_staged, n_arc_staged, stage_overflow;
  long long out_off;
  // results of the tail, read by the caller (RunUtterance) from here
  int32_t r_sizes[4];
  long long r_out_off;
  float r_final_rel;
  int32_t r_arena_order;  // the export is the general one's (arena order, four words an arc)
};
static_assert(sizeof(TailShared) <= offsetof(Shared, ll),
              "the tail reuses the search's LDS block, below what a periodic sweep must leave alone");
__device__ __forceinline__ TailShared &TailLds() { return *reinterpret_cast<TailShared *>(kd_smem); }

__device__ __forceinline__ float FInf() { return __int_as_float(0x7F800000); }

// Exclusive prefix sum of one int per thread over the workgroup; `total` = the sum.
__device__ __forceinline__ int BlockExclScanI(TailShared &s, int v, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = WaveInclusiveSumInt(v);
  if (lane == 63) s.red_i[wave] = x;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < kWaves ? s.red_i[lane] : 0, xs = WaveInclusiveSumInt(w);
    if (lane < kWaves) s.red_i[lane] = xs - w;
    if (lane == kWaves - 1) s.count = xs;
  }
  __syncthreads();
  const int off = s.red_i[wave] + x - v;
  total = s.count;
  __syncthreads();
  return off;
}
static_assert(kWaves <= 16, "BlockExclScanI scans at most 16 wave totals");

// Applies one link block to the frame's new-extra keys (frames that do not fit the
// LDS-resident path: thousands of tokens and links).  Four links per lane at a time:
// their records, then the four destination extras (a gather), are requested together
// (every lane issues every load -- a lane without a link reads a valid dummy --
// because hipcc waits for a conditional load before it issues the next).
__device__ __forceinline__ void ApplyLinks(CParams &p, TailShared &s, KD_G uint32_t *ne,
                                           bool ne_in_lds, long long lb, int ln, long long base) {
  constexpr int U = 4;
  const float lattice_beam = p.lat.lattice_beam;
  for (int i0 = threadIdx.x; i0 < ln; i0 += U * kBlock) {
    int32_t dst[U], src[U];
    float delta[U], next_extra[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kBlock;
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
      if (link_extra_cost > lattice_beam) {  // :264 excise
        ToG(p.link_dst)[lb + i0 + u * kBlock] = -1;
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

// ComputeFinalCosts (:522-560) over the last frame's tokens.
__device__ __forceinline__ void LatticeFinalCosts(CParams &p, TailShared &s, int utt, int F,
                                                  float &final_best_cost, bool &any_final_out,
                                                  float &final_rel) {
  const int tid = threadIdx.x;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  const long long baseF = ToG(p.frame_base)[fb + F];
  const int nF = ToG(p.frame_count)[fb + F];
  if (tid == 0) { s.best_key = kFInfKey; s.bestwf_key = kFInfKey; s.any_final = 0; }
  __syncthreads();
  {
    uint32_t bk = kFInfKey, bwk = kFInfKey;
    int af = 0;
    for (int i = tid; i < nF; i += kBlock) {
      const float cost = static_cast<float>(ToG(p.tok_cost)[baseF + i]);
      const float final_cost = ToG(p.states)[ToG(p.tok_state)[baseF + i]].final_cost;
      const float cost_with_final = cost + final_cost;
      const uint32_t k1 = FloatKey(cost), k2 = FloatKey(cost_with_final);
      bk = k1 < bk ? k1 : bk;
      bwk = k2 < bwk ? k2 : bwk;
      af |= final_cost != FInf();
    }
    if (bk != kFInfKey) atomicMin(&s.best_key, bk);
    if (bwk != kFInfKey) atomicMin(&s.bestwf_key, bwk);
    if (af) s.any_final = 1;
  }
  __syncthreads();
  const float best_cost = KeyFloat(s.best_key), best_wf = KeyFloat(s.bestwf_key);
  any_final_out = s.any_final != 0;
  final_best_cost = best_wf != FInf() ? best_wf : best_cost;  // :553-559
  final_rel = (best_cost == FInf() && best_wf == FInf()) ? FInf() : best_wf - best_cost;  // :544-552
  __syncthreads();
}

// What survives FinalizeDecoding, noted down by the sweep itself while a frame's extras
// and links are at hand (in LDS / registers): live tokens as (token index, frame), links
// as (link index, frame * 2 + emitting).  GetRawLattice then costs what the PRUNED
// lattice costs -- a thousand arcs -- not another three passes over the hundred thousand
// tokens and links the search recorded.  The lists live in the workgroup's candidate
// queue (idle in the tail); a lattice that outgrows them takes the general export.
struct Stage {
  KD_G uint32_t *tok_g, *tok_f, *arc_at, *arc_f;
  int cap;
  bool on;
};
__device__ __forceinline__ Stage TailStage(CParams &p, bool on) {
  const long long gslots = 1ll << p.g_log2_slots, wg = p.sched.wg_base + blockIdx.x;
  Stage st;
  st.tok_g = ToG(p.g_log_h) + wg * 2 * gslots;
  st.tok_f = ToG(p.g_log_arc) + wg * 2 * gslots;
  st.arc_at = ToG(p.g_log_src) + wg * 2 * gslots;
  st.arc_f = reinterpret_cast<KD_G uint32_t *>(ToG(p.g_log_nk) + wg * 2 * gslots);
  st.cap = static_cast<int>(2 * gslots);
  st.on = on;
  return st;
}
// one list entry per active lane, one LDS atomic per wave
__device__ __forceinline__ void StageTok(TailShared &s, const Stage &st, long long g, int f) {
  const int pos = AggInc(&s.n_tok_staged);
  if (pos < st.cap) { st.tok_g[pos] = static_cast<uint32_t>(g); st.tok_f[pos] = static_cast<uint32_t>(f); }
  else s.stage_overflow = 1;
}
__device__ __forceinline__ void StageArc(TailShared &s, const Stage &st, long long at, int f, int emitting) {
  const int pos = AggInc(&s.n_arc_staged);
  if (pos < st.cap) { st.arc_at[pos] = static_cast<uint32_t>(at); st.arc_f[pos] = static_cast<uint32_t>(2 * f + emitting); }
  else s.stage_overflow = 1;
}

// |a - b| > delta for extra costs (+inf == +inf: not moved; a token that died, or came
// back, has moved)
__device__ __forceinline__ bool Moved(float a, float b, float delta) {
  if (a == b) return false;
  return !(fabsf(a - b) <= delta);
}

// The sweep over a window of small frames.  A frame of the median size (68 links) costs
// what its chain of dependent memory round trips and barriers costs, whatever it holds;
// frame by frame that is 7-10 k cycles each.  Here the link records of up to kWinMax
// consecutive frames are fetched at once into registers (source and destination as
// positions in the window's key array) and the whole window is relaxed in LDS:
//   B[token] = min over its surviving links of max(B[destination] + delta, 0)
// starting from +inf (from ABOVE), in place, every link of the window applied each
// round until a round lowers nothing.  Links whose destination lies above the window
// (the emitting links of its top frame) are applied once, from final values.  The
// links form a DAG -- emitting links lead one frame up, epsilon links follow the graph's
// acyclic epsilon arcs -- so the fixpoint is unique and equals what the frame-by-frame
// sweep computes: a key only ever holds a candidate that a real chain of links
// produces, never less than the true value, and when a round changes nothing every key
// is the minimum over the links at their destinations' final values.  A link is applied
// only while its cost is within lattice_beam (its cost only falls from round to round),
// and excised (:264) after the last round if it still is not.  Lanes walk the links in
// frame order (top frame first), so most windows settle in a few rounds.
// Returns the number of frames swept (>= 1: the caller has checked that frame f fits).
__device__ __forceinline__ bool WinFits(int n, int en, int pn) {
  return n <= kWinToks && en + pn <= kWinLinks;
}

__device__ __forceinline__ int SweepWindow(CParams &p, TailShared &s, const Stage &st, long long fb,
                                           int f, bool periodic, int visited, float delta,
                                           const uint32_t *&next_b, long long &next_base, int &flip,
                                           bool &stop) {
  const int tid = threadIdx.x, lane = tid & 63;
  const float lattice_beam = p.lat.lattice_beam;
  WinShared &w = s.win;
  // (The previous window's descriptors are read once more AFTER it returned -- LatticePrune's
  // from_window() takes the next frames' descriptors from them -- by every wave for itself.
  // Without this barrier the first wave could be back here, overwriting them, while a wave
  // that lagged a few thousand cycles was still reading: that wave then saw another frame's
  // sizes, took another branch than the rest of the workgroup and met them at the wrong
  // barrier -- a launch that never ends, or indices into nowhere.  Rare (the builder's own
  // global loads come first), and sensitive to anything that shifts the waves' timing: the
  // stall / memory fault of round 2's hlg-large runs with another register allocation, and
  // one bench run in six of round 3 after an unrelated edit of this function.)
#ifndef KD_TEST_DROP_SWEEP_BARRIER  // (tools only: the jitter build must FIND the race this barrier closes)
  LdsBarrier();
#endif
  if (tid < 64) {  // the descriptors of frames f, f - 1, ...; how many fit
    const int k = f - lane;
    const bool in = lane < kWinMax && k >= 0;
    const long long at = fb + (in ? k : f);
    const long long base = ToG(p.frame_base)[at], eb = ToG(p.link_emit_base)[at], pb = ToG(p.link_eps_base)[at];
    int n = ToG(p.frame_count)[at], en = ToG(p.link_emit_count)[at], pn = ToG(p.link_eps_count)[at];
    if (!in) { n = 0; en = 0; pn = 0; }
    const int ct = WaveInclusiveSumInt(n), cl = WaveInclusiveSumInt(en + pn);
    const unsigned long long m = __ballot(in && ct <= kWinToks && cl <= kWinLinks);
    const int W = m == ~0ull ? 64 : __builtin_ctzll(~m);
    if (lane < kWinMax) {
      w.base[lane] = base; w.eb[lane] = eb; w.pb[lane] = pb;
      w.n[lane] = n; w.en[lane] = en; w.pn[lane] = pn;
      w.tok_off[lane] = ct - n;
      w.seg_off[2 * lane] = cl - en - pn;
      w.seg_off[2 * lane + 1] = cl - pn;
      w.fchanged[lane] = 0;
    }
    if (lane == kWinMax - 1) { w.tok_off[kWinMax] = ct; w.seg_off[2 * kWinMax] = cl; }
    if (lane == 0) { w.W = W; w.flag[0] = 0; w.flag[1] = 0; w.flag[2] = 0; }
  }
  __syncthreads();
  const int W = w.W, NT = w.tok_off[W], NL = w.seg_off[2 * W];
  uint32_t *B = s.ne + flip * (kLdsToks / 2);
  flip ^= 1;
  for (int i = tid; i < NT; i += kBlock) B[i] = kFInfKey;
  // this lane's links: link g = tid + r * kBlock of the window's link order
  constexpr int kDead = -1, kSrcBits = 16, kEmitBit = 1 << 24, kDoneBit = 1 << 25;
  static_assert(kWinToks < (1 << kSrcBits) && kWinMax <= 256, "link slot packing");
  uint32_t at[kWinRegs];
  int meta[kWinRegs];  // source position | frame index << 16 | emitting | applied for good
  int32_t dst[kWinRegs], src[kWinRegs];
  float dl[kWinRegs];
#pragma unroll
  for (int r = 0; r < kWinRegs; ++r) {
    const int g = tid + r * kBlock;
    if (g < NL) {
      int lo = 0, hi = 2 * W;  // seg_off[lo] <= g < seg_off[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (w.seg_off[mid] <= g) lo = mid; else hi = mid;
      }
      const int j = lo >> 1;
      at[r] = static_cast<uint32_t>(((lo & 1) ? w.pb[j] : w.eb[j]) + (g - w.seg_off[lo]));
      meta[r] = j << kSrcBits | ((lo & 1) ? 0 : kEmitBit);
    } else {
      at[r] = 0;  // (every lane issues every load: a lane without a link reads the pool's first record)
      meta[r] = kDead;
    }
  }
#pragma unroll
  for (int r = 0; r < kWinRegs; ++r) {
    dst[r] = ToG(p.link_dst)[at[r]];
    src[r] = ToG(p.link_src)[at[r]];
    dl[r] = ToG(p.link_delta)[at[r]];
  }
  // the extras above the window, when they are not in LDS: a second gather
  const bool above_global = next_b == nullptr;
  float above[kWinRegs];
  if (above_global) {
    const long long safe = w.base[0];
#pragma unroll
    for (int r = 0; r < kWinRegs; ++r) {
      const bool top = meta[r] != kDead && (meta[r] >> kSrcBits) == (kEmitBit >> kSrcBits) && dst[r] >= 0;
      above[r] = ToG(p.tok_extra)[top ? dst[r] : safe];
    }
  }
  LdsBarrier();  // B is +inf everywhere
#pragma unroll
  for (int r = 0; r < kWinRegs; ++r) {
    if (meta[r] == kDead) continue;
    if (dst[r] < 0) { meta[r] = kDead; continue; }  // excised by an earlier sweep
    const int j = (meta[r] >> kSrcBits) & 0xFF;
    const bool emitting = meta[r] & kEmitBit;
    const int srcl = w.tok_off[j] + static_cast<int>(src[r] - w.base[j]);
    meta[r] |= srcl;
    if (!emitting) {
      dst[r] = w.tok_off[j] + static_cast<int>(dst[r] - w.base[j]);
    } else if (j > 0) {
      dst[r] = w.tok_off[j - 1] + static_cast<int>(dst[r] - w.base[j - 1]);
    } else {  // the destination's extra is final: this link is decided here
      const float next_extra = above_global ? above[r] : KeyFloat(next_b[dst[r] - next_base]);
      float link_extra_cost = next_extra + dl[r];  // :257-260
      if (link_extra_cost > lattice_beam) {  // :264 excise
        ToG(p.link_dst)[at[r]] = -1;
        meta[r] = kDead;
        continue;
      }
      if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
      atomicMin(&B[srcl], FloatKey(link_extra_cost));
      meta[r] |= kDoneBit;
    }
  }
  LdsBarrier();
  for (int round = 0;; ++round) {
    int ch = 0;
    // (a slot whose destination is still +inf, or dead, falls through in a few
    // instructions -- a wave all of whose lanes do skips the slot -- so a round costs
    // what the frames around the front of the propagation cost, not the whole window)
#pragma unroll
    for (int r = 0; r < kWinRegs; ++r) {
      if (meta[r] == kDead || (meta[r] & kDoneBit)) continue;
      float link_extra_cost = KeyFloat(B[dst[r]]) + dl[r];  // :257-260
      if (!(link_extra_cost <= lattice_beam)) continue;  // (not yet, or never: decided below)
      if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
      const uint32_t k = FloatKey(link_extra_cost);
      if (atomicMin(&B[meta[r] & ((1 << kSrcBits) - 1)], k) > k) ch = 1;
    }
    if (ch) w.flag[round % 3] = 1;
    if (tid == 0) w.flag[(round + 1) % 3] = 0;
    LdsBarrier();
    if (!w.flag[round % 3]) break;
  }
  // Every key is final.  PERIODIC sweeps: how far down does the reference go?  Frame g that
  // an earlier sweep has visited (g <= visited) is swept again only if an extra cost of
  // frame g + 1 moved by more than delta in THIS sweep (must_prune_forward_links,
  // lattice-simple-decoder.cc:208-216, :286-290; a frame never swept before started from
  // the zeros the search wrote).  The window was relaxed as a whole, which is right for
  // its frames down to the first one that rule excludes -- values only travel downwards --
  // so that frame and the ones below it are left exactly as they were: `cut` frames count.
  int cut = W;
  if (periodic && f - (W - 1) <= visited + 1) {
    for (int g0 = tid; g0 < NT; g0 += 4 * kBlock) {
      int j[4];
      float was[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = g0 + u * kBlock < NT ? g0 + u * kBlock : g0;
        int lo = 0, hi = W;  // tok_off[lo] <= g < tok_off[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (w.tok_off[mid] <= g) lo = mid; else hi = mid;
        }
        j[u] = lo;
        // (only the frames whose verdict decides about a visited frame below them)
        was[u] = f - lo <= visited + 1 ? ToG(p.tok_extra)[w.base[lo] + (g - w.tok_off[lo])] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = g0 + u * kBlock;
        if (g < NT && f - j[u] <= visited + 1 && Moved(was[u], KeyFloat(B[g]), delta)) w.fchanged[j[u]] = 1;
      }
    }
    LdsBarrier();  // the frames' flags are complete
    const bool excluded = lane >= 1 && lane < W && f - lane <= visited && !w.fchanged[lane - 1];
    const unsigned long long ex = __ballot(excluded);  // (the same answer in every wave)
    if (ex) cut = __builtin_ctzll(ex);
  }
  // the links that stay above the beam go (:264)
#pragma unroll
  for (int r = 0; r < kWinRegs; ++r) {
    if (meta[r] == kDead) continue;
    if (((meta[r] >> kSrcBits) & 0xFF) >= cut) { meta[r] = kDead; continue; }  // (frame not swept)
    if (meta[r] & kDoneBit) continue;
    const float link_extra_cost = KeyFloat(B[dst[r]]) + dl[r];
    if (!(link_extra_cost <= lattice_beam)) {
      ToG(p.link_dst)[at[r]] = -1;
      meta[r] = kDead;
    }
  }
  // the tokens' extras go to the pool (a token without a surviving link: +inf, which is
  // how PruneTokensForFrame's verdict is recorded here)
  const int NTc = w.tok_off[cut];
  for (int g0 = tid; g0 < NTc; g0 += 4 * kBlock) {
    int j[4];
    long long gi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + u * kBlock < NTc ? g0 + u * kBlock : g0;
      int lo = 0, hi = W;  // tok_off[lo] <= g < tok_off[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (w.tok_off[mid] <= g) lo = mid; else hi = mid;
      }
      j[u] = lo;
      gi[u] = w.base[lo] + (g - w.tok_off[lo]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + u * kBlock;
      if (g >= NTc) continue;
      const float ex = KeyFloat(B[g]);
      ToG(p.tok_extra)[gi[u]] = ex;
      if (st.on && ex != FInf()) StageTok(s, st, gi[u], f - j[u]);
    }
  }
  if (st.on) {
#pragma unroll
    for (int r = 0; r < kWinRegs; ++r)
      if (meta[r] != kDead) StageArc(s, st, at[r], f - ((meta[r] >> kSrcBits) & 0xFF), (meta[r] & kEmitBit) ? 1 : 0);
    if (tid < W && w.n[tid] == 0) s.empty_frame = 1;
  }
  next_b = B + w.tok_off[cut - 1];
  next_base = w.base[cut - 1];
  // the sweep ends here if a frame of the window was excluded, or if the frame below the
  // window is one that the rule excludes
  stop = periodic && (cut < W || (f - W <= visited && !w.fchanged[W - 1]));
  // (the next window's builder, or a frame that reads these extras from global memory,
  // starts with a full barrier of its own, after which w and B's other half are reused)
  LdsBarrier();
  return cut;
}

// FinalizeDecoding: the exact backward sweep over all frames -- and, with `periodic`,
// PruneActiveTokens (:198-223), the same sweep started one frame below the newest (whose
// tokens keep extra_cost 0: every token still active is taken to be on a best path), with
// no final costs, run until a frame that an earlier sweep has already visited
// (f <= `visited`) comes out with no extra cost moved by more than `delta` = lattice_beam
// x prune_scale (the reference's must_prune_forward_links chain, :208-216, and its
// tolerance, :286-288).  What a periodic sweep excises the final one would excise too:
// extras only grow while the utterance does, so a stale extra under-estimates a link's
// cost and the final, exact sweep recomputes every frame from below -- the finalized
// lattice does not depend on how often or how far the periodic sweeps went.  Returns
// the lowest frame swept.
__device__ __forceinline__ int LatticePrune(CParams &p, TailShared &s, int utt, int F,
                                            float final_best_cost, bool any_final, const Stage &st,
                                            bool periodic = false, int visited = -1, float delta = 0.0f) {
  const int tid = threadIdx.x;
  if (tid == 0) { s.n_tok_staged = 0; s.n_arc_staged = 0; s.stage_overflow = 0; s.empty_frame = 0; }
  // (periodic: F is the newest frame; the sweep starts at F - 1, whose emitting links end
  // in F's tokens -- extra_cost 0.0f as the search wrote them, read from global memory)
  const int f_top = periodic ? F - 1 : F;
  const int f_last = periodic ? -1 : F;  // the frame that takes the final-cost term
  int f_low = f_top + 1;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  const float lattice_beam = p.lat.lattice_beam;
  const uint32_t *next_b = nullptr;  // the next frame's final extras, while they are in LDS
  long long next_base = 0;
  // frame descriptors are fetched one frame ahead
  struct Desc { long long base, eb, pb; int n, en, pn; };
  auto describe = [&](int f) {
    Desc d;
    d.base = ToG(p.frame_base)[fb + f];
    d.n = ToG(p.frame_count)[fb + f];
    d.eb = f == f_last ? 0 : ToG(p.link_emit_base)[fb + f];
    d.en = f == f_last ? 0 : ToG(p.link_emit_count)[fb + f];
    d.pb = ToG(p.link_eps_base)[fb + f];
    d.pn = ToG(p.link_eps_count)[fb + f];
    return d;
  };
  // The link records an LDS-resident frame starts from: this lane's epsilon links
  // (source, destination, the cost difference of :257-260) and its first emitting link.
  // They are requested one frame early -- while frame f iterates to its fixpoint the
  // records of frame f - 1 are already on their way (nothing frame f does changes them),
  // so a frame no longer begins with an exposed memory round trip; the descriptors
  // that say where those records are come two frames ahead for the same reason.
  // (Every lane issues every load -- a lane without a link reads the pool's first
  // record -- so they are in flight together; loads behind `if (i < pn)` complete one
  // after the other.)
  struct Pre {
    int e_dst[kEpsRegs], e_src[kEpsRegs];
    float e_delta[kEpsRegs];
    int m_dst0, m_src0;
    float m_delta0;
  };
  auto fits_lds = [&](const Desc &d) { return 3 * d.n <= kLdsToks / 2 && d.pn <= kEpsRegs * kBlock; };
  auto fetch = [&](const Desc &d, Pre &r) {
#pragma unroll
    for (int k = 0; k < kEpsRegs; ++k) {
      const int i = tid + k * kBlock;
      const long long at = i < d.pn ? d.pb + i : 0;
      r.e_dst[k] = ToG(p.link_dst)[at];
      r.e_src[k] = ToG(p.link_src)[at];
      r.e_delta[k] = ToG(p.link_delta)[at];
    }
    const long long at0 = tid < d.en ? d.eb + tid : 0;
    r.m_dst0 = ToG(p.link_dst)[at0];
    r.m_src0 = ToG(p.link_src)[at0];
    r.m_delta0 = ToG(p.link_delta)[at0];
  };
  if (f_top < 0) return 0;
  // small frames are swept a window at a time (SweepWindow); the last frame (its tokens
  // carry the final-cost term) and frames with more links than the lanes' registers
  // hold go one by one, below
  auto win_ok = [&](const Desc &d, int f) { return f != f_last && WinFits(d.n, d.en, d.pn); };
  auto from_window = [&](int j) {
    Desc d;
    d.base = s.win.base[j]; d.n = s.win.n[j];
    d.eb = s.win.eb[j]; d.en = s.win.en[j];
    d.pb = s.win.pb[j]; d.pn = s.win.pn[j];
    return d;
  };
  // (a descriptor is the same in every lane: once it has arrived it is kept in scalar
  // registers, which leaves the windows nine more vector registers)
  auto uniform = [&](const Desc &v) {
    Desc d;
    d.base = Uni(v.base); d.eb = Uni(v.eb); d.pb = Uni(v.pb);
    d.n = Uni(v.n); d.en = Uni(v.en); d.pn = Uni(v.pn);
    return d;
  };
  int flip = 0;  // which half of the key array the next LDS-resident frame or window takes
#if defined(KD_PROFILE) && defined(KD_PROF_TAIL)
  // diagnostic build `prof_tail`: cycles in windows (bucket 16) and in frames swept one by
  // one (18) -- a BREAKDOWN of time that buckets 21 (FinalizeDecoding) and 23 (periodic
  // sweep + collector) already hold, so this build's bucket sum counts it twice
  long long prof_lap = static_cast<long long>(__builtin_amdgcn_s_memtime()), prof_cyc[2] = {0, 0};
  auto prof = [&](int which) {
    const long long now = static_cast<long long>(__builtin_amdgcn_s_memtime());
    prof_cyc[which] += now - prof_lap;
    prof_lap = now;
  };
  auto prof_flush = [&]() {
    if (tid != 0) return;
    if (periodic) {  // (flushed with the search's own buckets when the utterance ends)
      Lds().cyc[16] += prof_cyc[0];
      Lds().cyc[18] += prof_cyc[1];
    } else {
      ToG(p.stats)[static_cast<long long>(utt) * kStatStride + KD_NUM_STATS + 16] += prof_cyc[0];
      ToG(p.stats)[static_cast<long long>(utt) * kStatStride + KD_NUM_STATS + 18] += prof_cyc[1];
    }
  };
#else
  auto prof = [](int) {};
  auto prof_flush = []() {};
#endif
  Desc ahead = uniform(describe(f_top)), ahead2 = f_top > 0 ? describe(f_top - 1) : ahead;
  Pre pre;
  bool have_pre = false;
  if (!win_ok(ahead, f_top) && fits_lds(ahead)) { fetch(ahead, pre); have_pre = true; }
  for (int f = f_top; f >= 0; --f) {
    if (win_ok(ahead, f)) {
      bool stop;
      const int W = SweepWindow(p, s, st, fb, f, periodic, visited, delta, next_b, next_base, flip, stop);
      f_low = f - W + 1;
      prof(0);
      if (stop) break;
      f = f_low;  // (the loop steps to the frame below the window)
      if (f > 0) {  // the builder read the descriptors of kWinMax frames
        ahead = uniform(W < kWinMax ? from_window(W) : describe(f - 1));
        ahead2 = f > 1 ? (W + 1 < kWinMax ? from_window(W + 1) : describe(f - 2)) : ahead;
        have_pre = !win_ok(ahead, f - 1) && fits_lds(ahead);
        if (have_pre) fetch(ahead, pre);
      }
      continue;
    }
    f_low = f;
    if (tid == 0) s.frame_changed = 0;  // (read behind the barriers at the frame's end)
    const Desc d = ahead;
    ahead = uniform(ahead2);  // (requested a frame ago)
    if (f > 1) ahead2 = describe(f - 2);
    const long long base = d.base;
    const int n = d.n;
    const bool last = f == f_last;
    const long long eb = d.eb;
    const int en = d.en;
    const long long pb = d.pb;
    const int pn = d.pn;
    if (3 * n <= kLdsToks / 2 && pn <= kEpsRegs * kBlock) {
      // The frame's fixpoint entirely in LDS.  Emitting links end in the next
      // frame, whose extras are final: their contribution (A) is computed once.
      // Epsilon links are fetched once into registers; each round recomputes
      // C = min(A, epsilon links over the current estimate B) and B := C, as the
      // loop below does through global memory.  Frames alternate between the two
      // halves of the key array (`flip`), so the next frame's extras are still there.
      uint32_t *half = s.ne + flip * (kLdsToks / 2);
      flip ^= 1;
      uint32_t *A = half, *B = half + n, *C = half + 2 * n;
      for (int i = tid; i < n; i += kBlock) {
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
      // this frame's link records (requested a frame ago) ...
      if (!have_pre) fetch(d, pre);
      int e_src[kEpsRegs], e_dst[kEpsRegs];
      float e_delta[kEpsRegs];
#pragma unroll
      for (int r = 0; r < kEpsRegs; ++r) {
        e_dst[r] = pre.e_dst[r];
        e_src[r] = pre.e_src[r];
        e_delta[r] = pre.e_delta[r];
      }
      const int32_t m_dst0 = pre.m_dst0, m_src0 = pre.m_src0;
      const float m_delta0 = pre.m_delta0;
      // ... and the request for the next frame's (f - 1)
      have_pre = f > 0 && fits_lds(ahead) && !win_ok(ahead, f - 1);
      if (have_pre) fetch(ahead, pre);
#pragma unroll
      for (int r = 0; r < kEpsRegs; ++r) {
        const int i = tid + r * kBlock;
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
      // emitting links, into A: four per lane at a time, their records requested
      // together (the very first one came with the prefetch above)
      constexpr int kEmitU = 4;
      for (int i0 = tid; i0 < en; i0 += kEmitU * kBlock) {
        int32_t dst[kEmitU], src[kEmitU];
        float dl[kEmitU];
#pragma unroll
        for (int u = 0; u < kEmitU; ++u) {
          const int i = i0 + u * kBlock;
          const long long at = eb + (i < en ? i : i0);
          if (u == 0 && i0 == tid) {
            dst[u] = m_dst0; src[u] = m_src0; dl[u] = m_delta0;
          } else {
            dst[u] = ToG(p.link_dst)[at];
            src[u] = ToG(p.link_src)[at];
            dl[u] = ToG(p.link_delta)[at];
          }
          if (i >= en) dst[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < kEmitU; ++u) {
          if (dst[u] < 0) continue;  // already excised (or no link)
          const long long at = eb + i0 + u * kBlock;
          const float next_extra = next_b ? KeyFloat(next_b[dst[u] - next_base]) : ToG(p.tok_extra)[dst[u]];
          float link_extra_cost = next_extra + dl[u];  // :257-260
          if (link_extra_cost > lattice_beam) {  // :264 excise
            ToG(p.link_dst)[at] = -1;
            continue;
          }
          if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
          atomicMin(&A[src[u] - base], FloatKey(link_extra_cost));
          if (st.on) StageArc(s, st, at, f, 1);
        }
      }
      LdsBarrier();
      while (true) {
        for (int i = tid; i < n; i += kBlock) C[i] = A[i];
        if (tid == 0) s.changed = 0;
        LdsBarrier();
#pragma unroll
        for (int r = 0; r < kEpsRegs; ++r) {
          if (e_src[r] < 0) continue;
          float link_extra_cost = KeyFloat(B[e_dst[r]]) + e_delta[r];
          if (link_extra_cost > lattice_beam) {  // :264 excise
            ToG(p.link_dst)[pb + tid + r * kBlock] = -1;
            e_src[r] = -1;
            continue;
          }
          if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;  // :276-281
          atomicMin(&C[e_src[r]], FloatKey(link_extra_cost));
        }
        LdsBarrier();
        int ch = 0;
        for (int i = tid; i < n; i += kBlock) {
          float v = KeyFloat(C[i]);
          if (last && v > lattice_beam) v = FInf();  // :507-509
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
      for (int i = tid; i < n; i += kBlock) {
        const float ex = KeyFloat(B[i]);
        if (periodic && Moved(ToG(p.tok_extra)[base + i], ex, delta)) s.frame_changed = 1;
        ToG(p.tok_extra)[base + i] = ex;
        if (st.on && ex != FInf()) StageTok(s, st, base + i, f);
      }
      if (st.on) {
#pragma unroll
        for (int r = 0; r < kEpsRegs; ++r)
          if (e_src[r] >= 0) StageArc(s, st, pb + tid + r * kBlock, f, 0);
        if (n == 0 && tid == 0) s.empty_frame = 1;
      }
      next_b = B;
      next_base = base;
      // (a later frame that reads these extras from global memory -- the path below --
      // does so behind a full barrier of its own)
      LdsBarrier();
      prof(1);
      if (periodic) {
        // (frame f - 1, visited by an earlier sweep, is swept again only if an extra cost
        // of frame f moved by more than delta: lattice-simple-decoder.cc:208-216)
        const bool stop = f - 1 <= visited && !s.frame_changed;
        LdsBarrier();  // (everybody has read the flag before the next frame clears it)
        if (stop) break;
      }
      continue;
    }
    next_b = nullptr;  // this frame's extras go to global memory only
    have_pre = false;  // (a prefetch made for this frame is dropped: it takes the path below)
    const bool ne_in_lds = n <= kLdsToks;
    // (periodic: has any extra moved by more than delta?  The first round's `was` is what
    // the previous sweep left; later rounds only move further from it)
    bool first_round = true;
    KD_G uint32_t *gne = ToG(p.lat.scratch) + base;  // (the frame's own slice of the pool-wide scratch)
    while (true) {
      // start value: +inf, or for the last frame the token's own final term (:467)
      for (int i = tid; i < n; i += kBlock) {
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
      for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {  // (four tokens per lane in flight)
        float was[4];
        uint32_t now[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kBlock < n ? i0 + u * kBlock : i0;
          was[u] = ToG(p.tok_extra)[base + i];
          now[u] = ne_in_lds ? s.ne[i]
                             : __hip_atomic_load(ToFlat(&gne[i]), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u * kBlock >= n) continue;
          float v = KeyFloat(now[u]);
          if (last && v > lattice_beam) v = FInf();  // :507-509
          if (v != was[u]) {
            ToG(p.tok_extra)[base + i0 + u * kBlock] = v;
            ch = 1;
            if (periodic && first_round && Moved(was[u], v, delta)) s.frame_changed = 1;
          }
        }
      }
      if (ch) s.changed = 1;
      __syncthreads();
      const int changed = s.changed;
      __syncthreads();
      first_round = false;
      // without epsilon links nothing in this frame depends on this frame
      if (!changed || pn == 0) break;
    }
    if (periodic) {
      const bool stop = f - 1 <= visited && !s.frame_changed;
      __syncthreads();
      if (stop) break;
    }
    if (st.on) {  // (a frame of thousands of tokens: read back what the loop above left)
      for (int i = tid; i < n; i += kBlock)
        if (ToG(p.tok_extra)[base + i] != FInf()) StageTok(s, st, base + i, f);
      for (int i = tid; i < en; i += kBlock)
        if (ToG(p.link_dst)[eb + i] >= 0) StageArc(s, st, eb + i, f, 1);
      for (int i = tid; i < pn; i += kBlock)
        if (ToG(p.link_dst)[pb + i] >= 0) StageArc(s, st, pb + i, f, 0);
      if (n == 0 && tid == 0) s.empty_frame = 1;
    }
    prof(1);
  }
  prof_flush();
  __syncthreads();
  return f_low;
}

// GetRawLattice from the lists FinalizeDecoding left (Stage).  States are numbered
// frame-major like the reference's (:623-640 walks the frames in order) and, inside a frame,
// by graph state; arcs are ordered by (source state id, graph arc): the SAME arrays whatever
// order the sweep's windows and waves listed the survivors in (the reference's own order --
// token-list and link-list order, :584-657 -- is a property of its containers; what a caller
// can rely on there, and here, is that the same input gives the same arrays).  Two counting
// sorts over a thousand entries, once per utterance: tokens by frame (an atomic counter per
// frame, a prefix sum over the frames), then by state inside the frame (a token's rank is
// the number of its frame's tokens with a lower state: frames of the pruned lattice hold a
// handful); arcs by source id, then by graph arc inside the source's handful.
__device__ __forceinline__ uint32_t LatticeExportStaged(CParams &p, TailShared &s, int utt, int F,
                                                        bool any_final, const Stage &st) {
  const int tid = threadIdx.x;
  const int ns = s.n_tok_staged, na = s.n_arc_staged;
  if (s.empty_frame || ns == 0) return 0;  // GetRawLattice returns false (:615-619)
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  KD_G int32_t *soff = ToG(p.lat.frame_state_off) + fb;
  // scratch: three more of the workgroup's idle queue arrays (2 * table slots words each)
  const long long gslots = 1ll << p.g_log2_slots, wg = p.sched.wg_base + blockIdx.x;
  KD_G uint32_t *sc_a = reinterpret_cast<KD_G uint32_t *>(ToG(p.g_log_ac) + wg * 2 * gslots);
  KD_G uint32_t *sc_b = reinterpret_cast<KD_G uint32_t *>(ToG(p.g_log_gr) + wg * 2 * gslots);
  KD_G uint32_t *sc_c = ToG(p.g_wl) + wg * 2 * gslots;
  if (tid == 0) { s.n_finals = 0; s.start_id = -1; }
  for (int k = tid; k <= F; k += kBlock) soff[k] = 0;
  __syncthreads();
  // tokens per frame (and a provisional place inside the frame)
  for (int j = tid; j < ns; j += kBlock) {
    const int f = static_cast<int>(st.tok_f[j]);
    sc_c[j] = static_cast<uint32_t>(atomicAdd(ToFlat(&soff[f]), 1));
  }
  __syncthreads();
  int carry = 0;
  for (int k0 = 0; k0 <= F; k0 += kBlock) {  // counts -> offsets
    const int k = k0 + tid;
    // (the counters were bumped by L2 atomics, which do not refresh this CU's L1: an
    // agent-scope load reads what L2 holds)
    const int v = k <= F ? __hip_atomic_load(ToFlat(&soff[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int total;
    const int before = BlockExclScanI(s, v, total);
    if (k <= F) soff[k] = carry + before;
    carry += total;
  }
  __syncthreads();
  // the frames' segments: graph state (bit 31: a final state of the last frame) and token
  for (int j = tid; j < ns; j += kBlock) {
    const uint32_t g = st.tok_g[j];
    const int f = static_cast<int>(st.tok_f[j]);
    const int32_t state = ToG(p.tok_state)[g];
    bool fin = false;
    if (f == F) {  // :645-652
      const float fc = ToG(p.states)[state].final_cost;
      fin = !any_final || fc != FInf();
      if (fin) atomicAdd(&s.n_finals, 1);
    }
    const int at = soff[f] + static_cast<int>(sc_c[j]);
    sc_a[at] = static_cast<uint32_t>(state) | (fin ? 0x80000000u : 0u);
    sc_b[at] = g;
  }
  __syncthreads();
  const int nf = s.n_finals;
  if (tid == 0) {
    const unsigned long long words = static_cast<unsigned long long>(ns) + nf + 3ull * na;
    s.out_off = PoolTake(&p.ctl->out_top, p.out_base, p.lat.out_cap, words);
  }
  __syncthreads();
  const long long off = s.out_off;
  if (off < 0) return KD_UTT_OUT_FULL;
  KD_G unsigned long long *o_state = ToG(p.lat.out) + off, *o_final = o_state + ns,
                          *o_arc = o_final + nf;
  // (the lent device buffer, if any: the same words at the same offsets)
  const bool two = p.lat.out2 != nullptr;
  KD_G unsigned long long *d_state = ToG(two ? p.lat.out2 : p.lat.out) + off, *d_final = d_state + ns,
                          *d_arc = d_final + nf;
  auto pack = [](uint32_t lo, uint32_t hi) { return static_cast<unsigned long long>(hi) << 32 | lo; };
  // a token's id: its frame's offset + the number of the frame's tokens with a lower state
  // (one token per state and frame); the finals likewise, among the last frame's finals
  for (int j = tid; j < ns; j += kBlock) {
    const int f = static_cast<int>(st.tok_f[j]);
    const int at = soff[f] + static_cast<int>(sc_c[j]);
    const int lo = soff[f], hi = f < F ? soff[f + 1] : ns;
    const uint32_t mine = sc_a[at], state = mine & 0x7FFFFFFFu;
    int rank = 0, frank = 0;
    for (int k = lo; k < hi; ++k) {
      const uint32_t o = sc_a[k];
      const bool lower = (o & 0x7FFFFFFFu) < state;
      rank += lower;
      frank += lower && (o >> 31);
    }
    const int id = lo + rank;
    ToG(p.lat.tok_newid)[sc_b[at]] = id;
    o_state[id] = pack(static_cast<uint32_t>(f), state);
    if (two) d_state[id] = pack(static_cast<uint32_t>(f), state);
    if (f == 0 && static_cast<int32_t>(state) == p.start) s.start_id = id;  // :623-629
    if (mine >> 31) {
      const float fc = ToG(p.states)[state].final_cost;
      o_final[frank] = pack(static_cast<uint32_t>(id), __float_as_uint(any_final ? fc : 0.0f));
      if (two) d_final[frank] = pack(static_cast<uint32_t>(id), __float_as_uint(any_final ? fc : 0.0f));
    }
  }
  __syncthreads();  // every listed token's id is in place
  // arcs per source state (the prefix sum reuses sc_a as counters: ns + 1 words)
  for (int k = tid; k <= ns; k += kBlock) sc_a[k] = 0;
  __syncthreads();
  for (int j = tid; j < na; j += kBlock) {
    const int sid = ToG(p.lat.tok_newid)[ToG(p.link_src)[st.arc_at[j]]];
    sc_c[j] = atomicAdd(ToFlat(&sc_a[sid]), 1u);  // provisional place inside the source's arcs
  }
  __syncthreads();
  carry = 0;
  for (int k0 = 0; k0 <= ns; k0 += kBlock) {
    const int k = k0 + tid;
    const int v = k <= ns ? static_cast<int>(__hip_atomic_load(ToFlat(&sc_a[k]), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT)) : 0;
    int total;
    const int before = BlockExclScanI(s, v, total);
    if (k <= ns) sc_a[k] = static_cast<uint32_t>(carry + before);
    carry += total;
  }
  __syncthreads();
  for (int j = tid; j < na; j += kBlock) {  // the sources' segments: the links' graph arcs
    const long long at = st.arc_at[j];
    const int sid = ToG(p.lat.tok_newid)[ToG(p.link_src)[at]];
    sc_b[sc_a[sid] + sc_c[j]] = static_cast<uint32_t>(ToG(p.link_arc)[at]);
  }
  __syncthreads();
  for (int j = tid; j < na; j += kBlock) {
    const long long at = st.arc_at[j];
    const int32_t src = ToG(p.link_src)[at], dst = ToG(p.link_dst)[at], arc = ToG(p.link_arc)[at];
    const float gr = ToG(p.link_graph)[at], ac = ToG(p.link_ac)[at];
    const int sid = ToG(p.lat.tok_newid)[src];
    const int lo = static_cast<int>(sc_a[sid]), hi = static_cast<int>(sc_a[sid + 1]);
    int rank = 0;  // (a token has one link per graph arc: the arc orders its links)
    for (int k = lo; k < hi; ++k) rank += sc_b[k] < static_cast<uint32_t>(arc);
    const ArcRec ar = LoadArcRec(p.arcs, static_cast<uint32_t>(arc));
    KD_G unsigned long long *w = o_arc + 3ll * (lo + rank);
    w[0] = pack(static_cast<uint32_t>(sid), static_cast<uint32_t>(ToG(p.lat.tok_newid)[dst]));
    w[1] = pack(static_cast<uint32_t>(FullIlabel(p, ar, static_cast<uint32_t>(arc))), static_cast<uint32_t>(ToG(p.arc_olabel)[arc]));
    w[2] = pack(__float_as_uint(gr), __float_as_uint(ac));
    if (two) {
      KD_G unsigned long long *d = d_arc + 3ll * (lo + rank);
      d[0] = w[0];
      d[1] = w[1];
      d[2] = w[2];
    }
  }
  __syncthreads();
  if (tid == 0) {
    s.r_sizes[0] = ns;
    s.r_sizes[1] = na;
    s.r_sizes[2] = nf;
    s.r_sizes[3] = s.start_id;
    s.r_out_off = off;
  }
  return 0;
}

// GetRawLattice (:584-657).  Frames do not depend on one another here, so the waves of
// the workgroup take frames independently (a wave needs no barrier to rank the items of
// its frame: ballot and popcount).  Three passes over the frames: count live tokens and
// links; (prefix sum over the frames; one atomic takes the utterance's region of the
// output pool;) number the tokens and write states and final weights; write the arcs.
// Returns 0 or KD_UTT_OUT_FULL (the host grows the pool and asks again).
__device__ __forceinline__ uint32_t LatticeExport(CParams &p, TailShared &s, int utt, int F,
                                                  bool use_final_probs, bool finalized,
                                                  bool any_final) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  if (F < 1) return 0;  // KALDI_DECODER_ASSERT(num_frames > 0), :610: the host reports it
  KD_G int32_t *soff = ToG(p.lat.frame_state_off) + fb, *aoff = ToG(p.lat.frame_arc_off) + fb;
  if (tid == 0) { s.start_id = -1; s.empty_frame = 0; s.n_finals = 0; }
  __syncthreads();
  // (a token without surviving forward links has extra_cost +inf: FinalizeDecoding's or a
  // periodic sweep's doing -- the reference deletes it, :310-334; the search writes 0)
  // (PruneActiveTokens never calls PruneTokensForFrame(0) -- :216-219 prunes frame f + 1 --
  // so until FinalizeDecoding the reference still lists frame 0's link-less tokens)
  auto alive = [&](long long g, int f = 1) { return (f == 0 && !finalized) || ToG(p.tok_extra)[g] != FInf(); };
  constexpr int kT = 4;
  // pass 1: live tokens and surviving links of every frame
  for (int f = wave; f <= F; f += kWaves) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += 64 * kT) {
      bool a[kT];
#pragma unroll
      for (int u = 0; u < kT; ++u) {
        const int i = c0 + u * 64 + lane;
        a[u] = i < n && alive(base + i, f);
      }
#pragma unroll
      for (int u = 0; u < kT; ++u) cnt += __popcll(__ballot(a[u]));
    }
    int links = 0;
    for (int blk = 0; blk < 2; ++blk) {  // links leaving this frame's tokens: epsilon, then emitting
      if (blk == 1 && f == F) break;
      const long long lb = blk ? ToG(p.link_emit_base)[fb + f] : ToG(p.link_eps_base)[fb + f];
      const int ln = blk ? ToG(p.link_emit_count)[fb + f] : ToG(p.link_eps_count)[fb + f];
      for (int c0 = 0; c0 < ln; c0 += 64 * kT) {
        bool a[kT];
#pragma unroll
        for (int u = 0; u < kT; ++u) {
          const int i = c0 + u * 64 + lane;
          a[u] = i < ln && ToG(p.link_dst)[lb + i] >= 0;
        }
#pragma unroll
        for (int u = 0; u < kT; ++u) links += __popcll(__ballot(a[u]));
      }
    }
    if (lane == 0) {
      soff[f] = cnt;
      aoff[f] = links;
      if (cnt == 0) s.empty_frame = 1;  // :615-619
    }
  }
  __syncthreads();
  if (s.empty_frame) return 0;  // GetRawLattice returns false: sizes stay 0, start -1
  // first state id / arc position of each frame (exclusive scans over the frames)
  int ns = 0, na = 0;
  for (int f0 = 0; f0 <= F; f0 += kBlock) {
    const int f = f0 + tid;
    const int v1 = f <= F ? soff[f] : 0, v2 = f <= F ? aoff[f] : 0;
    int t1, t2;
    const int o1 = BlockExclScanI(s, v1, t1);
    const int o2 = BlockExclScanI(s, v2, t2);
    if (f <= F) { soff[f] = ns + o1; aoff[f] = na + o2; }
    ns += t1;
    na += t2;
  }
  // final states: the live tokens of the last frame that are final (or all of them)
  {
    const long long base = ToG(p.frame_base)[fb + F];
    const int n = ToG(p.frame_count)[fb + F];
    int c = 0;
    for (int i = tid; i < n; i += kBlock) {
      if (!alive(base + i)) continue;
      const float fc = ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost;
      c += (!(use_final_probs && any_final) || fc != FInf()) ? 1 : 0;
    }
    c = WaveSumInt(c);
    if (lane == 0 && c) atomicAdd(&s.n_finals, c);
  }
  __syncthreads();
  const int nf = s.n_finals;
  // the utterance's region of the output pool: states | finals | arcs, in 8-byte words
  if (tid == 0) {
    const unsigned long long words = static_cast<unsigned long long>(ns) + nf + 4ull * na;
    s.out_off = PoolTake(&p.ctl->out_top, p.out_base, p.lat.out_cap, words);
  }
  __syncthreads();
  const long long off = s.out_off;
  if (off < 0) return KD_UTT_OUT_FULL;
  KD_G unsigned long long *o_state = ToG(p.lat.out) + off, *o_final = o_state + ns,
                          *o_arc = o_final + nf;
  auto pack = [](uint32_t lo, uint32_t hi) { return static_cast<unsigned long long>(hi) << 32 | lo; };
  // pass 2: ids, states, final weights, the start state
  for (int f = wave; f <= F; f += kWaves) {
    const long long base = ToG(p.frame_base)[fb + f];
    const int n = ToG(p.frame_count)[fb + f];
    int before = soff[f], fin_before = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int i = c0 + lane;
      const bool a = i < n && alive(base + i, f);
      const int32_t st = i < n ? ToG(p.tok_state)[base + i] : 0;
      const unsigned long long m = __ballot(a);
      const int id = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                  __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
      if (i < n) ToG(p.lat.tok_newid)[base + i] = a ? id : -1;
      if (a) {
        o_state[id] = pack(static_cast<uint32_t>(f), static_cast<uint32_t>(st));
        if (f == 0 && st == p.start) s.start_id = id;  // :623-629
      }
      before += __popcll(m);
      if (f == F) {  // :645-652
        const float fc = a ? ToG(p.states)[st].final_cost : FInf();
        const bool is_fin = a && (!(use_final_probs && any_final) || fc != FInf());
        const unsigned long long fm = __ballot(is_fin);
        const int k = fin_before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(fm >> 32),
                                       __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(fm), 0));
        if (is_fin)
          o_final[k] = pack(static_cast<uint32_t>(id),
                            __float_as_uint((use_final_probs && any_final) ? fc : 0.0f));
        fin_before += __popcll(fm);
      }
    }
  }
  __syncthreads();  // every token's id is in place
  // pass 3: arcs
  for (int f = wave; f <= F; f += kWaves) {
    int before = aoff[f];
    for (int blk = 0; blk < 2; ++blk) {
      if (blk == 1 && f == F) break;
      const long long lb = blk ? ToG(p.link_emit_base)[fb + f] : ToG(p.link_eps_base)[fb + f];
      const int ln = blk ? ToG(p.link_emit_count)[fb + f] : ToG(p.link_eps_count)[fb + f];
      for (int c0 = 0; c0 < ln; c0 += 64 * kT) {
        int32_t dst[kT];
#pragma unroll
        for (int u = 0; u < kT; ++u) {
          const int i = c0 + u * 64 + lane;
          dst[u] = i < ln ? ToG(p.link_dst)[lb + i] : -1;
        }
#pragma unroll
        for (int u = 0; u < kT; ++u) {
          const unsigned long long m = __ballot(dst[u] >= 0);
          if (dst[u] >= 0) {
            const long long at = lb + c0 + u * 64 + lane;
            const int k = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                       __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
            const int32_t arc = ToG(p.link_arc)[at];
            const ArcRec ar = LoadArcRec(p.arcs, static_cast<uint32_t>(arc));
            KD_G unsigned long long *w = o_arc + 4ll * k;
            w[3] = pack(static_cast<uint32_t>(arc), 0u);  // (what the host orders a source's arcs by)
            w[0] = pack(static_cast<uint32_t>(ToG(p.lat.tok_newid)[ToG(p.link_src)[at]]),
                        static_cast<uint32_t>(ToG(p.lat.tok_newid)[dst[u]]));
            w[1] = pack(static_cast<uint32_t>(FullIlabel(p, ar, static_cast<uint32_t>(arc))), static_cast<uint32_t>(ToG(p.arc_olabel)[arc]));
            w[2] = pack(__float_as_uint(ToG(p.link_graph)[at]), __float_as_uint(ToG(p.link_ac)[at]));
          }
          before += __popcll(m);
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    s.r_sizes[0] = ns;
    s.r_sizes[1] = na;
    s.r_sizes[2] = nf;
    s.r_sizes[3] = s.start_id;
    s.r_out_off = off;
    s.r_arena_order = 1;
  }
  return 0;
}

// ---- the collector ---------------------------------------------------------------
// The reference frees what PruneActiveTokens finds dead (ForwardLinks at :264-275, tokens
// in PruneTokensForFrame, :310-334).  Here records live in append-only pools, so what
// SURVIVES of frames [lo, hi) is moved instead: live tokens and surviving links of each
// frame are copied, compacted, into the utterance's archive; link endpoints follow their
// tokens through a forwarding word per token (tok_newid); the frame index is pointed at
// the copies; and the nursery chunks that only held such frames are freed for reuse by
// this utterance (TakeChunks).
// Work is dealt out in PIECES of a frame's tokens / epsilon links / emitting links
// (256 records, more when a pass would otherwise have more than kMaxPieces): frame sizes
// are heavy-tailed -- a median of 68 links, some of 20 000 -- and a wave per frame left
// eleven waves waiting for the one with the large frame.  Pass A counts each piece's
// survivors (ballot + popcount, no workgroup barrier inside a pass), a prefix sum over
// the pieces gives every piece its place, the archive space is taken in one piece per
// pool -- so a pool that is exhausted is noticed before anything has moved -- and passes
// B (tokens) and C (links) copy.
// (the stress build, `make stress`, shrinks these so that small test cases reach the paths
// that only very large collections take at the shipped sizes: piece sizes doubled until
// the pieces fit, collections split over several calls)
#ifndef KD_COLLECT_FRAMES
#define KD_COLLECT_FRAMES 256
#endif
#ifndef KD_MAX_PIECES
#define KD_MAX_PIECES 8192
#endif
constexpr int kCollectFrames = KD_COLLECT_FRAMES;  // frames per call
constexpr int kMaxPieces = KD_MAX_PIECES;          // pieces per call and record kind
constexpr int kPiece = 256;                        // records per piece (x 2^k)
// (a non-empty block is at least one piece however large the pieces get: the doubling of
// the piece size in LatticeCollect ends because of this)
static_assert(kMaxPieces >= 2 * kCollectFrames, "one piece per frame / link block always fits");
constexpr int kCU = 4;               // 64-wide steps of a piece a wave has in flight
static_assert(kPiece == 64 * kCU, "a piece is one unrolled step (or 2^k of them)");
static_assert(kChunkRing <= 64, "one lane per retired chunk");

__device__ __forceinline__ long long ArchiveAlloc(CParams &p, KD_G UttMem *m, bool links, long long need) {
  long long &next = *(links ? ToFlat(&m->alink_next) : ToFlat(&m->atok_next));
  long long &end = *(links ? ToFlat(&m->alink_end) : ToFlat(&m->atok_end));
  if (next + need > end) {
    const long long standard = links ? p.link_chunk : p.tok_chunk;
    const long long want = need > standard ? need : standard;
    const long long b = PoolChunk(p, links ? &m->link : &m->tok, links, static_cast<unsigned long long>(want));
    if (b < 0) return -1;
    next = b;
    end = b + want;
  }
  const long long r = next;
  next += need;
  return r;
}

// rank of this lane among the lanes of `m` below it
__device__ __forceinline__ int RankIn(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
}

__device__ __forceinline__ uint32_t LatticeCollect(CParams &p, TailShared &s, int utt, int lo, int hi) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long fb = static_cast<long long>(utt) * (p.max_frames + 1);
  KD_G UttMem *m = ToG(p.mem) + utt;
  const int nfr = hi - lo;  // <= kCollectFrames
  // tables in the key array: the frames' descriptors; where each frame's / link block's
  // pieces begin; which frame / block a piece belongs to; the pieces' survivor counts
  // (then their places)
  long long *d_base = reinterpret_cast<long long *>(s.ne), *d_eb = d_base + kCollectFrames,
            *d_pb = d_eb + kCollectFrames, *arch = d_pb + kCollectFrames;  // arch[2]: the archive space taken
  int32_t *d_n = reinterpret_cast<int32_t *>(arch + 2), *d_en = d_n + kCollectFrames, *d_pn = d_en + kCollectFrames;
  int32_t *tp_pref = d_pn + kCollectFrames;        // [nfr + 1]
  int32_t *lp_pref = tp_pref + kCollectFrames + 1;  // [2 nfr + 1]: block 2j = epsilon, 2j + 1 = emitting links of frame lo + j
  int32_t *tpc = lp_pref + 2 * kCollectFrames + 1;  // [PT + 1]
  int32_t *lpc = tpc + kMaxPieces + 1;              // [PL + 1]
  uint16_t *tp_frame = reinterpret_cast<uint16_t *>(lpc + kMaxPieces + 1), *lp_block = tp_frame + kMaxPieces;
  static_assert((3 * 8 + 3 * 4) * kCollectFrames + 16 + 4 * (3 * kCollectFrames + 2) + 8 * (kMaxPieces + 1) +
                        4 * kMaxPieces <= 4 * kLdsToks, "the collector's tables fit the key array");
  static_assert(2 * kCollectFrames <= 65536, "piece owners are 16-bit");
  for (int j = tid; j < nfr; j += kBlock) {
    const long long at = fb + lo + j;
    d_base[j] = ToG(p.frame_base)[at];
    d_n[j] = ToG(p.frame_count)[at];
    d_eb[j] = ToG(p.link_emit_base)[at];
    d_en[j] = ToG(p.link_emit_count)[at];
    d_pb[j] = ToG(p.link_eps_base)[at];
    d_pn[j] = ToG(p.link_eps_count)[at];
  }
  __syncthreads();
  // ---- the pieces (kCollectFrames <= kBlock: a thread per frame)
  static_assert(kCollectFrames <= kBlock, "a thread per collected frame");
  int shift = 0, PT = 0, PL = 0;
  while (true) {
    const int S = kPiece << shift;
    const int n = tid < nfr ? d_n[tid] : 0, pn = tid < nfr ? d_pn[tid] : 0, en = tid < nfr ? d_en[tid] : 0;
    const int tq = (n + S - 1) / S, pq = (pn + S - 1) / S, eq = (en + S - 1) / S;
    const int tb = BlockExclScanI(s, tq, PT);
    const int lb = BlockExclScanI(s, pq + eq, PL);
    if (PT <= kMaxPieces && PL <= kMaxPieces) {  // (the same totals in every thread)
      if (tid < nfr) {
        tp_pref[tid] = tb;
        lp_pref[2 * tid] = lb;
        lp_pref[2 * tid + 1] = lb + pq;
        for (int q = 0; q < tq; ++q) tp_frame[tb + q] = static_cast<uint16_t>(tid);
        for (int q = 0; q < pq; ++q) lp_block[lb + q] = static_cast<uint16_t>(2 * tid);
        for (int q = 0; q < eq; ++q) lp_block[lb + pq + q] = static_cast<uint16_t>(2 * tid + 1);
      }
      if (tid == 0) { tp_pref[nfr] = PT; lp_pref[2 * nfr] = PL; }
      break;
    }
    ++shift;
  }
  const int S = kPiece << shift;
  __syncthreads();
  // ---- pass A: what survives of each piece (kCU 64-wide steps in flight together)
  for (int q = wave; q < PT + PL; q += kWaves) {
    int c = 0;
    if (q < PT) {
      const int j = tp_frame[q];
      const long long base = d_base[j];
      const int n = d_n[j], c_begin = (q - tp_pref[j]) * S, c_end = c_begin + S < n ? c_begin + S : n;
      for (int c0 = c_begin; c0 < c_end; c0 += 64 * kCU) {
        float ex[kCU];
#pragma unroll
        for (int u = 0; u < kCU; ++u) {
          const int i = c0 + u * 64 + lane;
          ex[u] = ToG(p.tok_extra)[base + (i < c_end ? i : c0)];
        }
#pragma unroll
        for (int u = 0; u < kCU; ++u)
          c += __popcll(__ballot(c0 + u * 64 + lane < c_end && ex[u] != FInf()));
      }
      if (lane == 0) tpc[q] = c;
    } else {
      const int ql = q - PT, blk = lp_block[ql], j = blk >> 1;
      const long long lb = (blk & 1) ? d_eb[j] : d_pb[j];
      const int ln = (blk & 1) ? d_en[j] : d_pn[j];
      const int c_begin = (ql - lp_pref[blk]) * S, c_end = c_begin + S < ln ? c_begin + S : ln;
      for (int c0 = c_begin; c0 < c_end; c0 += 64 * kCU) {
        int32_t d[kCU];
#pragma unroll
        for (int u = 0; u < kCU; ++u) {
          const int i = c0 + u * 64 + lane;
          d[u] = ToG(p.link_dst)[lb + (i < c_end ? i : c0)];
        }
#pragma unroll
        for (int u = 0; u < kCU; ++u)
          c += __popcll(__ballot(c0 + u * 64 + lane < c_end && d[u] >= 0));
      }
      if (lane == 0) lpc[ql] = c;
    }
  }
  __syncthreads();
  // ---- counts -> places (exclusive prefix sums over the pieces; totals behind them)
  for (int which = 0; which < 2; ++which) {
    int32_t *pc = which ? lpc : tpc;
    const int P = which ? PL : PT;
    int carry = 0;
    for (int q0 = 0; q0 < P; q0 += kBlock) {
      const int q = q0 + tid;
      const int v = q < P ? pc[q] : 0;
      int total;
      const int before = BlockExclScanI(s, v, total);
      if (q < P) pc[q] = carry + before;
      carry += total;
    }
    if (tid == 0) pc[P] = carry;
  }
  __syncthreads();
  // ---- the archive space, in one piece per pool
  if (tid == 0) {
    const long long bt = ArchiveAlloc(p, m, false, tpc[PT]);
    const long long bl = bt < 0 ? -1 : ArchiveAlloc(p, m, true, lpc[PL]);
    s.alloc_err = bt < 0 ? KD_UTT_ARENA_FULL : bl < 0 ? KD_UTT_LINKS_FULL : 0u;
    arch[0] = bt;
    arch[1] = bl;
  }
  __syncthreads();
  if (s.alloc_err) return s.alloc_err;
  const long long bt = arch[0], bl = arch[1];
  // ---- pass B: tokens; every token of these frames gets its forwarding word
  for (int q = wave; q < PT; q += kWaves) {
    const int j = tp_frame[q];
    const long long base = d_base[j];
    const int n = d_n[j], c_begin = (q - tp_pref[j]) * S, c_end = c_begin + S < n ? c_begin + S : n;
    long long at = bt + tpc[q];
    for (int c0 = c_begin; c0 < c_end; c0 += 64 * kCU) {
      float ex[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = c0 + u * 64 + lane;
        ex[u] = i < c_end ? ToG(p.tok_extra)[base + i] : FInf();
      }
      // (a dead token's state and cost are not needed: its lane reads the step's first
      // record instead, so that every lane's loads are issued together)
      int32_t st[kCU];
      double co[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const long long g = base + (ex[u] != FInf() ? c0 + u * 64 + lane : c0);
        st[u] = ToG(p.tok_state)[g];
        co[u] = ToG(p.tok_cost)[g];
      }
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = c0 + u * 64 + lane;
        const bool a = ex[u] != FInf();
        const unsigned long long mk = __ballot(a);
        const long long to = at + RankIn(mk);
        if (a) {
          ToG(p.tok_state)[to] = st[u];
          ToG(p.tok_cost)[to] = co[u];
          ToG(p.tok_extra)[to] = ex[u];
        }
        if (i < c_end) ToG(p.lat.tok_newid)[base + i] = a ? static_cast<int32_t>(to) : -1;
        at += __popcll(mk);
      }
    }
  }
  __syncthreads();  // every forwarding word is in place
  // ---- the emitting links of frame lo - 1 (archived earlier) end in frame lo's tokens
  if (lo > 0) {
    const long long lb = ToG(p.link_emit_base)[fb + lo - 1];
    const int ln = ToG(p.link_emit_count)[fb + lo - 1];
    for (int i = tid; i < ln; i += kBlock) {
      const int32_t d = ToG(p.link_dst)[lb + i];
      if (d >= 0) ToG(p.link_dst)[lb + i] = ToG(p.lat.tok_newid)[d];
    }
  }
  // ---- pass C: links, endpoints forwarded (an emitting link of frame hi - 1 ends in
  // frame hi, which stays where it is until the next collection)
  for (int ql = wave; ql < PL; ql += kWaves) {
    const int blk = lp_block[ql], j = blk >> 1;
    const long long lb = (blk & 1) ? d_eb[j] : d_pb[j];
    const int ln = (blk & 1) ? d_en[j] : d_pn[j];
    const int c_begin = (ql - lp_pref[blk]) * S, c_end = c_begin + S < ln ? c_begin + S : ln;
    const bool fwd_dst = lo + j + (blk & 1) < hi;
    long long at = bl + lpc[ql];
    for (int c0 = c_begin; c0 < c_end; c0 += 64 * kCU) {
      int32_t d[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const int i = c0 + u * 64 + lane;
        d[u] = i < c_end ? ToG(p.link_dst)[lb + i] : -1;
      }
      int32_t src[kCU], arc[kCU], dst[kCU];
      float ac[kCU], gr[kCU], dl[kCU];
#pragma unroll
      for (int u = 0; u < kCU; ++u) {  // (an excised link's lane reads the step's first record)
        const long long l = lb + (d[u] >= 0 ? c0 + u * 64 + lane : c0);
        src[u] = ToG(p.link_src)[l];
        arc[u] = ToG(p.link_arc)[l];
        ac[u] = ToG(p.link_ac)[l];
        gr[u] = ToG(p.link_graph)[l];
        dl[u] = ToG(p.link_delta)[l];
      }
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        src[u] = ToG(p.lat.tok_newid)[src[u]];
        dst[u] = fwd_dst ? ToG(p.lat.tok_newid)[d[u] >= 0 ? d[u] : 0] : d[u];
      }
#pragma unroll
      for (int u = 0; u < kCU; ++u) {
        const bool a = d[u] >= 0;
        const unsigned long long mk = __ballot(a);
        const long long to = at + RankIn(mk);
        if (a) {
          ToG(p.link_src)[to] = src[u];
          ToG(p.link_dst)[to] = dst[u];
          ToG(p.link_arc)[to] = arc[u];
          ToG(p.link_ac)[to] = ac[u];
          ToG(p.link_graph)[to] = gr[u];
          ToG(p.link_delta)[to] = dl[u];
        }
        at += __popcll(mk);
      }
    }
  }
  __syncthreads();  // (every old block has been read: the frame index can move)
  for (int j = tid; j < nfr; j += kBlock) {
    const int k = lo + j;
    const int t0 = tpc[tp_pref[j]], t1 = tpc[tp_pref[j + 1]];
    const int e0 = lpc[lp_pref[2 * j]], e1 = lpc[lp_pref[2 * j + 1]], e2 = lpc[lp_pref[2 * j + 2]];
    ToG(p.frame_base)[fb + k] = bt + t0;
    ToG(p.frame_count)[fb + k] = t1 - t0;
    ToG(p.link_eps_base)[fb + k] = bl + e0;
    ToG(p.link_eps_count)[fb + k] = e1 - e0;
    ToG(p.link_emit_base)[fb + k] = bl + e1;
    ToG(p.link_emit_count)[fb + k] = e2 - e1;
  }
  // ---- nursery chunks that held nothing younger than hi - 1 are free again (one wave
  // per pool, one lane per retired chunk: the lists live in global memory, and a
  // single lane walking them paid a memory round trip per entry)
  if (tid == 0) m->gc_frame = hi;
  if (wave < 2) {
    KD_G ChunkRing *r = wave ? &m->link : &m->tok;
    const int nr = r->n_retired, nfree = r->n_free;
    const bool have = lane < nr;
    const int32_t start = r->retired_start[have ? lane : 0], last = r->retired_last[have ? lane : 0];
    const bool old = have && last < hi;
    const unsigned long long mo = __ballot(old), mk = __ballot(have && !old);
    const int fpos = nfree + RankIn(mo);
    // (a freed chunk the full free list cannot take is lost until InitDecoding)
    if (old && fpos < kChunkRing) r->free_start[fpos] = start;
    if (have && !old) {
      const int kp = RankIn(mk);
      r->retired_start[kp] = start;
      r->retired_last[kp] = last;
    }
    if (lane == 0) {
      const int nf = nfree + __popcll(mo);
      r->n_free = nf < kChunkRing ? nf : kChunkRing;
      r->n_retired = __popcll(mk);
    }
  }
  __syncthreads();
  return 0;
}

// PruneActiveTokens between two frames of the search (frame F is the newest), then the
// collection of the frames that are older than prune_interval.  Uses the search's LDS
// block: the caller re-establishes what the search keeps there.  Returns 0 or the
// KD_UTT_* bit of the pool the archive could not get space from (nothing has moved then).
__device__ __noinline__ uint32_t PeriodicPrune(CParams *pp, int utt, int F, int visited) {
  CParams &p = *pp;
  TailShared &s = TailLds();
  Stage st = TailStage(p, false);
  LatticePrune(p, s, utt, F, 0.0f, false, st, /*periodic=*/true, visited,
               p.lat.lattice_beam * p.lat.prune_scale);
  const int hi = F - p.lat.prune_interval;
  uint32_t r = 0;
#if defined(KD_PROFILE) && defined(KD_PROF_TAIL)
  const long long lap = static_cast<long long>(__builtin_amdgcn_s_memtime());
#endif
  while (r == 0) {
    const int lo = Uni(ToG(p.mem)[utt].gc_frame);
    if (lo >= hi) break;
    r = LatticeCollect(p, s, utt, lo, hi - lo > kCollectFrames ? lo + kCollectFrames : hi);
  }
#if defined(KD_PROFILE) && defined(KD_PROF_TAIL)
  // diagnostic build `prof_tail`: the collector's cycles on top of bucket 17 (the search's
  // link records); part of bucket 23 too
  if (threadIdx.x == 0) Lds().cyc[17] += static_cast<long long>(__builtin_amdgcn_s_memtime()) - lap;
#endif
  return r;
}

// The tail of an utterance that has reached its target frame F.
// Results: TailShared::r_* (sizes 0 / start -1 / offset -1 when nothing was exported).
__device__ __noinline__ uint32_t LatticeTail(CParams *pp, int utt, int F, uint32_t flags) {
  CParams &p = *pp;
  TailShared &s = TailLds();
  if (threadIdx.x == 0) {
    s.r_sizes[0] = 0; s.r_sizes[1] = 0; s.r_sizes[2] = 0; s.r_sizes[3] = -1;
    s.r_out_off = -1;
    s.r_arena_order = 0;
  }
#if defined(KD_PROFILE) && !defined(KD_PROF_SIZE)
  // diagnostic build: cycles of the tail's three parts, buckets 20-22 of the utterance
  long long lap = static_cast<long long>(__builtin_amdgcn_s_memtime());
  auto tail_lap = [&](int bucket) {
    if (threadIdx.x == 0) {
      const long long now = static_cast<long long>(__builtin_amdgcn_s_memtime());
      ToG(p.stats)[static_cast<long long>(utt) * kStatStride + KD_NUM_STATS + bucket] += now - lap;
      lap = now;
    }
  };
#else
  auto tail_lap = [](int) {};
#endif
  const bool was_finalized = Uni(ToG(p.lat.finalized)[utt]) != 0;
  float final_best_cost = 0.0f, final_rel = 0.0f;
  bool any_final = false;
  if (!was_finalized) {
    LatticeFinalCosts(p, s, utt, F, final_best_cost, any_final, final_rel);
    if (threadIdx.x == 0) {
      ToG(p.lat.final_best_cost)[utt] = final_best_cost;
      ToG(p.lat.any_final)[utt] = any_final ? 1 : 0;
      ToG(p.lat.final_rel)[utt] = final_rel;
    }
  } else {  // (fixed by FinalizeDecoding, :413-417)
    final_best_cost = ToG(p.lat.final_best_cost)[utt];
    any_final = ToG(p.lat.any_final)[utt] != 0;
    final_rel = ToG(p.lat.final_rel)[utt];
  }
  if (threadIdx.x == 0) s.r_final_rel = final_rel;
  tail_lap(20);
  bool finalized = was_finalized, staged = false;
  // (Decode(): FinalizeDecoding and GetRawLattice(use_final_probs = true) in one tail)
  const Stage st = TailStage(p, (flags & kUttFinalize) && (flags & kUttExport) && !was_finalized && F >= 1);
  if ((flags & kUttFinalize) && !was_finalized) {
    LatticePrune(p, s, utt, F, final_best_cost, any_final, st);
    if (threadIdx.x == 0) ToG(p.lat.finalized)[utt] = 1;
    finalized = true;
    // (the export's scratch arrays are as long as the lists, and one of them holds ns + 1 words)
    staged = st.on && !s.stage_overflow && s.n_tok_staged < st.cap;
    tail_lap(21);
  }
  uint32_t r = 0;
  if (flags & (kUttExport | kUttExportNoFinal)) {
    if (staged) r = LatticeExportStaged(p, s, utt, F, any_final, st);
    else r = LatticeExport(p, s, utt, F, (flags & kUttExport) != 0, finalized, any_final);
    tail_lap(22);
  }
  return r;
}
