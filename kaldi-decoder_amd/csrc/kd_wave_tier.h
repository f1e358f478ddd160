// kaldi-decoder_amd/csrc/kd_wave_tier.h -- the WAVE tier of the lattice search: a frame whose
// frontier is at most one token per lane of ONE wavefront (64 tokens) is processed by that
// wave alone, with no workgroup barrier at all.  Part of kd_search.hip (included inside its
// namespace, after the tails).
//
// Replaces, for such frames, from /root/reference/kaldi-decoder/csrc/:
//   lattice-simple-decoder.cc:364-402  ProcessEmitting (+ FindOrAddToken :82-120)
//   lattice-simple-decoder.cc:122-191  ProcessNonemitting
//   faster-decoder.cc:244-336          GetCutoff (the branches that cannot bind at <= 64 tokens)
//
// Why: on an HLG decode at beam 15 the MEDIAN frame holds 2 tokens and 90 % hold fewer than
// 70 (tools/frame_hist.py), yet a frame of the workgroup-wide LDS-resident tier costs 36 k
// cycles whatever it holds -- a chain of ~20 workgroup barriers and five or six dependent
// memory round trips with twelve waves in lock step.  One wave needs no barrier (its lanes
// run in lock step; LDS operations of a wave complete in order), so a frame costs its
// memory round trips and nothing else, and the other waves of the workgroup are free to
// run OTHER utterances' frames meanwhile: twelve independent latency chains per CU instead
// of one.  (NOT in the shipped build: -DKD_WAVE_TIER, see the note in kd_search.hip.)
//
// Same semantics as SmallFrame / FrameFront + FrameBack (DESIGN.md "Order dependence"):
// float costs `cur + ac + graph`, the cutoff is the value the reference's running cutoff
// ends at, strict improvement replaces, every arc below the cutoff is a forward link, an
// epsilon link is kept iff it was generated from its source token's final cost.  A frame
// that does not fit the wave's table or queue is not committed (nothing has left the wave's
// LDS by then) and runs again in the workgroup-wide tiers.

constexpr int kWvFront = 64;                 // frontier: at most one token per lane
constexpr int kWvLog2 = 8;
constexpr int kWvSlots = 1 << kWvLog2;       // the wave's token table
constexpr int kWvFill = kWvSlots * 3 / 4;    // ... holds at most this many tokens
constexpr int kWvQueue = 256;                // candidate queue entries
constexpr int kWvV = 512;                    // widest log-likelihood row a wave stages itself
constexpr uint32_t kWvDead = 0xFFFFFFFFu;    // cost word of an empty slot (above FloatKey(+inf))
constexpr uint32_t kWvEps = 0x8000u;         // queue entry's source is a table slot (epsilon arc)

struct WaveLds {
  float ll[kWvV];               // this frame's log-likelihood row (when the wave stages it)
  uint32_t key[kWvSlots];       // state | has-epsilon flag; after compaction: index in the block
  uint32_t cost[kWvSlots];      // FloatKey of the token's cost
  uint32_t q_h[kWvQueue];       // destination state | flag; after insert: its slot, or kKeyEmpty
  uint32_t q_arc[kWvQueue];
  uint32_t q_nk[kWvQueue];      // FloatKey of the candidate's cost
  float q_ac[kWvQueue];         // acoustic cost (emitting) / source cost at generation (epsilon)
  float q_gr[kWvQueue];
  uint16_t q_src[kWvQueue];     // frontier index, or kWvEps | source slot
  uint16_t live[kWvFill];       // slots that hold a token, in arena order
  int32_t pre[kWvFront + 1];    // expansion: exclusive prefix sums of the tokens' degrees
  uint32_t wl[2][kWvSlots / 32];  // epsilon worklists: slot bitmasks
  int32_t alloc_ok;
  int32_t pad[6];
};
static_assert(sizeof(WaveLds) % 16 == 0, "wave regions are 16-byte aligned");

// Orders the wave's LDS traffic: the hardware completes a wave's LDS operations in order,
// this keeps the COMPILER from moving a lane's read above another lane's write.
__device__ __forceinline__ void WaveSync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Wave-wide minimum and inclusive prefix sum on the VALU's data-parallel primitives (DPP:
// row shifts inside the four rows of sixteen lanes, then the row broadcasts) -- seven
// vector instructions, no LDS round trip.  (__shfl_* compile to ds_bpermute: six dependent
// LDS-latency steps per reduction, ~700 cycles; a frame of the wave tier does ten of them.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int DppMove(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t WaveMinU32(uint32_t v) {
  auto mn = [](uint32_t a, int b) { const uint32_t y = static_cast<uint32_t>(b); return y < a ? y : a; };
  const int id = -1;  // 0xFFFFFFFF
  v = mn(v, DppMove<0x111, 0xf>(id, static_cast<int>(v)));  // row_shr:1
  v = mn(v, DppMove<0x112, 0xf>(id, static_cast<int>(v)));  // row_shr:2
  v = mn(v, DppMove<0x114, 0xf>(id, static_cast<int>(v)));  // row_shr:4
  v = mn(v, DppMove<0x118, 0xf>(id, static_cast<int>(v)));  // row_shr:8: lane 15 of a row holds the row's
  v = mn(v, DppMove<0x142, 0xa>(id, static_cast<int>(v)));  // row_bcast:15 into rows 1 and 3
  v = mn(v, DppMove<0x143, 0xc>(id, static_cast<int>(v)));  // row_bcast:31 into rows 2 and 3
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ int WaveInclScan(int v) {
  v += DppMove<0x111, 0xf>(0, v);
  v += DppMove<0x112, 0xf>(0, v);
  v += DppMove<0x114, 0xf>(0, v);
  v += DppMove<0x118, 0xf>(0, v);  // inclusive within each row of sixteen
  v += DppMove<0x142, 0xa>(0, v);  // rows 1 and 3 += the total of the row below
  v += DppMove<0x143, 0xc>(0, v);  // rows 2 and 3 += the total of rows 0 and 1
  return v;
}
__device__ __forceinline__ int LaneRank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
}

// What a wave accumulates over the frames it runs (wave-uniform; folded into the utterance's
// counters by the caller).
struct WaveStats {
  int frames = 0, tokens_in = 0, expanded = 0, arcs = 0, kept = 0, tokens_out = 0, eps_arcs = 0,
      eps_exp = 0, max_tokens = 0, links = 0;
};

// The wave's frontier: lane i < n holds token i of the arena block at `base`.
struct WaveFront {
  uint32_t st;
  float cost;
  int abeg, deg;   // its emitting arcs
};

// GetCutoff (faster-decoder.cc:244-336) for a frontier of n <= 64 tokens: can the wave tier
// decide it without a selection?  Yes when there are no limits (:249-272) and when neither
// limit can bind (n <= max_active, n <= min_active: :289, :311-323).
__device__ __forceinline__ bool WaveCutoffOk(CParams &p, int n) {
  if (p.max_active == INT_MAX && p.min_active == 0) return true;
  return n <= p.max_active && n <= p.min_active;
}

__device__ __forceinline__ void WaveClearTable(WaveLds &w) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int h = lane; h < kWvSlots; h += 64) {
    w.key[h] = kKeyEmpty;
    w.cost[h] = kWvDead;
  }
  WaveSync();
}

// Finds or claims the slot of `k` in the wave's table; kKeyEmpty: the table is full.
__device__ __forceinline__ uint32_t WaveClaim(WaveLds &w, uint32_t k) {
  uint32_t h = ((k & kStateMask) * 2654435761u) >> (32 - kWvLog2);
  for (int probe = 0; probe < kWvSlots; ++probe) {
    const uint32_t cur = w.key[h];
    if (cur == k) return h;
    if (cur == kKeyEmpty) {
      const uint32_t old = atomicCAS(&w.key[h], kKeyEmpty, k);
      if (old == kKeyEmpty || old == k) return h;
    }
    h = (h + 1) & (kWvSlots - 1);
  }
  return kKeyEmpty;
}

// Scores the arcs of up to 64 tokens, 64 consecutive arcs of the concatenated arc ranges
// per row whatever the tokens' degrees (a binary search over the prefix sums in w.pre finds
// each lane's token), kRows rows in flight, and appends what survives the running cutoff to
// the wave's queue.  Lane t holds token t: its first arc `abeg`, its degree `deg` (0: not
// expanded), its cost, and what the queue entry names as source.  `run_min` (wave-uniform)
// is the lowest new cost so far (emitting) -- tightened after every group of rows.
// Returns false if the queue overflowed.
template <bool EPS>
__device__ __forceinline__ bool WaveExpand(CParams &p, WaveLds &w, const float *ll, int abeg, int deg,
                                           float cost, uint32_t src, float adaptive_beam,
                                           float cutoff, uint32_t &run_min, int &q_n, int &n_arcs) {
  constexpr int kRows = 4;
  const int lane = threadIdx.x & 63;
  const int incl = WaveInclScan(deg);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  n_arcs += total;
  if (total == 0) return true;
  w.pre[lane] = incl - deg;
  if (lane == 0) w.pre[kWvFront] = total;
  WaveSync();
  for (int r0 = 0; r0 < total; r0 += 64 * kRows) {
    ArcRec ar[kRows];
    int tok[kRows];
    uint32_t aidx[kRows];
    bool on[kRows];
#pragma unroll
    for (int g = 0; g < kRows; ++g) {
      const int pos = r0 + 64 * g + lane;
      on[g] = pos < total;
      const int ps = on[g] ? pos : 0;
      int t = 0;  // the last token whose prefix sum is <= pos
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1)
        if (w.pre[t + step] <= ps) t += step;
      tok[g] = t;
      aidx[g] = static_cast<uint32_t>(__shfl(abeg, t, 64) + (ps - w.pre[t]));
    }
#pragma unroll
    for (int g = 0; g < kRows; ++g) ar[g] = LoadArcRecIf(p.arcs, aidx[g], on[g], 0u);
    uint32_t nk[kRows];
    float aux[kRows];
    bool keep[kRows];
    unsigned long long bal[kRows];
    const float bound = EPS ? cutoff : KeyFloat(run_min) + adaptive_beam;
    uint32_t mk = kFInfKey;
#pragma unroll
    for (int g = 0; g < kRows; ++g) {
      const float cur_cost = __shfl(cost, tok[g], 64);
      float tot_cost;
      if constexpr (EPS) {
        // lattice-simple-decoder.cc:170-174: float, strict
        tot_cost = cur_cost + ar[g].weight;
        aux[g] = cur_cost;
      } else {
        // lattice-simple-decoder.cc:383-390: float, cur + ac + graph
        const float ac_cost = -ll[on[g] ? ar[g].ilabel - 1 : 0];
        tot_cost = cur_cost + ac_cost + ar[g].weight;
        aux[g] = ac_cost;
      }
      keep[g] = on[g] && tot_cost < bound;
      nk[g] = FloatKey(tot_cost);
      bal[g] = __ballot(keep[g]);
      if (!EPS && keep[g] && nk[g] < mk) mk = nk[g];
    }
#pragma unroll
    for (int g = 0; g < kRows; ++g) {
      if (bal[g] == 0) continue;
      const int pos = q_n + LaneRank(bal[g]);
      const uint32_t s_of = static_cast<uint32_t>(__shfl(static_cast<int>(src), tok[g], 64));
      if (keep[g] && pos < kWvQueue) {
        w.q_h[pos] = ar[g].next;
        w.q_arc[pos] = aidx[g];
        w.q_nk[pos] = nk[g];
        w.q_ac[pos] = aux[g];
        w.q_gr[pos] = ar[g].weight;
        w.q_src[pos] = static_cast<uint16_t>(s_of);
      }
      q_n += __popcll(bal[g]);
    }
    if (q_n > kWvQueue) return false;
    if constexpr (!EPS) {
      if (__ballot(mk != kFInfKey)) {
        mk = WaveMinU32(mk);
        run_min = mk < run_min ? mk : run_min;
      }
    }
  }
  return true;
}

// Insert phase over queue entries [b, e): claim + relax; a token that is new, or (epsilon
// rounds) improved, and has epsilon arcs is flagged in worklist mask `out`.  Returns false
// if the table overflowed.
template <bool EPS>
__device__ __forceinline__ bool WaveInsert(WaveLds &w, int b, int e, uint32_t cutkey, int out,
                                           int &n_live, int &n_emit_links) {
  const int lane = threadIdx.x & 63;
  bool full = false;
  for (int i0 = b; i0 < e; i0 += 64) {
    const int i = i0 + lane;
    const bool act = i < e;
    const uint32_t k = w.q_h[act ? i : b];
    const uint32_t nk = w.q_nk[act ? i : b];
    uint32_t h = kKeyEmpty;
    bool fresh = false, push = false;
    // emitting entries were queued under the running cutoff: apply the final one
    if (act && (EPS || nk < cutkey)) {
      h = WaveClaim(w, k);
      if (h == kKeyEmpty) {
        full = true;
      } else {
        const uint32_t old = atomicMin(&w.cost[h], nk);
        fresh = old == kWvDead;  // the state's first token this frame
        push = (k & kHasEps) && (fresh || (EPS && old > nk));  // :63-65, :97-113
      }
    }
    const unsigned long long fm = __ballot(fresh);
    if (fresh) {
      const int pos = n_live + LaneRank(fm);
      if (pos < kWvFill) w.live[pos] = static_cast<uint16_t>(h);
    }
    n_live += __popcll(fm);
    if (push) atomicOr(&w.wl[out][h >> 5], 1u << (h & 31));
    if (!EPS) n_emit_links += __popcll(__ballot(h != kKeyEmpty));
    if (act) w.q_h[i] = h;
    if (__ballot(full) || n_live > kWvFill) return false;
  }
  WaveSync();
  return true;
}

// One frame of one utterance by one wave.  `fr` / `n` / `base`: the frontier (in: this
// frame's; out: the next frame's, valid in the lanes when n <= kWvFront -- otherwise only
// in the arena).  `ll`: this frame's log-likelihood row in LDS.  The utterance's pool
// cursors are c.s.arena_next / link_next (TakeChunks).  Returns 0, kRedo (does not fit:
// nothing committed, the table is clean) or a fatal KD_UTT_* bit.
template <int M>
__device__ __forceinline__ uint32_t WaveFrame(FrameCtx<M> &c, WaveLds &w, const float *ll, int frame,
                                              WaveFront &fr, int &n, long long &base,
                                              uint32_t &best_key_out, WaveStats &ws) {
  static_assert(M == 1, "the wave tier runs the lattice decoders");
  CParams &p = c.p;
  Shared &s = c.s;
  const int lane = threadIdx.x & 63;
  const int n_in = n;
  const long long base_in = base;
  const bool has = lane < n_in;
  // ---- GetCutoff: no limits, or limits that cannot bind (WaveCutoffOk)
  const uint32_t my_key = has ? FloatKey(fr.cost) : kFInfKey;
  const uint32_t best_key = WaveMinU32(my_key);
  float adaptive_beam = p.beam;
  if (!(p.max_active == INT_MAX && p.min_active == 0))
    adaptive_beam = static_cast<float>(__longlong_as_double(0x7FF0000000000000ll) -
                                       static_cast<double>(KeyFloat(best_key)) + p.beam_delta);
  // ---- emitting arcs of every token (LatticeSimpleDecoder expands all of prev_toks_,
  // :376-401); the best token goes first so that the running cutoff starts tight
  // (faster-decoder.cc:176-189 bounds its cutoff the same way)
  const unsigned long long bm = __ballot(has && my_key == best_key);
  const int b = bm ? __builtin_ctzll(bm) : 0;
  int e_abeg = fr.abeg, e_deg = has ? fr.deg : 0;
  float e_cost = fr.cost;
  uint32_t e_src = static_cast<uint32_t>(lane);
  if (b != 0) {  // swap lanes 0 and b in the expansion order
    const int a0 = __shfl(e_abeg, 0, 64), ab = __shfl(e_abeg, b, 64);
    const int d0 = __shfl(e_deg, 0, 64), db = __shfl(e_deg, b, 64);
    const float c0 = __shfl(e_cost, 0, 64), cb = __shfl(e_cost, b, 64);
    if (lane == 0) { e_abeg = ab; e_deg = db; e_cost = cb; e_src = static_cast<uint32_t>(b); }
    if (lane == b) { e_abeg = a0; e_deg = d0; e_cost = c0; e_src = 0u; }
  }
  int q_n = 0, n_arcs = 0;
  uint32_t run_min = kFInfKey;
  if (!WaveExpand<false>(p, w, ll, e_abeg, e_deg, e_cost, e_src, adaptive_beam, 0.0f, run_min, q_n, n_arcs))
    return kRedo;
  const int q_emit = q_n;
  const float cutoff = KeyFloat(run_min) + adaptive_beam;  // float, :388-390
  const uint32_t cutkey = FloatKey(cutoff);
  WaveSync();
  if (lane < kWvSlots / 32) { w.wl[0][lane] = 0; w.wl[1][lane] = 0; }
  WaveSync();
  int n_live = 0, n_emit_links = 0;
  if (!WaveInsert<false>(w, 0, q_emit, cutkey, 0, n_live, n_emit_links)) {
    WaveClearTable(w);
    return kRedo;
  }
  // ---- ProcessNonemitting to a fixpoint (:122-191)
  int cur = 0, q_begin = q_emit, eps_arcs = 0, eps_exp = 0;
  while (true) {
    const uint32_t word = lane < kWvSlots / 32 ? w.wl[cur][lane] : 0u;
    if (__ballot(word != 0) == 0) break;
    WaveSync();
    if (lane < kWvSlots / 32) w.wl[cur ^ 1][lane] = 0;
#pragma unroll 1
    for (int j = 0; j < kWvSlots / 64; ++j) {
      const int slot = j * 64 + lane;
      const bool flagged = (w.wl[cur][slot >> 5] >> (slot & 31)) & 1u;
      if (__ballot(flagged) == 0) continue;
      const float cst = KeyFloat(w.cost[slot]);
      const bool ok = flagged && !(cst > cutoff);  // faster-decoder.cc:74
      const StateRec r = LoadState(p.states, ok ? w.key[slot] & kStateMask : static_cast<uint32_t>(p.start));
      eps_exp += __popcll(__ballot(ok));
      if (!WaveExpand<true>(p, w, ll, r.emit_end, ok ? r.arc_end - r.emit_end : 0, cst,
                            kWvEps | static_cast<uint32_t>(slot), 0.0f, cutoff, run_min, q_n, eps_arcs)) {
        WaveClearTable(w);
        return kRedo;
      }
    }
    WaveSync();
    if (!WaveInsert<true>(w, q_begin, q_n, 0u, cur ^ 1, n_live, n_emit_links)) {
      WaveClearTable(w);
      return kRedo;
    }
    q_begin = q_n;
    cur ^= 1;
  }
  // ---- commit: the frame's token block and room for its forward links (at most one per
  // queue entry below the cutoff) from this utterance's chunks of the pools
  const int n_q = q_begin, n_out = n_live;
  const int link_room = n_emit_links + (n_q - q_emit);
  if (Uni(s.arena_next) + n_out > Uni(s.arena_end) || Uni(s.link_next) + link_room > Uni(s.link_end)) {
    if (lane == 0) w.alloc_ok = static_cast<int32_t>(TakeChunks(p, s, c.utt, frame + 1, n_out, link_room));
    WaveSync();
    const uint32_t e = static_cast<uint32_t>(Uni(w.alloc_ok));
    if (e) {
      WaveClearTable(w);
      return e;
    }
  }
  const long long base_out = Uni(s.arena_next), lbase = Uni(s.link_next);
  // C1: live list -> token records; a slot's key word now holds its index in the block
  const bool next_wave = n_out <= kWvFront;
  uint32_t st0 = 0, ck0 = kFInfKey, best_next = kFInfKey;
  ArcRange next_rec{0, 0};
  for (int i0 = 0; i0 < n_out; i0 += 64) {
    const int i = i0 + lane;
    if (i < n_out) {
      const int h = w.live[i];
      const uint32_t ck = w.cost[h], st = w.key[h] & kStateMask;
      const long long g = base_out + i;
      ToG(p.tok_state)[g] = static_cast<int32_t>(st);
      ToG(p.tok_cost)[g] = static_cast<double>(KeyFloat(ck));
      ToG(p.tok_extra)[g] = 0.0f;  // lattice-simple-decoder.cc:91-95
      w.key[h] = static_cast<uint32_t>(i);
      best_next = ck < best_next ? ck : best_next;
      if (i0 == 0) { st0 = st; ck0 = ck; }
    }
  }
  // (the new frontier's arc ranges: requested here, used when the frame ends)
  if (next_wave) next_rec = LoadArcRange(p.states, lane < n_out ? st0 : static_cast<uint32_t>(p.start));
  WaveSync();
  // forward links (see FrameBack): emitting entries below the cutoff leave the previous
  // frame's tokens; epsilon entries stay inside this frame and are kept only if generated
  // from their source token's final cost (:164-165 deletes and regenerates)
  int n_emit_w = 0, n_eps_w = 0;
  for (int i0 = 0; i0 < n_q; i0 += 64) {
    const int i = i0 + lane, ii = i < n_q ? i : 0;
    const uint32_t h = w.q_h[ii], sr = w.q_src[ii], aidx = w.q_arc[ii], nk = w.q_nk[ii];
    const float qa = w.q_ac[ii], qg = w.q_gr[ii];
    const bool ok = i < n_q && h != kKeyEmpty, eps = i >= q_emit;
    const uint32_t hd = ok ? h : 0u, hs = ok && eps ? sr & (kWvSlots - 1) : 0u;
    const uint32_t kd = w.key[hd], ks = w.key[hs];
    const float cdf = KeyFloat(w.cost[hd]), csf = KeyFloat(w.cost[hs]);
    const bool we = ok && !eps, wp = ok && eps && qa == csf;
    const unsigned long long me = __ballot(we), mp = __ballot(wp);
    if (we || wp) {
      const long long at = we ? lbase + n_emit_w + LaneRank(me)
                              : lbase + n_emit_links + n_eps_w + LaneRank(mp);
      ToG(p.link_src)[at] = static_cast<int32_t>(we ? base_in + sr : base_out + ks);
      ToG(p.link_dst)[at] = static_cast<int32_t>(base_out + kd);
      ToG(p.link_arc)[at] = static_cast<int32_t>(aidx);
      ToG(p.link_ac)[at] = we ? qa : 0.0f;
      ToG(p.link_graph)[at] = qg;
      ToG(p.link_delta)[at] = KeyFloat(nk) - cdf;
    }
    n_emit_w += __popcll(me);
    n_eps_w += __popcll(mp);
  }
  WaveSync();
  // clear the table
  for (int i = lane; i < n_out; i += 64) {
    const int h = w.live[i];
    w.key[h] = kKeyEmpty;
    w.cost[h] = kWvDead;
  }
  if (lane == 0) {
    const long long fb = static_cast<long long>(c.utt) * (p.max_frames + 1);
    ToG(p.frame_base)[fb + frame + 1] = base_out;
    ToG(p.frame_count)[fb + frame + 1] = n_out;
    ToG(p.link_emit_base)[fb + frame] = lbase;
    ToG(p.link_emit_count)[fb + frame] = n_emit_links;
    ToG(p.link_eps_base)[fb + frame + 1] = lbase + n_emit_links;
    ToG(p.link_eps_count)[fb + frame + 1] = n_eps_w;
    s.arena_next = base_out + n_out;
    s.link_next = lbase + n_emit_links + n_eps_w;
  }
  ws.frames += 1;
  ws.tokens_in += n_in;
  ws.expanded += n_in;
  ws.arcs += n_arcs;
  ws.kept += q_emit;
  ws.tokens_out += n_out;
  ws.eps_arcs += eps_arcs;
  ws.eps_exp += eps_exp;
  ws.max_tokens = n_out > ws.max_tokens ? n_out : ws.max_tokens;
  ws.links += n_emit_links + n_eps_w;
  best_key_out = WaveMinU32(best_next);
  n = n_out;
  base = base_out;
  if (next_wave) {
    fr.st = st0;
    fr.cost = KeyFloat(ck0);
    fr.abeg = next_rec.arc_begin;
    fr.deg = next_rec.emit_end - next_rec.arc_begin;
  }
  WaveSync();
  return 0;
}

// The frontier of a wave-tier frame that follows a frame of another tier: arena -> lanes.
template <int M>
__device__ __forceinline__ void WaveLoadFront(CParams &p, long long base, int n, WaveFront &fr) {
  const int lane = threadIdx.x & 63;
  const bool has = lane < n;
  const uint32_t st = has ? static_cast<uint32_t>(ToG(p.tok_state)[base + lane]) : static_cast<uint32_t>(p.start);
  const double cst = has ? ToG(p.tok_cost)[base + lane] : 0.0;
  const ArcRange r = LoadArcRange(p.states, st);
  fr.st = st;
  fr.cost = static_cast<float>(cst);
  fr.abeg = r.arc_begin;
  fr.deg = r.emit_end - r.arc_begin;
}
