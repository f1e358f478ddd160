// kaldi-decoder_amd/csrc/kd_path_tail.h -- ReachedFinal and GetBestPath of the 1-best
// decoders (ref: /root/reference/kaldi-decoder/csrc/faster-decoder.cc:347-424,
// simple-decoder.cc:78-148), run by the search kernel's workgroup in the tail of an
// utterance.  Part of kd_search.hip (included inside its namespace, after
// kd_lattice_tail.h, whose TailShared / Stage it reuses).
//
// The last frame's tokens are scanned by the whole workgroup (ReachedFinal, the best
// token: cost, or cost + Final(state) when a final state is active; ties between equal
// totals go to the lowest state id -- the reference takes the first in HashList order,
// which no parallel search can reproduce; DESIGN.md "Order dependence").  One lane
// follows the back-pointers once, noting (arc, cost) of every token on the way in the
// workgroup's idle candidate-queue arrays; then every lane turns one of them into an
// output arc -- the linear LatticeArc FST before RemoveEpsLocal, in path order -- in
// the utterance's region of the output pool (pinned host memory).  Round 1 ran this as
// two more kernels with four read-backs and a prefix sum on the host in between, and
// those launches waited behind the persistent search workgroups of the other streams.

struct PathResult {  // what LatticeTail's r_* fields are for the lattice decoders
  int32_t len;            // arcs on the path; -1: no token survived (GetBestPath false)
  int32_t reached_final;
  float final_weight;     // Final(state) when reached and use_final_probs, else 0 (One)
  float final_rel;        // SimpleDecoder::FinalRelativeCost (simple-decoder.cc:78-100)
  long long out_off;
};

__device__ __noinline__ uint32_t PathTail(CParams *pp, int utt, int F, uint32_t flags) {
  CParams &p = *pp;
  TailShared &s = TailLds();
  const int tid = threadIdx.x;
  const bool use_final_probs = (flags & kUttPathNoFinal) == 0;
  const Stage st = TailStage(p, true);  // tok_g: arcs, arc_f / arc_at: the costs' two halves
  __shared__ unsigned long long sp_best, sp_bestc, sp_bestwf;
  __shared__ unsigned int sp_state;
  __shared__ int sp_final, sp_len;
  __shared__ long long sp_tok, sp_end;
  constexpr unsigned long long kKeyInfD = 0xFFF0000000000000ull;
  if (tid == 0) {
    sp_best = ~0ull; sp_state = 0xFFFFFFFFu; sp_final = 0; sp_tok = -1; sp_bestc = ~0ull;
    sp_bestwf = ~0ull; sp_len = -1; sp_end = -1;
    s.r_sizes[0] = -1; s.r_sizes[1] = 0; s.r_sizes[2] = 0; s.r_sizes[3] = 0;
    s.r_out_off = -1;
    s.r_final_rel = FInf();
  }
  __syncthreads();
  const long long fi = static_cast<long long>(utt) * (p.max_frames + 1) + F;
  const long long base = ToG(p.frame_base)[fi];
  const int n = ToG(p.frame_count)[fi];
  // ReachedFinal (:347-354); FinalRelativeCost's two minima (simple-decoder.cc:83-91)
  {
    int any_final = 0;
    unsigned long long bc = ~0ull, bwf = ~0ull;
    for (int i = tid; i < n; i += kBlock) {
      const double c = ToG(p.tok_cost)[base + i];
      const float fc = ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost;
      if (CostKey(c) != kKeyInfD && !(fc == FInf())) any_final = 1;
      const unsigned long long k1 = CostKey(c), k2 = CostKey(c + fc);
      bc = k1 < bc ? k1 : bc;
      bwf = k2 < bwf ? k2 : bwf;
    }
    bc = WaveMinU64(bc);
    bwf = WaveMinU64(bwf);
    if ((tid & 63) == 0) {
      if (bc != ~0ull) atomicMin(&sp_bestc, bc);
      if (bwf != ~0ull) atomicMin(&sp_bestwf, bwf);
    }
    if (any_final) sp_final = 1;
  }
  __syncthreads();
  const bool is_final = sp_final != 0;
  // best token (:367-384)
  {
    unsigned long long mk = ~0ull;
    for (int i = tid; i < n; i += kBlock) {
      double c = ToG(p.tok_cost)[base + i];
      if (is_final) {
        c = c + ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost;
        if (CostKey(c) == kKeyInfD) continue;  // this_cost != infinity
      }
      const unsigned long long k = CostKey(c);
      mk = k < mk ? k : mk;
    }
    mk = WaveMinU64(mk);
    if ((tid & 63) == 0 && mk != ~0ull) atomicMin(&sp_best, mk);
  }
  __syncthreads();
  const unsigned long long best = sp_best;
  if (best != ~0ull) {
    for (int i = tid; i < n; i += kBlock) {
      double c = ToG(p.tok_cost)[base + i];
      if (is_final) c = c + ToG(p.states)[ToG(p.tok_state)[base + i]].final_cost;
      if (CostKey(c) == best) atomicMin(&sp_state, static_cast<unsigned int>(ToG(p.tok_state)[base + i]));
    }
  }
  __syncthreads();
  if (best != ~0ull) {
    for (int i = tid; i < n; i += kBlock)
      if (static_cast<unsigned int>(ToG(p.tok_state)[base + i]) == sp_state) sp_tok = base + i;
  }
  __syncthreads();
  // the walk: one lane, one round trip per token (its three words are requested together)
  if (tid == 0) {
    const long long bt = sp_tok;
    int len = -1;
    if (bt >= 0) {
      len = 0;
      long long t = bt;
      while (t >= 0) {
        const int32_t a = ToG(p.tok_arc)[t];
        const long long pr = ToG(p.tok_prev)[t];
        const double c = ToG(p.tok_cost)[t];
        const float ac = p.simple ? ToG(p.tok_extra)[t] : 0.0f;
        if (len < st.cap) {
          st.tok_g[len] = static_cast<uint32_t>(a);
          st.tok_f[len] = __float_as_uint(ac);
          const unsigned long long cb = static_cast<unsigned long long>(__double_as_longlong(c));
          st.arc_at[len] = static_cast<uint32_t>(cb);
          st.arc_f[len] = static_cast<uint32_t>(cb >> 32);
        }
        if (a == static_cast<int32_t>(kArcNone)) break;  // the start token: its cost is entry `len`
        len++;
        t = pr;
      }
    }
    sp_len = len;
  }
  __syncthreads();
  const int len = sp_len;
  if (tid == 0) {
    float rel = FInf();  // no tokens: infinity (:82)
    if (n > 0) {
      rel = static_cast<float>(KeyCost(sp_bestwf) - KeyCost(sp_bestc));
      if (rel != rel) rel = FInf();  // NaN (:93-97)
    }
    s.r_final_rel = rel;
    s.r_sizes[0] = len;
    s.r_sizes[1] = is_final ? 1 : 0;
    float fw = 0.0f;
    if (len >= 0 && is_final && use_final_probs)
      fw = ToG(p.states)[ToG(p.tok_state)[sp_tok]].final_cost;  // :416-421
    s.r_sizes[2] = static_cast<int32_t>(__float_as_uint(fw));
  }
  if (len <= 0) { __syncthreads(); return 0; }
  // a path longer than the workgroup's staging queue (2 * table slots: 262 k arcs at the
  // default): the limit that was hit is the queue, not the output pool and not the token
  // table -- a status of its own, so that the host enlarges the queue (it grows with the
  // tables: GrowTables) and, where it cannot, says which limit this was
  if (len >= st.cap) { __syncthreads(); return KD_UTT_PATH_QUEUE_FULL; }
  if (tid == 0) {
    const unsigned long long words = 3ull * len;
    s.out_off = PoolTake(&p.ctl->out_top, p.out_base, p.lat.out_cap, words);
  }
  __syncthreads();
  const long long off = s.out_off;
  if (off < 0) return KD_UTT_OUT_FULL;
  KD_G unsigned long long *o = ToG(p.lat.out) + off;
  auto pack = [](uint32_t lo, uint32_t hi) { return static_cast<unsigned long long>(hi) << 32 | lo; };
  auto cost_at = [&](int k) {
    return __longlong_as_double(static_cast<long long>(
        static_cast<unsigned long long>(st.arc_f[k]) << 32 | st.arc_at[k]));
  };
  for (int j = tid; j < len; j += kBlock) {
    const int k = len - 1 - j;  // walk entry of path position j
    const ArcRec arc = LoadArcRec(p.arcs, st.tok_g[k]);
    // :394-396: tot = float(cost - prev.cost); ac = tot - graph.  SimpleDecoder keeps
    // the arc's own (graph, acoustic) pair instead (simple-decoder.h:93-96).
    const float tot_cost = cost_at(k) - cost_at(k + 1);
    const float graph_cost = arc.weight;
    const float ac_cost = p.simple ? __uint_as_float(st.tok_f[k]) : tot_cost - graph_cost;
    KD_G unsigned long long *w = o + 3ll * j;
    w[0] = pack(static_cast<uint32_t>(FullIlabel(p, arc, st.tok_g[k])), static_cast<uint32_t>(ToG(p.arc_olabel)[st.tok_g[k]]));
    w[1] = pack(arc.next & kStateMask, __float_as_uint(graph_cost));
    w[2] = pack(__float_as_uint(ac_cost), 0u);
  }
  __syncthreads();
  if (tid == 0) s.r_out_off = off;
  return 0;
}
