// oracle/kd_oracle.cc -- CPU restatement of kaldi-decoder's token-passing search.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under kaldi-decoder_amd/ may link, import
// or call this file; it exists so tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can check the HIP path against an independent CPU statement
// of the reference algorithm.
//
// PARITY UNPINNED.  The reference (k2-fsa/kaldi-decoder v0.3.0) ships no decoder
// tests, golden vectors or fixtures (only hash-list-test.cc and eigen-test.cc,
// /root/reference/kaldi-decoder/csrc/CMakeLists.txt:29-32) and its decoders
// cannot be built in this image: they include OpenFst / kaldifst / Eigen
// headers that cmake fetches from the network
// (/root/reference/cmake/kaldifst.cmake:4-5, cmake/eigen.cmake:4-6).  What IS
// pinned: (1) the HashList iteration order below is checked element for element
// against the reference's own header-only HashList, compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/; (2) 1-best results are
// checked against an independent exact (unpruned) numpy Viterbi in
// tests/viterbi_ref.py; (3) the three restated decoders are checked against one
// another (FasterDecoder == SimpleDecoder == shortest path through the
// LatticeSimpleDecoder raw lattice).
//
// Each function cites the reference lines it follows.  "ref:" paths are relative
// to /root/reference/kaldi-decoder/csrc/.
//
// Two search policies are provided for FasterDecoder:
//   policy 0  "reference order": HashList iteration order, cutoff tightened
//             while iterating, first-inserted wins ties -- the reference as is.
//   policy 1  "canonical": the order-free semantics the GPU implements: the
//             emitting cutoff is the final (tightest) value of the reference's
//             running cutoff, only tokens below it survive, ties are broken
//             by the lowest arc index.  Every token policy 1 keeps, policy 0
//             keeps too with the same cost (DESIGN.md "Order dependence").

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr double kInfD = std::numeric_limits<double>::infinity();
constexpr float kInfF = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------
// Graph: CSR copy of an fst::Fst<StdArc>; arcs stay in the FST's own order
// because the reference walks ArcIterator order and tests ilabel per arc.
// ---------------------------------------------------------------------------
struct Graph {
  int32_t num_states = 0, start = -1;
  std::vector<int32_t> row_ptr, ilabel, olabel, nextstate;
  std::vector<float> weight, final_cost;  // final_cost: +inf == Weight::Zero()
  std::vector<int32_t> num_ieps;          // fst.NumInputEpsilons(state)
  // index of each arc once every state's arcs are stably partitioned into
  // emitting-first order: the arc numbering of the GPU layout, used only as a
  // tie-break key in the canonical policy.
  std::vector<int32_t> canon_idx;
};

// ---------------------------------------------------------------------------
// HashList<int32, int32> restated (ref: hash-list.h:36-131, hash-list-inl.h).
// A singly linked list threaded through hash buckets: buckets are chained in
// the order they first became occupied, a bucket's elements are adjacent in
// the list, and a new bucket's first element is appended at the list tail
// (hash-list-inl.h:128-173).  Elements live in an index pool instead of
// pointer blocks; the free list only recycles storage and cannot change the
// iteration order.
// ---------------------------------------------------------------------------
struct HashList {
  static constexpr int32_t kNil = -1;
  struct Node { int32_t key, val, next; };
  struct Bucket { int64_t prev_bucket; int32_t last; };
  std::vector<Node> pool;
  std::vector<Bucket> buckets;
  int32_t head = kNil, free_head = kNil;
  int64_t tail_bucket = -1;
  size_t hash_size = 0;

  void SetSize(size_t n) {  // hash-list-inl.h:26-35: buckets only ever grow
    hash_size = n;
    if (n > buckets.size()) buckets.resize(n, Bucket{0, kNil});
  }
  size_t Size() const { return hash_size; }
  int32_t Clear() {  // hash-list-inl.h:38-51: empty the hash, hand the list over
    for (int64_t b = tail_bucket; b != -1; b = buckets[b].prev_bucket)
      buckets[b].last = kNil;
    tail_bucket = -1;
    int32_t h = head;
    head = kNil;
    return h;
  }
  int32_t GetList() const { return head; }
  void Delete(int32_t e) { pool[e].next = free_head; free_head = e; }
  int32_t New() {
    if (free_head != kNil) { int32_t e = free_head; free_head = pool[e].next; return e; }
    pool.push_back(Node{0, 0, kNil});
    return static_cast<int32_t>(pool.size()) - 1;
  }
  int32_t BucketHead(const Bucket &b) const {
    return b.prev_bucket == -1 ? head : pool[buckets[b.prev_bucket].last].next;
  }
  int32_t Find(int32_t key) const {  // hash-list-inl.h:65-83
    const Bucket &b = buckets[static_cast<size_t>(key) % hash_size];
    if (b.last == kNil) return kNil;
    int32_t stop = pool[b.last].next;
    for (int32_t e = BucketHead(b); e != stop; e = pool[e].next)
      if (pool[e].key == key) return e;
    return kNil;
  }
  // Returns the element holding `key`; inserts (key,val) if absent.
  int32_t Insert(int32_t key, int32_t val) {  // hash-list-inl.h:128-173
    size_t idx = static_cast<size_t>(key) % hash_size;
    Bucket &b = buckets[idx];
    if (b.last != kNil) {
      int32_t stop = pool[b.last].next;
      for (int32_t e = BucketHead(b); e != stop; e = pool[e].next)
        if (pool[e].key == key) return e;
    }
    int32_t e = New();
    Bucket &bb = buckets[idx];
    pool[e].key = key;
    pool[e].val = val;
    if (bb.last == kNil) {
      if (tail_bucket == -1) head = e;
      else pool[buckets[tail_bucket].last].next = e;
      pool[e].next = kNil;
      bb.last = e;
      bb.prev_bucket = tail_bucket;
      tail_bucket = static_cast<int64_t>(idx);
    } else {
      pool[e].next = pool[bb.last].next;
      pool[bb.last].next = e;
      bb.last = e;
    }
    return e;
  }
};

// ---------------------------------------------------------------------------
// Decodable: row-major [rows, cols] float matrix with a frame offset
// (ref: decodable-ctc.cc:22-38).
// ---------------------------------------------------------------------------
struct Decodable {
  const float *p = nullptr;
  int32_t rows = 0, cols = 0, offset = 0;
  float LogLikelihood(int32_t frame, int32_t index) const {
    return p[static_cast<int64_t>(frame - offset) * cols + index - 1];
  }
  int32_t NumFramesReady() const { return offset + rows; }
  bool IsLastFrame(int32_t frame) const { return frame == NumFramesReady() - 1; }
};

struct Stats {
  int64_t frames = 0, tokens_in = 0, tokens_expanded = 0, arcs_scanned = 0,
          arcs_emit_kept = 0, tokens_out = 0, eps_arcs_scanned = 0,
          eps_states_expanded = 0, max_tokens = 0;
};

// Best path as plain arrays (the linear FST GetBestPath builds, before
// RemoveEpsLocal which lives in kaldifst and is absent here).
struct Path {
  bool ok = false;
  std::vector<int32_t> ilabel, olabel, state;
  std::vector<float> graph_cost, ac_cost;
  float final_graph = 0.f, final_ac = 0.f;
};

// ---------------------------------------------------------------------------
// FasterDecoder restated (ref: faster-decoder.h:24-63,110-156, faster-decoder.cc)
// ---------------------------------------------------------------------------
struct FasterOpts {
  float beam = 16.f;
  int32_t max_active = std::numeric_limits<int32_t>::max();
  int32_t min_active = 20;
  float beam_delta = 0.5f, hash_ratio = 2.0f;
};

struct FasterDecoder {
  // Token (faster-decoder.h:110-156): the arc is referenced by index; tokens
  // live in an append-only arena (the reference ref-counts and frees them,
  // which changes memory use, not results).
  struct Token { int32_t arc, prev; double cost; };
  const Graph &g;
  FasterOpts opts;
  int policy;
  HashList toks;
  std::vector<Token> arena;
  std::vector<int32_t> queue;
  std::vector<float> tmp_array;
  int32_t num_frames_decoded = -1;
  Stats stats;
  // canonical policy frontier: state -> token index, rebuilt per frame
  std::unordered_map<int32_t, int32_t> cur;

  FasterDecoder(const Graph &graph, const FasterOpts &o, int pol)
      : g(graph), opts(o), policy(pol) {
    // faster-decoder.cc:21-32
    if (!(o.hash_ratio >= 1.0f)) throw std::runtime_error("hash_ratio >= 1.0");
    if (!(o.max_active > 1)) throw std::runtime_error("max_active > 1");
    if (!(o.min_active >= 0 && o.min_active < o.max_active))
      throw std::runtime_error("0 <= min_active < max_active");
    toks.SetSize(1000);
  }

  int32_t NextState(const Token &t) const { return t.arc < 0 ? g.start : g.nextstate[t.arc]; }

  void ClearToks(int32_t list) {  // faster-decoder.cc:34-40
    for (int32_t e = list, nx; e != HashList::kNil; e = nx) {
      nx = toks.pool[e].next;
      toks.Delete(e);
    }
  }

  int32_t NewToken(int32_t arc, float ac_cost, int32_t prev, bool emitting) {
    // faster-decoder.h:120-139: cost = prev.cost + weight (+ ac_cost)
    double c;
    float w = arc < 0 ? 0.f : g.weight[arc];
    if (prev >= 0) c = emitting ? arena[prev].cost + w + ac_cost : arena[prev].cost + w;
    else c = emitting ? static_cast<double>(w + ac_cost) : static_cast<double>(w);
    arena.push_back(Token{arc, prev, c});
    return static_cast<int32_t>(arena.size()) - 1;
  }

  void InitDecoding() {  // faster-decoder.cc:42-56
    ClearToks(toks.Clear());
    arena.clear();
    cur.clear();
    stats = Stats();
    if (g.start < 0) throw std::runtime_error("graph has no start state");
    int32_t t = NewToken(-1, 0.f, -1, false);
    if (policy == 0) {
      toks.Insert(g.start, t);
      ProcessNonemitting(std::numeric_limits<float>::max());
    } else {
      cur[g.start] = t;
      CanonicalClosure(std::numeric_limits<float>::max());
    }
    num_frames_decoded = 0;
  }

  void AdvanceDecoding(const Decodable &d, int32_t max_num_frames) {
    // faster-decoder.cc:126-152
    if (num_frames_decoded < 0)
      throw std::runtime_error("You must call InitDecoding() before AdvanceDecoding()");
    int32_t ready = d.NumFramesReady();
    if (!(ready >= num_frames_decoded)) throw std::runtime_error("num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, num_frames_decoded + max_num_frames);
    while (num_frames_decoded < target) {
      if (policy == 0) {
        double cutoff = ProcessEmitting(d);
        ProcessNonemitting(cutoff);
      } else {
        CanonicalFrame(d);
      }
    }
  }

  // -- policy 0 -------------------------------------------------------------
  double GetCutoff(int32_t list_head, size_t *tok_count, float *adaptive_beam,
                   int32_t *best_elem) {  // faster-decoder.cc:244-336
    double best_cost = kInfD;
    size_t count = 0;
    if (opts.max_active == std::numeric_limits<int32_t>::max() && opts.min_active == 0) {
      for (int32_t e = list_head; e != HashList::kNil; e = toks.pool[e].next, ++count) {
        double w = arena[toks.pool[e].val].cost;
        if (w < best_cost) { best_cost = w; *best_elem = e; }
      }
      *tok_count = count;
      *adaptive_beam = opts.beam;
      return best_cost + opts.beam;
    }
    tmp_array.clear();
    for (int32_t e = list_head; e != HashList::kNil; e = toks.pool[e].next, ++count) {
      double w = arena[toks.pool[e].val].cost;
      tmp_array.push_back(static_cast<float>(w));  // narrowed to float, :278
      if (w < best_cost) { best_cost = w; *best_elem = e; }
    }
    *tok_count = count;
    return CutoffFromArray(best_cost, adaptive_beam);
  }

  // faster-decoder.cc:293-335, shared by both policies (value-based only).
  double CutoffFromArray(double best_cost, float *adaptive_beam) {
    double beam_cutoff = best_cost + opts.beam, min_active_cutoff = kInfD,
           max_active_cutoff = kInfD;
    if (tmp_array.size() > static_cast<size_t>(opts.max_active)) {
      std::nth_element(tmp_array.begin(), tmp_array.begin() + opts.max_active, tmp_array.end());
      max_active_cutoff = tmp_array[opts.max_active];
    }
    if (max_active_cutoff < beam_cutoff) {
      *adaptive_beam = max_active_cutoff - best_cost + opts.beam_delta;
      return max_active_cutoff;
    }
    if (tmp_array.size() > static_cast<size_t>(opts.min_active)) {
      if (opts.min_active == 0) {
        min_active_cutoff = best_cost;
      } else {
        std::nth_element(tmp_array.begin(), tmp_array.begin() + opts.min_active,
                         tmp_array.size() > static_cast<size_t>(opts.max_active)
                             ? tmp_array.begin() + opts.max_active
                             : tmp_array.end());
        min_active_cutoff = tmp_array[opts.min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) {
      *adaptive_beam = min_active_cutoff - best_cost + opts.beam_delta;
      return min_active_cutoff;
    }
    *adaptive_beam = opts.beam;
    return beam_cutoff;
  }

  double ProcessEmitting(const Decodable &d) {  // faster-decoder.cc:155-241
    int32_t frame = num_frames_decoded;
    int32_t last_toks = toks.Clear();
    size_t tok_cnt = 0;
    float adaptive_beam = 0.f;
    int32_t best_elem = HashList::kNil;
    double weight_cutoff = GetCutoff(last_toks, &tok_cnt, &adaptive_beam, &best_elem);
    {  // PossiblyResizeHash, :338-345
      size_t new_sz = static_cast<size_t>(static_cast<float>(tok_cnt) * opts.hash_ratio);
      if (new_sz > toks.Size()) toks.SetSize(new_sz);
    }
    double next_weight_cutoff = kInfD;
    if (best_elem != HashList::kNil) {  // :176-189 bound from the best token
      int32_t state = toks.pool[best_elem].key;
      const Token &tok = arena[toks.pool[best_elem].val];
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float ac_cost = -1 * d.LogLikelihood(frame, g.ilabel[a]);
        double new_weight = g.weight[a] + tok.cost + ac_cost;
        if (new_weight + adaptive_beam < next_weight_cutoff)
          next_weight_cutoff = new_weight + adaptive_beam;
      }
    }
    stats.frames++;
    stats.tokens_in += static_cast<int64_t>(tok_cnt);
    stats.max_tokens = std::max<int64_t>(stats.max_tokens, tok_cnt);
    for (int32_t e = last_toks, e_tail; e != HashList::kNil; e = e_tail) {  // :196-237
      int32_t state = toks.pool[e].key;
      int32_t ti = toks.pool[e].val;
      double tcost = arena[ti].cost;
      if (tcost < weight_cutoff) {
        stats.tokens_expanded++;
        for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
          if (g.ilabel[a] == 0) continue;
          stats.arcs_scanned++;
          float ac_cost = -1 * d.LogLikelihood(frame, g.ilabel[a]);
          double new_weight = g.weight[a] + tcost + ac_cost;
          if (new_weight < next_weight_cutoff) {
            stats.arcs_emit_kept++;
            int32_t nt = NewToken(a, ac_cost, ti, true);
            int32_t e_found = toks.Insert(g.nextstate[a], nt);
            if (new_weight + adaptive_beam < next_weight_cutoff)
              next_weight_cutoff = new_weight + adaptive_beam;
            if (toks.pool[e_found].val != nt) {
              if (arena[toks.pool[e_found].val].cost > arena[nt].cost)  // strictly better
                toks.pool[e_found].val = nt;
              else
                arena.pop_back();  // the new token is dropped
            }
          }
        }
      }
      e_tail = toks.pool[e].next;
      toks.Delete(e);
    }
    num_frames_decoded++;
    return next_weight_cutoff;
  }

  void ProcessNonemitting(double cutoff) {  // faster-decoder.cc:59-119
    for (int32_t e = toks.GetList(); e != HashList::kNil; e = toks.pool[e].next)
      queue.push_back(e);
    while (!queue.empty()) {
      int32_t e = queue.back();
      queue.pop_back();
      int32_t state = toks.pool[e].key;
      int32_t ti = toks.pool[e].val;
      if (arena[ti].cost > cutoff) continue;
      if (g.num_ieps[state] != 0) stats.eps_states_expanded++;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        stats.eps_arcs_scanned++;
        int32_t nt = NewToken(a, 0.f, ti, false);
        if (arena[nt].cost > cutoff) { arena.pop_back(); continue; }
        int32_t e_found = toks.Insert(g.nextstate[a], nt);
        if (toks.pool[e_found].val == nt) { queue.push_back(e_found); continue; }
        if (arena[toks.pool[e_found].val].cost > arena[nt].cost) {
          toks.pool[e_found].val = nt;
          queue.push_back(e_found);
        } else {
          arena.pop_back();
        }
      }
    }
    size_t n = 0;
    for (int32_t e = toks.GetList(); e != HashList::kNil; e = toks.pool[e].next) ++n;
    stats.tokens_out += static_cast<int64_t>(n);
  }

  // -- policy 1 -------------------------------------------------------------
  // One frame of the order-free semantics.  Candidate costs use the same
  // expressions as policy 0, so any token both policies keep has the same
  // double cost bit for bit.
  void CanonicalFrame(const Decodable &d) {
    int32_t frame = num_frames_decoded;
    // GetCutoff by value (faster-decoder.cc:244-336)
    double best_cost = kInfD;
    tmp_array.clear();
    bool unconstrained = opts.max_active == std::numeric_limits<int32_t>::max() &&
                         opts.min_active == 0;
    for (auto &kv : cur) {
      double w = arena[kv.second].cost;
      if (!unconstrained) tmp_array.push_back(static_cast<float>(w));
      if (w < best_cost) best_cost = w;
    }
    float adaptive_beam = opts.beam;
    double weight_cutoff = unconstrained ? best_cost + opts.beam
                                         : CutoffFromArray(best_cost, &adaptive_beam);
    stats.frames++;
    stats.tokens_in += static_cast<int64_t>(cur.size());
    stats.max_tokens = std::max<int64_t>(stats.max_tokens, cur.size());
    // pass A: the value the reference's running cutoff ends at
    double best_new = kInfD;
    std::vector<std::pair<int32_t, int32_t>> src;  // (state, token) expanded
    for (auto &kv : cur) {
      double tcost = arena[kv.second].cost;
      if (!(tcost < weight_cutoff)) continue;
      src.push_back({kv.first, kv.second});
      stats.tokens_expanded++;
      for (int32_t a = g.row_ptr[kv.first]; a < g.row_ptr[kv.first + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        stats.arcs_scanned++;
        float ac_cost = -1 * d.LogLikelihood(frame, g.ilabel[a]);
        double new_weight = g.weight[a] + tcost + ac_cost;
        if (new_weight < best_new) best_new = new_weight;
      }
    }
    double next_cutoff = best_new + adaptive_beam;
    // pass B: keep candidates strictly below it; min cost, then lowest arc
    struct Best { double cost; int32_t arc, prev; float ac; };
    std::unordered_map<int32_t, Best> nxt;
    for (auto &sp : src) {
      double tcost = arena[sp.second].cost;
      for (int32_t a = g.row_ptr[sp.first]; a < g.row_ptr[sp.first + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float ac_cost = -1 * d.LogLikelihood(frame, g.ilabel[a]);
        double new_weight = g.weight[a] + tcost + ac_cost;
        if (!(new_weight < next_cutoff)) continue;
        stats.arcs_emit_kept++;
        // the token's stored cost, faster-decoder.h:126 (same value: IEEE
        // addition is commutative, association is unchanged)
        double c = tcost + g.weight[a] + ac_cost;
        auto it = nxt.find(g.nextstate[a]);
        if (it == nxt.end()) nxt.emplace(g.nextstate[a], Best{c, a, sp.second, ac_cost});
        else if (c < it->second.cost ||
                 (c == it->second.cost && g.canon_idx[a] < g.canon_idx[it->second.arc]))
          it->second = Best{c, a, sp.second, ac_cost};
      }
    }
    cur.clear();
    for (auto &kv : nxt) {
      arena.push_back(Token{kv.second.arc, kv.second.prev, kv.second.cost});
      cur[kv.first] = static_cast<int32_t>(arena.size()) - 1;
    }
    num_frames_decoded++;
    CanonicalClosure(next_cutoff);
  }

  // Epsilon closure to a fixpoint with a fixed cutoff (faster-decoder.cc:59-119
  // without its order dependence): expand tokens with cost <= cutoff, keep
  // results <= cutoff, strict improvement replaces; afterwards every token's
  // back-pointer is the lowest-index arc that achieves its final cost.
  void CanonicalClosure(double cutoff) {
    std::vector<int32_t> work;
    for (auto &kv : cur) if (g.num_ieps[kv.first] != 0) work.push_back(kv.first);
    std::sort(work.begin(), work.end());
    while (!work.empty()) {
      int32_t state = work.back();
      work.pop_back();
      int32_t ti = cur[state];
      double tcost = arena[ti].cost;
      if (tcost > cutoff) continue;
      stats.eps_states_expanded++;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        stats.eps_arcs_scanned++;
        double c = tcost + g.weight[a];
        if (c > cutoff) continue;
        int32_t ns = g.nextstate[a];
        auto it = cur.find(ns);
        if (it == cur.end()) {
          arena.push_back(Token{a, ti, c});
          cur[ns] = static_cast<int32_t>(arena.size()) - 1;
          if (g.num_ieps[ns] != 0) work.push_back(ns);
        } else if (c < arena[it->second].cost) {
          arena.push_back(Token{a, ti, c});
          it->second = static_cast<int32_t>(arena.size()) - 1;
          if (g.num_ieps[ns] != 0) work.push_back(ns);
        }
      }
    }
    // tie-break pass over epsilon arcs: lowest canonical arc index among the
    // arcs (emitting winner already stored, or epsilon) that achieve the cost.
    for (auto &kv : cur) {
      int32_t state = kv.first;
      double tcost = arena[kv.second].cost;
      if (tcost > cutoff) continue;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        double c = tcost + g.weight[a];
        auto it = cur.find(g.nextstate[a]);
        if (it == cur.end()) continue;
        Token &dst = arena[it->second];
        if (c == dst.cost && dst.arc >= 0 && g.canon_idx[a] < g.canon_idx[dst.arc]) {
          // in place: a frame's token is one record, and this frame's other
          // tokens must keep pointing at it
          dst.arc = a;
          dst.prev = kv.second;
        }
      }
    }
    stats.tokens_out += static_cast<int64_t>(cur.size());
  }

  // -- results ---------------------------------------------------------------
  template <class F> void ForEachToken(F f) const {
    if (policy == 0) {
      for (int32_t e = toks.GetList(); e != HashList::kNil; e = toks.pool[e].next)
        f(toks.pool[e].key, toks.pool[e].val);
    } else {
      std::vector<std::pair<int32_t, int32_t>> v(cur.begin(), cur.end());
      std::sort(v.begin(), v.end());  // canonical tie rule: lowest state first
      for (auto &kv : v) f(kv.first, kv.second);
    }
  }

  bool ReachedFinal() const {  // faster-decoder.cc:347-354
    bool r = false;
    ForEachToken([&](int32_t state, int32_t ti) {
      if (arena[ti].cost != kInfD && g.final_cost[state] != kInfF) r = true;
    });
    return r;
  }

  Path GetBestPath(bool use_final_probs) const {  // faster-decoder.cc:356-424
    Path out;
    int32_t best_tok = -1;
    bool is_final = ReachedFinal();
    if (!is_final) {
      ForEachToken([&](int32_t, int32_t ti) {
        if (best_tok < 0 || arena[best_tok].cost > arena[ti].cost) best_tok = ti;
      });
    } else {
      double best_cost = kInfD;
      ForEachToken([&](int32_t state, int32_t ti) {
        double this_cost = arena[ti].cost + g.final_cost[state];
        if (this_cost < best_cost && this_cost != kInfD) { best_cost = this_cost; best_tok = ti; }
      });
    }
    if (best_tok < 0) return out;
    for (int32_t t = best_tok; t >= 0; t = arena[t].prev) {
      const Token &tok = arena[t];
      if (tok.arc < 0) break;  // the dummy start token carries no information
      float tot_cost = tok.cost - (tok.prev >= 0 ? arena[tok.prev].cost : 0.0);
      float graph_cost = g.weight[tok.arc];
      float ac_cost = tot_cost - graph_cost;
      out.ilabel.push_back(g.ilabel[tok.arc]);
      out.olabel.push_back(g.olabel[tok.arc]);
      out.state.push_back(g.nextstate[tok.arc]);
      out.graph_cost.push_back(graph_cost);
      out.ac_cost.push_back(ac_cost);
    }
    std::reverse(out.ilabel.begin(), out.ilabel.end());
    std::reverse(out.olabel.begin(), out.olabel.end());
    std::reverse(out.state.begin(), out.state.end());
    std::reverse(out.graph_cost.begin(), out.graph_cost.end());
    std::reverse(out.ac_cost.begin(), out.ac_cost.end());
    if (is_final && use_final_probs) out.final_graph = g.final_cost[NextState(arena[best_tok])];
    out.ok = true;
    return out;
  }
};

// ---------------------------------------------------------------------------
// SimpleDecoder restated (ref: simple-decoder.h:24-138, simple-decoder.cc)
// ---------------------------------------------------------------------------
struct SimpleDecoder {
  struct Token { int32_t arc, prev; float ac; double cost; };
  const Graph &g;
  float beam;
  std::vector<Token> arena;
  std::unordered_map<int32_t, int32_t> cur_toks, prev_toks;
  int32_t num_frames_decoded = -1;

  SimpleDecoder(const Graph &graph, float b) : g(graph), beam(b) {}

  int32_t NewToken(int32_t arc, float ac, int32_t prev) {
    // simple-decoder.h:93-105: cost = prev.cost + (weight + ac) -- float sum first
    float w = arc < 0 ? 0.f : g.weight[arc];
    double c = prev >= 0 ? arena[prev].cost + (w + ac) : static_cast<double>(w + ac);
    arena.push_back(Token{arc, prev, ac, c});
    return static_cast<int32_t>(arena.size()) - 1;
  }
  void InitDecoding() {  // simple-decoder.cc:30-41
    cur_toks.clear();
    prev_toks.clear();
    arena.clear();
    if (g.start < 0) throw std::runtime_error("graph has no start state");
    cur_toks[g.start] = NewToken(-1, 0.f, -1);
    num_frames_decoded = 0;
    ProcessNonemitting();
  }
  void AdvanceDecoding(const Decodable &d, int32_t max_num_frames) {  // :43-67
    if (num_frames_decoded < 0)
      throw std::runtime_error("You must call InitDecoding() before AdvanceDecoding()");
    int32_t ready = d.NumFramesReady();
    if (!(ready >= num_frames_decoded)) throw std::runtime_error("num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, num_frames_decoded + max_num_frames);
    while (num_frames_decoded < target) {
      prev_toks.clear();
      cur_toks.swap(prev_toks);
      ProcessEmitting(d);
      ProcessNonemitting();
      PruneToks();
    }
  }
  void ProcessEmitting(const Decodable &d) {  // :150-193
    int32_t frame = num_frames_decoded;
    double cutoff = kInfF;
    for (auto &kv : prev_toks) {
      int32_t state = kv.first, ti = kv.second;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float acoustic_cost = -d.LogLikelihood(frame, g.ilabel[a]);
        double total_cost = arena[ti].cost + g.weight[a] + acoustic_cost;
        if (total_cost >= cutoff) continue;
        if (total_cost + beam < cutoff) cutoff = total_cost + beam;
        int32_t nt = NewToken(a, acoustic_cost, ti);
        auto it = cur_toks.find(g.nextstate[a]);
        if (it == cur_toks.end()) cur_toks[g.nextstate[a]] = nt;
        else if (arena[it->second].cost > arena[nt].cost) it->second = nt;
      }
    }
    num_frames_decoded++;
  }
  void ProcessNonemitting() {  // :195-241
    std::vector<int32_t> queue;
    double best_cost = kInfD;
    for (auto &kv : cur_toks) {
      queue.push_back(kv.first);
      best_cost = std::min(best_cost, arena[kv.second].cost);
    }
    double cutoff = best_cost + beam;
    while (!queue.empty()) {
      int32_t state = queue.back();
      queue.pop_back();
      int32_t ti = cur_toks[state];
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        int32_t nt = NewToken(a, 0.f, ti);
        if (arena[nt].cost > cutoff) { arena.pop_back(); continue; }
        auto it = cur_toks.find(g.nextstate[a]);
        if (it == cur_toks.end()) {
          cur_toks[g.nextstate[a]] = nt;
          queue.push_back(g.nextstate[a]);
        } else if (arena[it->second].cost > arena[nt].cost) {
          it->second = nt;
          queue.push_back(g.nextstate[a]);
        } else {
          arena.pop_back();
        }
      }
    }
  }
  void PruneToks() {  // :252-281
    if (cur_toks.empty()) return;
    double best_cost = kInfD;
    for (auto &kv : cur_toks) best_cost = std::min(best_cost, arena[kv.second].cost);
    std::vector<int32_t> retained;
    double cutoff = best_cost + beam;
    for (auto &kv : cur_toks) if (arena[kv.second].cost < cutoff) retained.push_back(kv.first);
    std::unordered_map<int32_t, int32_t> tmp;
    for (int32_t s : retained) tmp[s] = cur_toks[s];
    tmp.swap(cur_toks);
  }
  bool ReachedFinal() const {  // :69-76
    for (auto &kv : cur_toks)
      if (arena[kv.second].cost != kInfF && g.final_cost[kv.first] != kInfF) return true;
    return false;
  }
  float FinalRelativeCost() const {  // :78-100
    if (cur_toks.empty()) return kInfD;
    double best_cost = kInfD, best_cost_with_final = kInfD;
    for (auto &kv : cur_toks) {
      best_cost = std::min(best_cost, arena[kv.second].cost);
      best_cost_with_final =
          std::min(best_cost_with_final, arena[kv.second].cost + g.final_cost[kv.first]);
    }
    float extra_cost = best_cost_with_final - best_cost;
    if (extra_cost != extra_cost) return kInfD;
    return extra_cost;
  }
  Path GetBestPath(bool use_final_probs) const {  // :104-148
    Path out;
    int32_t best_tok = -1;
    bool is_final = ReachedFinal();
    if (!is_final) {
      for (auto &kv : cur_toks)
        if (best_tok < 0 || arena[best_tok].cost > arena[kv.second].cost) best_tok = kv.second;
    } else {
      double best_cost = kInfD;
      for (auto &kv : cur_toks) {
        double this_cost = arena[kv.second].cost + g.final_cost[kv.first];
        if (this_cost != kInfD && this_cost < best_cost) { best_cost = this_cost; best_tok = kv.second; }
      }
    }
    if (best_tok < 0) return out;
    for (int32_t t = best_tok; t >= 0 && arena[t].arc >= 0; t = arena[t].prev) {
      const Token &tok = arena[t];
      out.ilabel.push_back(g.ilabel[tok.arc]);
      out.olabel.push_back(g.olabel[tok.arc]);
      out.state.push_back(g.nextstate[tok.arc]);
      out.graph_cost.push_back(g.weight[tok.arc]);
      out.ac_cost.push_back(tok.ac);
    }
    std::reverse(out.ilabel.begin(), out.ilabel.end());
    std::reverse(out.olabel.begin(), out.olabel.end());
    std::reverse(out.state.begin(), out.state.end());
    std::reverse(out.graph_cost.begin(), out.graph_cost.end());
    std::reverse(out.ac_cost.begin(), out.ac_cost.end());
    if (is_final && use_final_probs) {
      const Token &bt = arena[best_tok];
      out.final_graph = g.final_cost[bt.arc < 0 ? g.start : g.nextstate[bt.arc]];
    }
    out.ok = true;
    return out;
  }
};


// ---------------------------------------------------------------------------
// Canonical (order-free) SimpleDecoder: what the GPU implements for it.
// Same values as SimpleDecoder above, without the dependence on unordered_map
// order: the emitting cutoff is min total_cost + beam (the value the running
// cutoff of simple-decoder.cc:170-176 ends at), ties take the lowest arc index,
// final-state ties the lowest state.
// ---------------------------------------------------------------------------
struct CanonSimpleDecoder {
  struct Token { int32_t arc, prev; float ac; double cost; };
  const Graph &g;
  float beam;
  std::vector<Token> arena;
  std::unordered_map<int32_t, int32_t> cur;  // state -> token, after PruneToks
  int32_t num_frames_decoded = -1;

  CanonSimpleDecoder(const Graph &graph, float b) : g(graph), beam(b) {}

  void Closure() {  // simple-decoder.cc:195-241
    double best_cost = kInfD;
    for (auto &kv : cur) best_cost = std::min(best_cost, arena[kv.second].cost);
    const double cutoff = best_cost + beam;
    std::vector<int32_t> work;
    for (auto &kv : cur) if (g.num_ieps[kv.first] != 0) work.push_back(kv.first);
    while (!work.empty()) {
      int32_t state = work.back();
      work.pop_back();
      int32_t ti = cur[state];
      double tcost = arena[ti].cost;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        double c = tcost + (g.weight[a] + 0.0f);
        if (c > cutoff) continue;
        auto it = cur.find(g.nextstate[a]);
        if (it == cur.end()) {
          arena.push_back(Token{a, ti, 0.f, c});
          cur[g.nextstate[a]] = static_cast<int32_t>(arena.size()) - 1;
          if (g.num_ieps[g.nextstate[a]]) work.push_back(g.nextstate[a]);
        } else if (c < arena[it->second].cost) {
          arena.push_back(Token{a, ti, 0.f, c});
          it->second = static_cast<int32_t>(arena.size()) - 1;
          if (g.num_ieps[g.nextstate[a]]) work.push_back(g.nextstate[a]);
        }
      }
    }
    for (auto &kv : cur) {  // tie-break: lowest canonical arc among achievers
      double tcost = arena[kv.second].cost;
      for (int32_t a = g.row_ptr[kv.first]; a < g.row_ptr[kv.first + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        double c = tcost + (g.weight[a] + 0.0f);
        if (c > cutoff) continue;
        auto it = cur.find(g.nextstate[a]);
        if (it == cur.end()) continue;
        Token &dst = arena[it->second];
        if (c == dst.cost && dst.arc >= 0 && g.canon_idx[a] < g.canon_idx[dst.arc]) {
          dst.arc = a; dst.prev = kv.second; dst.ac = 0.f;
        }
      }
    }
  }
  void InitDecoding() {
    arena.clear();
    cur.clear();
    if (g.start < 0) throw std::runtime_error("graph has no start state");
    arena.push_back(Token{-1, -1, 0.f, 0.0});
    cur[g.start] = 0;
    num_frames_decoded = 0;
    Closure();
  }
  void AdvanceDecoding(const Decodable &d, int32_t max_num_frames) {
    if (num_frames_decoded < 0)
      throw std::runtime_error("You must call InitDecoding() before AdvanceDecoding()");
    int32_t ready = d.NumFramesReady();
    if (!(ready >= num_frames_decoded)) throw std::runtime_error("num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, num_frames_decoded + max_num_frames);
    while (num_frames_decoded < target) {
      const int32_t frame = num_frames_decoded;
      double best_total = kInfD;
      for (auto &kv : cur)
        for (int32_t a = g.row_ptr[kv.first]; a < g.row_ptr[kv.first + 1]; ++a) {
          if (g.ilabel[a] == 0) continue;
          float ac = -d.LogLikelihood(frame, g.ilabel[a]);
          double total = arena[kv.second].cost + g.weight[a] + ac;
          best_total = std::min(best_total, total);
        }
      const double cutoff = best_total + beam;
      struct Best { double cost; int32_t arc, prev; float ac; };
      std::unordered_map<int32_t, Best> nxt;
      for (auto &kv : cur)
        for (int32_t a = g.row_ptr[kv.first]; a < g.row_ptr[kv.first + 1]; ++a) {
          if (g.ilabel[a] == 0) continue;
          float ac = -d.LogLikelihood(frame, g.ilabel[a]);
          double total = arena[kv.second].cost + g.weight[a] + ac;
          if (total >= cutoff) continue;  // :170
          double c = arena[kv.second].cost + (g.weight[a] + ac);  // simple-decoder.h:99
          auto it = nxt.find(g.nextstate[a]);
          if (it == nxt.end()) nxt.emplace(g.nextstate[a], Best{c, a, kv.second, ac});
          else if (c < it->second.cost ||
                   (c == it->second.cost && g.canon_idx[a] < g.canon_idx[it->second.arc]))
            it->second = Best{c, a, kv.second, ac};
        }
      cur.clear();
      for (auto &kv : nxt) {
        arena.push_back(Token{kv.second.arc, kv.second.prev, kv.second.ac, kv.second.cost});
        cur[kv.first] = static_cast<int32_t>(arena.size()) - 1;
      }
      num_frames_decoded++;
      Closure();
      // PruneToks, :252-281
      if (!cur.empty()) {
        double best = kInfD;
        for (auto &kv : cur) best = std::min(best, arena[kv.second].cost);
        const double pc = best + beam;
        for (auto it = cur.begin(); it != cur.end();)
          if (arena[it->second].cost < pc) ++it; else it = cur.erase(it);
      }
    }
  }
  std::vector<std::pair<int32_t, int32_t>> Sorted() const {
    std::vector<std::pair<int32_t, int32_t>> v(cur.begin(), cur.end());
    std::sort(v.begin(), v.end());
    return v;
  }
  bool ReachedFinal() const {
    for (auto &kv : cur)
      if (arena[kv.second].cost != kInfF && g.final_cost[kv.first] != kInfF) return true;
    return false;
  }
  float FinalRelativeCost() const {
    if (cur.empty()) return kInfD;
    double best_cost = kInfD, best_wf = kInfD;
    for (auto &kv : cur) {
      best_cost = std::min(best_cost, arena[kv.second].cost);
      best_wf = std::min(best_wf, arena[kv.second].cost + g.final_cost[kv.first]);
    }
    float extra = best_wf - best_cost;
    if (extra != extra) return kInfD;
    return extra;
  }
  Path GetBestPath(bool use_final_probs) const {
    Path out;
    int32_t best_tok = -1;
    bool is_final = ReachedFinal();
    double best_cost = kInfD;
    for (auto &kv : Sorted()) {
      double c = arena[kv.second].cost + (is_final ? static_cast<double>(g.final_cost[kv.first]) : 0.0);
      if (is_final) { if (c != kInfD && c < best_cost) { best_cost = c; best_tok = kv.second; } }
      else if (best_tok < 0 || c < best_cost) { best_cost = c; best_tok = kv.second; }
    }
    if (best_tok < 0) return out;
    for (int32_t t = best_tok; t >= 0 && arena[t].arc >= 0; t = arena[t].prev) {
      const Token &tok = arena[t];
      out.ilabel.push_back(g.ilabel[tok.arc]);
      out.olabel.push_back(g.olabel[tok.arc]);
      out.state.push_back(g.nextstate[tok.arc]);
      out.graph_cost.push_back(g.weight[tok.arc]);
      out.ac_cost.push_back(tok.ac);
    }
    std::reverse(out.ilabel.begin(), out.ilabel.end());
    std::reverse(out.olabel.begin(), out.olabel.end());
    std::reverse(out.state.begin(), out.state.end());
    std::reverse(out.graph_cost.begin(), out.graph_cost.end());
    std::reverse(out.ac_cost.begin(), out.ac_cost.end());
    if (is_final && use_final_probs) {
      const Token &bt = arena[best_tok];
      out.final_graph = g.final_cost[bt.arc < 0 ? g.start : g.nextstate[bt.arc]];
    }
    out.ok = true;
    return out;
  }
};

// ---------------------------------------------------------------------------
// Lattice output (what GetRawLattice builds: fst::VectorFst<LatticeArc>)
// ---------------------------------------------------------------------------
struct Lattice {
  int32_t num_states = 0, start = -1;
  std::vector<int32_t> src, dst, ilabel, olabel;
  std::vector<float> graph_cost, ac_cost;
  std::vector<int32_t> final_state;
  std::vector<float> final_graph, final_ac;
  std::vector<int32_t> state_frame;  // frame-plus-one of each state (extra info)
  std::vector<int32_t> state_graph;  // graph state of each lattice state (extra info)
};

// kaldi-math.h:36-44
inline bool ApproxEqual(float a, float b, float relative_tolerance = 0.001f) {
  if (a == b) return true;
  float diff = std::abs(a - b);
  if (diff == kInfF || diff != diff) return false;
  return diff <= relative_tolerance * (std::abs(a) + std::abs(b));
}

// ---------------------------------------------------------------------------
// LatticeSimpleDecoder restated (ref: lattice-simple-decoder.h/.cc)
// ---------------------------------------------------------------------------
struct LatticeSimpleOpts {
  float beam = 16.f, lattice_beam = 10.f;
  int32_t prune_interval = 25;
  float prune_scale = 0.1f;
};

struct LatticeSimpleDecoder {
  struct Token;
  struct ForwardLink {
    Token *next_tok; int32_t ilabel, olabel; float graph_cost, acoustic_cost; ForwardLink *next;
  };
  struct Token {
    float tot_cost, extra_cost; ForwardLink *links; Token *next;
    int32_t state = -1;  // graph state (not in the reference's Token; used only to label output)
    void DeleteForwardLinks() {
      for (ForwardLink *l = links, *m; l != nullptr; l = m) { m = l->next; delete l; }
      links = nullptr;
    }
  };
  struct TokenList {
    Token *toks = nullptr; bool must_prune_forward_links = true, must_prune_tokens = true;
  };
  const Graph &g;
  LatticeSimpleOpts config;
  int32_t num_toks = 0;
  std::unordered_map<int32_t, Token *> cur_toks, prev_toks;
  std::vector<TokenList> active_toks;
  bool decoding_finalized = false;
  std::unordered_map<Token *, float> final_costs;
  float final_relative_cost = 0.f, final_best_cost = 0.f;
  int64_t links_created = 0;

  LatticeSimpleDecoder(const Graph &graph, const LatticeSimpleOpts &o) : g(graph), config(o) {
    if (!(o.beam > 0.0 && o.lattice_beam > 0.0 && o.prune_interval > 0))  // .h:63-66
      throw std::runtime_error("beam > 0.0 && lattice_beam > 0.0 && prune_interval > 0");
  }
  ~LatticeSimpleDecoder() { ClearActiveTokens(); }
  int32_t NumFramesDecoded() const { return static_cast<int32_t>(active_toks.size()) - 1; }

  void ClearActiveTokens() {  // .cc:36-51
    for (auto &tl : active_toks)
      for (Token *tok = tl.toks; tok != nullptr;) {
        tok->DeleteForwardLinks();
        Token *nx = tok->next;
        delete tok;
        num_toks--;
        tok = nx;
      }
    active_toks.clear();
  }
  void InitDecoding() {  // .cc:17-34
    cur_toks.clear();
    prev_toks.clear();
    ClearActiveTokens();
    decoding_finalized = false;
    final_costs.clear();
    num_toks = 0;
    links_created = 0;
    if (g.start < 0) throw std::runtime_error("graph has no start state");
    active_toks.resize(1);
    Token *start_tok = new Token{0.0f, 0.0f, nullptr, nullptr};
    start_tok->state = g.start;
    active_toks[0].toks = start_tok;
    cur_toks[g.start] = start_tok;
    num_toks++;
    ProcessNonemitting();
  }
  bool Decode(const Decodable &d) {  // .cc:53-73
    InitDecoding();
    while (!d.IsLastFrame(NumFramesDecoded() - 1)) {
      if (NumFramesDecoded() % config.prune_interval == 0)
        PruneActiveTokens(config.lattice_beam * config.prune_scale);
      ProcessEmitting(d);
      PruneCurrentTokens(config.beam);
      ProcessNonemitting();
    }
    FinalizeDecoding();
    return !final_costs.empty();
  }
  // Decode() up to, and without, FinalizeDecoding (.cc:53-66): the state the reference is in
  // when its loop ends -- what a caller of the declared LatticeFasterDecoder's
  // AdvanceDecoding + GetRawLattice would see.  Test infrastructure for the KAT on
  // PruneActiveTokens' early stop (.cc:208-216, :290).
  void DecodeNoFinal(const Decodable &d) {
    InitDecoding();
    while (!d.IsLastFrame(NumFramesDecoded() - 1)) {
      if (NumFramesDecoded() % config.prune_interval == 0)
        PruneActiveTokens(config.lattice_beam * config.prune_scale);
      ProcessEmitting(d);
      PruneCurrentTokens(config.beam);
      ProcessNonemitting();
    }
  }
  Token *FindOrAddToken(int32_t state, int32_t frame, float tot_cost, bool *changed) {  // .cc:82-120
    Token *&toks = active_toks[frame].toks;
    auto it = cur_toks.find(state);
    if (it == cur_toks.end()) {
      Token *nt = new Token{tot_cost, 0.0f, nullptr, toks};
      nt->state = state;
      toks = nt;
      num_toks++;
      cur_toks[state] = nt;
      if (changed) *changed = true;
      return nt;
    }
    Token *tok = it->second;
    if (tok->tot_cost > tot_cost) { tok->tot_cost = tot_cost; if (changed) *changed = true; }
    else if (changed) *changed = false;
    return tok;
  }
  void ProcessNonemitting() {  // .cc:122-191
    int32_t frame = static_cast<int32_t>(active_toks.size()) - 2;
    std::vector<int32_t> queue;
    float best_cost = kInfF;
    for (auto &kv : cur_toks) {
      if (g.num_ieps[kv.first] != 0) queue.push_back(kv.first);
      best_cost = std::min(best_cost, kv.second->tot_cost);
    }
    float cutoff = best_cost + config.beam;
    while (!queue.empty()) {
      int32_t state = queue.back();
      queue.pop_back();
      Token *tok = cur_toks[state];
      tok->DeleteForwardLinks();
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        float graph_cost = g.weight[a], cur_cost = tok->tot_cost, tot_cost = cur_cost + graph_cost;
        if (tot_cost < cutoff) {
          bool changed;
          Token *nt = FindOrAddToken(g.nextstate[a], frame + 1, tot_cost, &changed);
          tok->links = new ForwardLink{nt, 0, g.olabel[a], graph_cost, 0, tok->links};
          links_created++;
          if (changed && g.num_ieps[g.nextstate[a]] != 0) queue.push_back(g.nextstate[a]);
        }
      }
    }
  }
  void PruneActiveTokens(float delta) {  // .cc:198-223
    int32_t cur_frame_plus_one = NumFramesDecoded();
    for (int32_t f = cur_frame_plus_one - 1; f >= 0; f--) {
      if (active_toks[f].must_prune_forward_links) {
        bool extra_costs_changed = false, links_pruned = false;
        PruneForwardLinks(f, &extra_costs_changed, &links_pruned, delta);
        if (extra_costs_changed && f > 0) active_toks[f - 1].must_prune_forward_links = true;
        if (links_pruned) active_toks[f].must_prune_tokens = true;
        active_toks[f].must_prune_forward_links = false;
      }
      if (f + 1 < cur_frame_plus_one && active_toks[f + 1].must_prune_tokens) {
        PruneTokensForFrame(f + 1);
        active_toks[f + 1].must_prune_tokens = false;
      }
    }
  }
  void PruneForwardLinks(int32_t frame, bool *extra_costs_changed, bool *links_pruned,
                         float delta) {  // .cc:228-305
    *extra_costs_changed = false;
    *links_pruned = false;
    bool changed = true;
    while (changed) {
      changed = false;
      for (Token *tok = active_toks[frame].toks; tok != nullptr; tok = tok->next) {
        ForwardLink *link, *prev_link = nullptr;
        float tok_extra_cost = kInfF;
        for (link = tok->links; link != nullptr;) {
          Token *next_tok = link->next_tok;
          float link_extra_cost =
              next_tok->extra_cost +
              ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - next_tok->tot_cost);
          if (link_extra_cost != link_extra_cost) throw std::runtime_error("NaN link_extra_cost");
          if (link_extra_cost > config.lattice_beam) {
            ForwardLink *next_link = link->next;
            if (prev_link != nullptr) prev_link->next = next_link; else tok->links = next_link;
            delete link;
            link = next_link;
            *links_pruned = true;
          } else {
            if (link_extra_cost < 0.0) link_extra_cost = 0.0;
            if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
            prev_link = link;
            link = link->next;
          }
        }
        if (std::fabs(tok_extra_cost - tok->extra_cost) > delta) changed = true;
        tok->extra_cost = tok_extra_cost;
      }
      if (changed) *extra_costs_changed = true;
    }
  }
  void PruneTokensForFrame(int32_t frame) {  // .cc:310-334
    Token *&toks = active_toks[frame].toks;
    Token *tok, *next_tok, *prev_tok = nullptr;
    for (tok = toks; tok != nullptr; tok = next_tok) {
      next_tok = tok->next;
      if (tok->extra_cost == kInfF) {
        if (prev_tok != nullptr) prev_tok->next = tok->next; else toks = tok->next;
        delete tok;
        num_toks--;
      } else {
        prev_tok = tok;
      }
    }
  }
  void PruneCurrentTokens(float beam) {  // .cc:339-362
    if (cur_toks.empty()) return;
    float best_cost = 1.0e+10;
    for (auto &kv : cur_toks) best_cost = std::min(best_cost, kv.second->tot_cost);
    std::vector<int32_t> retained;
    float cutoff = best_cost + beam;
    for (auto &kv : cur_toks) if (kv.second->tot_cost < cutoff) retained.push_back(kv.first);
    std::unordered_map<int32_t, Token *> tmp;
    for (int32_t s : retained) tmp[s] = cur_toks[s];
    tmp.swap(cur_toks);
  }
  void ProcessEmitting(const Decodable &d) {  // .cc:364-402
    int32_t frame = static_cast<int32_t>(active_toks.size()) - 1;
    active_toks.resize(active_toks.size() + 1);
    prev_toks.clear();
    cur_toks.swap(prev_toks);
    float cutoff = kInfF;
    for (auto &kv : prev_toks) {
      int32_t state = kv.first;
      Token *tok = kv.second;
      for (int32_t a = g.row_ptr[state]; a < g.row_ptr[state + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float ac_cost = -d.LogLikelihood(frame, g.ilabel[a]), graph_cost = g.weight[a],
              cur_cost = tok->tot_cost, tot_cost = cur_cost + ac_cost + graph_cost;
        if (tot_cost >= cutoff) continue;
        else if (tot_cost + config.beam < cutoff) cutoff = tot_cost + config.beam;
        Token *nt = FindOrAddToken(g.nextstate[a], frame + 1, tot_cost, nullptr);
        tok->links = new ForwardLink{nt, g.ilabel[a], g.olabel[a], graph_cost, ac_cost, tok->links};
        links_created++;
      }
    }
  }
  void FinalizeDecoding() {  // .cc:407-420
    int32_t final_frame_plus_one = NumFramesDecoded();
    PruneForwardLinksFinal();
    for (int32_t f = final_frame_plus_one - 1; f >= 0; f--) {
      bool b1, b2;
      PruneForwardLinks(f, &b1, &b2, 0.0f);
      PruneTokensForFrame(f + 1);
    }
    PruneTokensForFrame(0);
  }
  void PruneForwardLinksFinal() {  // .cc:425-520
    int32_t frame_plus_one = static_cast<int32_t>(active_toks.size()) - 1;
    ComputeFinalCosts(&final_costs, &final_relative_cost, &final_best_cost);
    decoding_finalized = true;
    cur_toks.clear();
    bool changed = true;
    float delta = 1.0e-05;
    while (changed) {
      changed = false;
      for (Token *tok = active_toks[frame_plus_one].toks; tok != nullptr; tok = tok->next) {
        ForwardLink *link, *prev_link = nullptr;
        float final_cost;
        if (final_costs.empty()) final_cost = 0.0;
        else {
          auto it = final_costs.find(tok);
          final_cost = it != final_costs.end() ? it->second : kInfF;
        }
        float tok_extra_cost = tok->tot_cost + final_cost - final_best_cost;
        for (link = tok->links; link != nullptr;) {
          Token *next_tok = link->next_tok;
          float link_extra_cost =
              next_tok->extra_cost +
              ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - next_tok->tot_cost);
          if (link_extra_cost > config.lattice_beam) {
            ForwardLink *next_link = link->next;
            if (prev_link != nullptr) prev_link->next = next_link; else tok->links = next_link;
            delete link;
            link = next_link;
          } else {
            if (link_extra_cost < 0.0) link_extra_cost = 0.0;
            if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
            prev_link = link;
            link = link->next;
          }
        }
        if (tok_extra_cost > config.lattice_beam) tok_extra_cost = kInfF;
        if (!ApproxEqual(tok->extra_cost, tok_extra_cost, delta)) changed = true;
        tok->extra_cost = tok_extra_cost;
      }
    }
  }
  void ComputeFinalCosts(std::unordered_map<Token *, float> *fc, float *final_relative,
                         float *final_best) const {  // .cc:522-560
    if (decoding_finalized) throw std::runtime_error("ComputeFinalCosts after FinalizeDecoding");
    if (fc) fc->clear();
    float best_cost = kInfF, best_cost_with_final = kInfF;
    for (auto &kv : cur_toks) {
      float final_cost = g.final_cost[kv.first];
      float cost = kv.second->tot_cost, cost_with_final = cost + final_cost;
      best_cost = std::min(cost, best_cost);
      best_cost_with_final = std::min(cost_with_final, best_cost_with_final);
      if (fc && final_cost != kInfF) (*fc)[kv.second] = final_cost;
    }
    if (final_relative) {
      if (best_cost == kInfF && best_cost_with_final == kInfF) *final_relative = kInfF;
      else *final_relative = best_cost_with_final - best_cost;
    }
    if (final_best) *final_best = best_cost_with_final != kInfF ? best_cost_with_final : best_cost;
  }
  float FinalRelativeCost() const {  // .cc:562-572
    if (!decoding_finalized) {
      float r;
      ComputeFinalCosts(nullptr, &r, nullptr);
      return r;
    }
    return final_relative_cost;
  }
  bool GetRawLattice(Lattice *out, bool use_final_probs) const {  // .cc:584-657
    if (decoding_finalized && !use_final_probs)
      throw std::runtime_error(
          "You cannot call FinalizeDecoding() and then call GetRawLattice() with "
          "use_final_probs == false");
    std::unordered_map<Token *, float> final_costs_local;
    const auto &fcs = decoding_finalized ? final_costs : final_costs_local;
    if (!decoding_finalized && use_final_probs) ComputeFinalCosts(&final_costs_local, nullptr, nullptr);
    *out = Lattice();
    int32_t num_frames = NumFramesDecoded();
    if (!(num_frames > 0)) throw std::runtime_error("num_frames > 0");
    std::unordered_map<Token *, int32_t> tok_map(num_toks / 2 + 3);
    for (int32_t f = 0; f <= num_frames; f++) {
      if (active_toks[f].toks == nullptr) { *out = Lattice(); return false; }
      for (Token *tok = active_toks[f].toks; tok != nullptr; tok = tok->next) {
        tok_map[tok] = out->num_states++;
        out->state_frame.push_back(f);
        out->state_graph.push_back(tok->state);
      }
      if (f == 0 && out->num_states > 0) out->start = out->num_states - 1;
    }
    int32_t cur_state = 0;
    for (int32_t f = 0; f <= num_frames; f++) {
      for (Token *tok = active_toks[f].toks; tok != nullptr; tok = tok->next, cur_state++) {
        for (ForwardLink *l = tok->links; l != nullptr; l = l->next) {
          out->src.push_back(cur_state);
          out->dst.push_back(tok_map.find(l->next_tok)->second);
          out->ilabel.push_back(l->ilabel);
          out->olabel.push_back(l->olabel);
          out->graph_cost.push_back(l->graph_cost);
          out->ac_cost.push_back(l->acoustic_cost);
        }
        if (f == num_frames) {
          if (use_final_probs && !fcs.empty()) {
            auto it = fcs.find(tok);
            if (it != fcs.end()) {
              out->final_state.push_back(cur_state);
              out->final_graph.push_back(it->second);
              out->final_ac.push_back(0.f);
            }
          } else {
            out->final_state.push_back(cur_state);
            out->final_graph.push_back(0.f);
            out->final_ac.push_back(0.f);
          }
        }
      }
    }
    return cur_state != 0;
  }
};


// ---------------------------------------------------------------------------
// Canonical (order-free) lattice decoder: what the GPU implements.
//
// LatticeSimpleDecoder's semantics (lattice-simple-decoder.cc) without its order
// dependence: per frame the emitting cutoff is the value the reference's running
// cutoff ends at (min tot_cost + beam, :386-390), every arc with tot_cost below
// it becomes a forward link (:393-398), a token's cost is the min over its
// incoming links, the epsilon closure links every arc with cur + graph < cutoff
// from each token's final cost (:164-180), and all pruning happens once, exactly
// (delta = 0), in FinalizeDecoding -- intermediate PruneActiveTokens calls only
// ever remove links the final pass removes too (:198-305).  Costs are float with
// the reference's association (cur + ac + graph, :383-385; cur + graph, :170-172).
//
// With max_active / min_active set this is the LatticeFasterDecoder the
// reference only declares (lattice-faster-decoder.h:23-134): tokens are expanded
// iff tot_cost < GetCutoff()'s weight cutoff and the emitting / epsilon cutoff is
// min tot_cost + adaptive_beam (faster-decoder.cc:244-336 applied to float costs).
// ---------------------------------------------------------------------------
struct CanonLatticeOpts {
  float beam = 16.f, lattice_beam = 10.f;
  int32_t max_active = std::numeric_limits<int32_t>::max(), min_active = 0;
  float beam_delta = 0.5f;
};

struct CanonLatticeDecoder {
  struct Tok { int32_t state; float tot, extra; };
  struct Link { int32_t src, dst, dst_frame, arc; float graph, ac; bool dead; };
  const Graph &g;
  CanonLatticeOpts o;
  std::vector<std::vector<Tok>> toks;    // [frame_plus_one], sorted by state
  std::vector<std::vector<Link>> links;  // [frame_plus_one] links leaving that frame's tokens
  bool finalized = false;
  float final_relative_cost = 0.f, final_best_cost = 0.f;
  bool any_final = false;
  std::vector<float> tmp_array;
  int64_t links_created = 0;

  CanonLatticeDecoder(const Graph &graph, const CanonLatticeOpts &opts) : g(graph), o(opts) {
    if (!(o.beam > 0.0 && o.lattice_beam > 0.0)) throw std::runtime_error("beam > 0.0 && lattice_beam > 0.0");
    if (!(o.max_active > 1 && o.min_active >= 0 && o.min_active < o.max_active))
      throw std::runtime_error("max_active > 1 && 0 <= min_active < max_active");
  }
  int32_t NumFramesDecoded() const { return static_cast<int32_t>(toks.size()) - 1; }

  // epsilon closure of `cur` (state -> cost) with a fixed cutoff; then freeze the
  // frame's tokens and emit its epsilon links from the final costs
  void CloseAndFreeze(std::unordered_map<int32_t, float> &cur, float cutoff) {
    std::vector<int32_t> work;
    for (auto &kv : cur) if (g.num_ieps[kv.first] != 0) work.push_back(kv.first);
    while (!work.empty()) {
      int32_t s = work.back();
      work.pop_back();
      float c = cur[s];
      for (int32_t a = g.row_ptr[s]; a < g.row_ptr[s + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        float tot = c + g.weight[a];
        if (!(tot < cutoff)) continue;
        auto it = cur.find(g.nextstate[a]);
        if (it == cur.end()) { cur[g.nextstate[a]] = tot; if (g.num_ieps[g.nextstate[a]]) work.push_back(g.nextstate[a]); }
        else if (it->second > tot) { it->second = tot; if (g.num_ieps[g.nextstate[a]]) work.push_back(g.nextstate[a]); }
      }
    }
    std::vector<Tok> fr;
    for (auto &kv : cur) fr.push_back(Tok{kv.first, kv.second, 0.0f});
    std::sort(fr.begin(), fr.end(), [](const Tok &a, const Tok &b) { return a.state < b.state; });
    std::vector<Link> el;
    auto idx_of = [&](int32_t state) {
      auto it = std::lower_bound(fr.begin(), fr.end(), state, [](const Tok &t, int32_t s) { return t.state < s; });
      return static_cast<int32_t>(it - fr.begin());
    };
    const int32_t f = static_cast<int32_t>(toks.size());
    for (int32_t i = 0; i < static_cast<int32_t>(fr.size()); ++i) {
      int32_t s = fr[i].state;
      if (g.num_ieps[s] == 0) continue;
      for (int32_t a = g.row_ptr[s]; a < g.row_ptr[s + 1]; ++a) {
        if (g.ilabel[a] != 0) continue;
        float tot = fr[i].tot + g.weight[a];
        if (!(tot < cutoff)) continue;
        el.push_back(Link{i, idx_of(g.nextstate[a]), f, a, g.weight[a], 0.0f, false});
      }
    }
    links_created += static_cast<int64_t>(el.size());
    toks.push_back(std::move(fr));
    links.push_back(std::move(el));
  }

  void InitDecoding() {
    toks.clear();
    links.clear();
    finalized = false;
    links_created = 0;
    if (g.start < 0) throw std::runtime_error("graph has no start state");
    std::unordered_map<int32_t, float> cur;
    cur[g.start] = 0.0f;
    CloseAndFreeze(cur, 0.0f + o.beam);  // lattice-simple-decoder.cc:135-154 with one token
  }

  // faster-decoder.cc:244-336 on this frame's float costs
  float GetCutoff(const std::vector<Tok> &fr, float *adaptive_beam) {
    float best = kInfF;
    for (auto &t : fr) best = std::min(best, t.tot);
    if (o.max_active == std::numeric_limits<int32_t>::max() && o.min_active == 0) {
      *adaptive_beam = o.beam;
      return kInfF;  // LatticeSimpleDecoder expands every token (:376-401)
    }
    tmp_array.clear();
    for (auto &t : fr) tmp_array.push_back(t.tot);
    double beam_cutoff = static_cast<double>(best) + o.beam, mn = kInfD, mx = kInfD;
    if (tmp_array.size() > static_cast<size_t>(o.max_active)) {
      std::nth_element(tmp_array.begin(), tmp_array.begin() + o.max_active, tmp_array.end());
      mx = tmp_array[o.max_active];
    }
    if (mx < beam_cutoff) {
      *adaptive_beam = mx - best + o.beam_delta;
      return static_cast<float>(mx);
    }
    if (tmp_array.size() > static_cast<size_t>(o.min_active)) {
      if (o.min_active == 0) mn = best;
      else {
        std::nth_element(tmp_array.begin(), tmp_array.begin() + o.min_active,
                         tmp_array.size() > static_cast<size_t>(o.max_active)
                             ? tmp_array.begin() + o.max_active : tmp_array.end());
        mn = tmp_array[o.min_active];
      }
    }
    if (mn > beam_cutoff) {
      *adaptive_beam = mn - best + o.beam_delta;
      return static_cast<float>(mn);
    }
    *adaptive_beam = o.beam;
    return static_cast<float>(beam_cutoff);
  }

  void ProcessFrame(const Decodable &d) {
    const int32_t frame = NumFramesDecoded();  // index into the decodable
    const std::vector<Tok> &prev = toks.back();
    float adaptive_beam;
    const float weight_cutoff = GetCutoff(prev, &adaptive_beam);
    float best_new = kInfF;
    for (auto &t : prev) {
      if (!(t.tot < weight_cutoff)) continue;
      for (int32_t a = g.row_ptr[t.state]; a < g.row_ptr[t.state + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float ac = -d.LogLikelihood(frame, g.ilabel[a]), tot = t.tot + ac + g.weight[a];
        best_new = std::min(best_new, tot);
      }
    }
    const float cutoff = best_new + adaptive_beam;
    std::unordered_map<int32_t, float> cur;
    struct Pending { int32_t src, state, arc; float ac; };
    std::vector<Pending> pend;
    for (int32_t i = 0; i < static_cast<int32_t>(prev.size()); ++i) {
      const Tok &t = prev[i];
      if (!(t.tot < weight_cutoff)) continue;
      for (int32_t a = g.row_ptr[t.state]; a < g.row_ptr[t.state + 1]; ++a) {
        if (g.ilabel[a] == 0) continue;
        float ac = -d.LogLikelihood(frame, g.ilabel[a]), tot = t.tot + ac + g.weight[a];
        if (!(tot < cutoff)) continue;  // :386
        auto it = cur.find(g.nextstate[a]);
        if (it == cur.end()) cur[g.nextstate[a]] = tot;
        else if (it->second > tot) it->second = tot;
        pend.push_back(Pending{i, g.nextstate[a], a, ac});
      }
    }
    CloseAndFreeze(cur, cutoff);
    const std::vector<Tok> &fr = toks.back();
    std::vector<Link> &el = links[links.size() - 2];  // links leaving the previous frame
    const int32_t f1 = static_cast<int32_t>(toks.size()) - 1;
    for (auto &pd : pend) {
      auto it = std::lower_bound(fr.begin(), fr.end(), pd.state, [](const Tok &t, int32_t s) { return t.state < s; });
      el.push_back(Link{pd.src, static_cast<int32_t>(it - fr.begin()), f1, pd.arc, g.weight[pd.arc], pd.ac, false});
    }
    links_created += static_cast<int64_t>(pend.size());
  }

  bool Decode(const Decodable &d) {  // lattice-simple-decoder.cc:53-73
    InitDecoding();
    while (!d.IsLastFrame(NumFramesDecoded() - 1)) ProcessFrame(d);
    FinalizeDecoding();
    return any_final;  // the reference returns !final_costs_.empty() (:72)
  }

  // PruneForwardLinks with delta = 0 for one frame (:228-305); `same` selects
  // whether links into this very frame (epsilon) need the fixpoint iteration.
  void PruneFrame(int32_t f, bool last) {
    std::vector<Tok> &fr = toks[f];
    std::vector<float> base(fr.size(), kInfF);
    if (last) {  // PruneForwardLinksFinal (:425-520)
      float best = kInfF, best_wf = kInfF;
      any_final = false;
      for (auto &t : fr) {
        float fc = g.final_cost[t.state];
        best = std::min(best, t.tot);
        best_wf = std::min(best_wf, t.tot + fc);
        if (fc != kInfF) any_final = true;
      }
      final_relative_cost = (best == kInfF && best_wf == kInfF) ? kInfF : best_wf - best;
      final_best_cost = best_wf != kInfF ? best_wf : best;
      for (size_t i = 0; i < fr.size(); ++i) {
        float fc = any_final ? g.final_cost[fr[i].state] : 0.0f;
        base[i] = fr[i].tot + fc - final_best_cost;
      }
    }
    bool changed = true;
    while (changed) {
      changed = false;
      std::vector<float> ne = base;
      for (Link &l : links[f]) {
        if (l.dead) continue;
        const Tok &src = fr[l.src];
        const Tok &nx = toks[l.dst_frame][l.dst];
        float le = nx.extra + ((src.tot + l.ac + l.graph) - nx.tot);
        if (le > o.lattice_beam) { l.dead = true; continue; }
        if (le < 0.0f) le = 0.0f;
        if (le < ne[l.src]) ne[l.src] = le;
      }
      for (size_t i = 0; i < fr.size(); ++i) {
        float v = ne[i];
        if (last && v > o.lattice_beam) v = kInfF;
        if (v != fr[i].extra) { changed = true; fr[i].extra = v; }
      }
    }
  }

  void FinalizeDecoding() {  // :407-420
    const int32_t last = NumFramesDecoded();
    PruneFrame(last, true);
    for (int32_t f = last - 1; f >= 0; --f) PruneFrame(f, false);
    finalized = true;
  }

  bool GetRawLattice(Lattice *out) const {  // :584-657, tokens labelled (frame, state)
    *out = Lattice();
    const int32_t nf = NumFramesDecoded();
    std::vector<std::vector<int32_t>> id(toks.size());
    for (int32_t f = 0; f <= nf; ++f) {
      id[f].assign(toks[f].size(), -1);
      bool any = false;
      for (size_t i = 0; i < toks[f].size(); ++i)
        if (toks[f][i].extra != kInfF) {
          id[f][i] = out->num_states++;
          out->state_frame.push_back(f);
          out->state_graph.push_back(toks[f][i].state);
          any = true;
          if (f == 0 && toks[f][i].state == g.start) out->start = id[f][i];
        }
      if (!any) { *out = Lattice(); return false; }
    }
    for (int32_t f = 0; f <= nf; ++f)
      for (const Link &l : links[f]) {
        if (l.dead || id[f][l.src] < 0 || id[l.dst_frame][l.dst] < 0) continue;
        out->src.push_back(id[f][l.src]);
        out->dst.push_back(id[l.dst_frame][l.dst]);
        out->ilabel.push_back(g.ilabel[l.arc]);
        out->olabel.push_back(g.olabel[l.arc]);
        out->graph_cost.push_back(l.graph);
        out->ac_cost.push_back(l.ac);
      }
    for (size_t i = 0; i < toks[nf].size(); ++i) {
      if (id[nf][i] < 0) continue;
      float fc = g.final_cost[toks[nf][i].state];
      if (any_final) {
        if (fc != kInfF) { out->final_state.push_back(id[nf][i]); out->final_graph.push_back(fc); out->final_ac.push_back(0.f); }
      } else {
        out->final_state.push_back(id[nf][i]); out->final_graph.push_back(0.f); out->final_ac.push_back(0.f);
      }
    }
    return out->num_states != 0;
  }
};

// Tropical shortest path over LatticeWeight (graph, ac): compare value1+value2,
// then value1 (kaldifst lattice-weight.h Compare; OpenFst ShortestPath is not in
// /root/reference, call site lattice-simple-decoder.cc:574-580).  Label
// correcting, so negative arc costs are fine; the lattice has no cycles.
bool LatticeShortestPath(const Lattice &lat, Path *out) {
  *out = Path();
  if (lat.num_states == 0 || lat.start < 0) return false;
  int32_t n = lat.num_states;
  std::vector<int32_t> first(n + 1, 0);
  for (int32_t s : lat.src) first[s + 1]++;
  for (int32_t i = 0; i < n; i++) first[i + 1] += first[i];
  std::vector<int32_t> order(lat.src.size()), fill(first.begin(), first.end() - 1);
  for (size_t a = 0; a < lat.src.size(); a++) order[fill[lat.src[a]]++] = static_cast<int32_t>(a);
  std::vector<float> d1(n, kInfF), d2(n, kInfF);
  std::vector<int32_t> back(n, -1);
  auto better = [](float a1, float a2, float b1, float b2) {
    float sa = a1 + a2, sb = b1 + b2;
    if (sa < sb) return true;
    if (sa > sb) return false;
    return a1 < b1;
  };
  std::vector<int32_t> queue{lat.start};
  std::vector<char> inq(n, 0);
  d1[lat.start] = d2[lat.start] = 0.f;
  inq[lat.start] = 1;
  for (size_t qi = 0; qi < queue.size(); qi++) {
    int32_t s = queue[qi];
    inq[s] = 0;
    for (int32_t k = first[s]; k < first[s + 1]; k++) {
      int32_t a = order[k], t = lat.dst[a];
      float n1 = d1[s] + lat.graph_cost[a], n2 = d2[s] + lat.ac_cost[a];
      if (better(n1, n2, d1[t], d2[t])) {
        d1[t] = n1; d2[t] = n2; back[t] = a;
        if (!inq[t]) { inq[t] = 1; queue.push_back(t); }
      }
    }
  }
  int32_t best = -1;
  float b1 = kInfF, b2 = kInfF;
  size_t bf = 0;
  for (size_t i = 0; i < lat.final_state.size(); i++) {
    int32_t s = lat.final_state[i];
    if (d1[s] == kInfF) continue;
    float n1 = d1[s] + lat.final_graph[i], n2 = d2[s] + lat.final_ac[i];
    if (best < 0 || better(n1, n2, b1, b2)) { best = s; b1 = n1; b2 = n2; bf = i; }
  }
  if (best < 0) return false;
  for (int32_t s = best; s != lat.start;) {
    int32_t a = back[s];
    out->ilabel.push_back(lat.ilabel[a]);
    out->olabel.push_back(lat.olabel[a]);
    out->graph_cost.push_back(lat.graph_cost[a]);
    out->ac_cost.push_back(lat.ac_cost[a]);
    out->state.push_back(s);
    s = lat.src[a];
  }
  std::reverse(out->ilabel.begin(), out->ilabel.end());
  std::reverse(out->olabel.begin(), out->olabel.end());
  std::reverse(out->graph_cost.begin(), out->graph_cost.end());
  std::reverse(out->ac_cost.begin(), out->ac_cost.end());
  std::reverse(out->state.begin(), out->state.end());
  out->final_graph = lat.final_graph[bf];
  out->final_ac = lat.final_ac[bf];
  out->ok = true;
  return true;
}

thread_local std::string g_err;

template <class F> int Guard(F f) {
  try { f(); return 0; }
  catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int CopyPath(const Path &p, int32_t cap, int32_t *n, int32_t *il, int32_t *ol, int32_t *st,
             float *gc, float *ac, float *final_w) {
  *n = static_cast<int32_t>(p.ilabel.size());
  final_w[0] = p.final_graph;
  final_w[1] = p.final_ac;
  if (!p.ok) return 0;
  if (*n > cap) { g_err = "path buffer too small"; return -1; }
  if (*n) {
    std::memcpy(il, p.ilabel.data(), *n * 4);
    std::memcpy(ol, p.olabel.data(), *n * 4);
    std::memcpy(st, p.state.data(), *n * 4);
    std::memcpy(gc, p.graph_cost.data(), *n * 4);
    std::memcpy(ac, p.ac_cost.data(), *n * 4);
  }
  return 1;
}

}  // namespace

// ---------------------------------------------------------------------------
// C API for ctypes (oracle/oracle.py)
// ---------------------------------------------------------------------------
extern "C" {

const char *kdo_last_error() { return g_err.c_str(); }

void *kdo_graph_create(int32_t num_states, int32_t start, const int32_t *row_ptr,
                       const int32_t *ilabel, const int32_t *olabel, const float *weight,
                       const int32_t *nextstate, const float *final_cost) {
  Graph *g = new Graph;
  g->num_states = num_states;
  g->start = start;
  int32_t na = row_ptr[num_states];
  g->row_ptr.assign(row_ptr, row_ptr + num_states + 1);
  g->ilabel.assign(ilabel, ilabel + na);
  g->olabel.assign(olabel, olabel + na);
  g->weight.assign(weight, weight + na);
  g->nextstate.assign(nextstate, nextstate + na);
  g->final_cost.assign(final_cost, final_cost + num_states);
  g->num_ieps.assign(num_states, 0);
  g->canon_idx.resize(na);
  for (int32_t s = 0; s < num_states; s++) {
    int32_t b = row_ptr[s], e = row_ptr[s + 1], ne = 0;
    for (int32_t a = b; a < e; a++) if (ilabel[a] == 0) ne++;
    g->num_ieps[s] = ne;
    int32_t pe = b, pn = e - ne;
    for (int32_t a = b; a < e; a++) g->canon_idx[a] = ilabel[a] != 0 ? pe++ : pn++;
  }
  return g;
}
void kdo_graph_free(void *g) { delete static_cast<Graph *>(g); }

// -- FasterDecoder ------------------------------------------------------------
void *kdo_faster_create(void *graph, float beam, int32_t max_active, int32_t min_active,
                        float beam_delta, float hash_ratio, int32_t policy) {
  FasterDecoder *d = nullptr;
  FasterOpts o;
  o.beam = beam; o.max_active = max_active; o.min_active = min_active;
  o.beam_delta = beam_delta; o.hash_ratio = hash_ratio;
  if (Guard([&] { d = new FasterDecoder(*static_cast<Graph *>(graph), o, policy); })) return nullptr;
  return d;
}
void kdo_faster_free(void *d) { delete static_cast<FasterDecoder *>(d); }
int kdo_faster_init(void *d) { return Guard([&] { static_cast<FasterDecoder *>(d)->InitDecoding(); }); }
int kdo_faster_advance(void *d, const float *p, int32_t rows, int32_t cols, int32_t offset,
                       int32_t max_num_frames) {
  Decodable dec{p, rows, cols, offset};
  return Guard([&] { static_cast<FasterDecoder *>(d)->AdvanceDecoding(dec, max_num_frames); });
}
int32_t kdo_faster_num_frames_decoded(void *d) { return static_cast<FasterDecoder *>(d)->num_frames_decoded; }
int32_t kdo_faster_reached_final(void *d) { return static_cast<FasterDecoder *>(d)->ReachedFinal(); }
int32_t kdo_faster_num_tokens(void *d) {
  int32_t n = 0;
  static_cast<FasterDecoder *>(d)->ForEachToken([&](int32_t, int32_t) { n++; });
  return n;
}
// current frontier as (state, cost) sorted by state
void kdo_faster_tokens(void *dv, int32_t *states, double *costs) {
  auto *d = static_cast<FasterDecoder *>(dv);
  std::vector<std::pair<int32_t, double>> v;
  d->ForEachToken([&](int32_t s, int32_t t) { v.push_back({s, d->arena[t].cost}); });
  std::sort(v.begin(), v.end());
  for (size_t i = 0; i < v.size(); i++) { states[i] = v[i].first; costs[i] = v[i].second; }
}
int kdo_faster_best_path(void *d, int32_t use_final_probs, int32_t cap, int32_t *n, int32_t *il,
                         int32_t *ol, int32_t *st, float *gc, float *ac, float *final_w) {
  Path p = static_cast<FasterDecoder *>(d)->GetBestPath(use_final_probs != 0);
  return CopyPath(p, cap, n, il, ol, st, gc, ac, final_w);
}
void kdo_faster_stats(void *d, int64_t *out9) {
  const Stats &s = static_cast<FasterDecoder *>(d)->stats;
  int64_t v[9] = {s.frames, s.tokens_in, s.tokens_expanded, s.arcs_scanned, s.arcs_emit_kept,
                  s.tokens_out, s.eps_arcs_scanned, s.eps_states_expanded, s.max_tokens};
  std::memcpy(out9, v, sizeof(v));
}

// -- SimpleDecoder --------------------------------------------------------------
void *kdo_simple_create(void *graph, float beam) {
  return new SimpleDecoder(*static_cast<Graph *>(graph), beam);
}
void kdo_simple_free(void *d) { delete static_cast<SimpleDecoder *>(d); }
int kdo_simple_init(void *d) { return Guard([&] { static_cast<SimpleDecoder *>(d)->InitDecoding(); }); }
int kdo_simple_advance(void *d, const float *p, int32_t rows, int32_t cols, int32_t offset,
                       int32_t max_num_frames) {
  Decodable dec{p, rows, cols, offset};
  return Guard([&] { static_cast<SimpleDecoder *>(d)->AdvanceDecoding(dec, max_num_frames); });
}
int32_t kdo_simple_num_frames_decoded(void *d) { return static_cast<SimpleDecoder *>(d)->num_frames_decoded; }
int32_t kdo_simple_reached_final(void *d) { return static_cast<SimpleDecoder *>(d)->ReachedFinal(); }
int32_t kdo_simple_num_tokens(void *d) { return static_cast<int32_t>(static_cast<SimpleDecoder *>(d)->cur_toks.size()); }
float kdo_simple_final_relative_cost(void *d) { return static_cast<SimpleDecoder *>(d)->FinalRelativeCost(); }
int kdo_simple_best_path(void *d, int32_t use_final_probs, int32_t cap, int32_t *n, int32_t *il,
                         int32_t *ol, int32_t *st, float *gc, float *ac, float *final_w) {
  Path p = static_cast<SimpleDecoder *>(d)->GetBestPath(use_final_probs != 0);
  return CopyPath(p, cap, n, il, ol, st, gc, ac, final_w);
}


// -- canonical SimpleDecoder (policy 1) ---------------------------------------------
void *kdo_csimple_create(void *graph, float beam) {
  return new CanonSimpleDecoder(*static_cast<Graph *>(graph), beam);
}
void kdo_csimple_free(void *d) { delete static_cast<CanonSimpleDecoder *>(d); }
int kdo_csimple_init(void *d) { return Guard([&] { static_cast<CanonSimpleDecoder *>(d)->InitDecoding(); }); }
int kdo_csimple_advance(void *d, const float *p, int32_t rows, int32_t cols, int32_t offset,
                        int32_t max_num_frames) {
  Decodable dec{p, rows, cols, offset};
  return Guard([&] { static_cast<CanonSimpleDecoder *>(d)->AdvanceDecoding(dec, max_num_frames); });
}
int32_t kdo_csimple_num_frames_decoded(void *d) { return static_cast<CanonSimpleDecoder *>(d)->num_frames_decoded; }
int32_t kdo_csimple_reached_final(void *d) { return static_cast<CanonSimpleDecoder *>(d)->ReachedFinal(); }
int32_t kdo_csimple_num_tokens(void *d) { return static_cast<int32_t>(static_cast<CanonSimpleDecoder *>(d)->cur.size()); }
float kdo_csimple_final_relative_cost(void *d) { return static_cast<CanonSimpleDecoder *>(d)->FinalRelativeCost(); }
void kdo_csimple_tokens(void *dv, int32_t *states, double *costs) {
  auto *d = static_cast<CanonSimpleDecoder *>(dv);
  size_t i = 0;
  for (auto &kv : d->Sorted()) { states[i] = kv.first; costs[i] = d->arena[kv.second].cost; ++i; }
}
int kdo_csimple_best_path(void *d, int32_t use_final_probs, int32_t cap, int32_t *n, int32_t *il,
                          int32_t *ol, int32_t *st, float *gc, float *ac, float *final_w) {
  Path p = static_cast<CanonSimpleDecoder *>(d)->GetBestPath(use_final_probs != 0);
  return CopyPath(p, cap, n, il, ol, st, gc, ac, final_w);
}

// -- LatticeSimpleDecoder ---------------------------------------------------------
struct LatHandle {
  LatticeSimpleDecoder dec;
  Lattice lat;
  LatHandle(const Graph &g, const LatticeSimpleOpts &o) : dec(g, o) {}
};
void *kdo_latsimple_create(void *graph, float beam, float lattice_beam, int32_t prune_interval,
                           float prune_scale) {
  LatticeSimpleOpts o;
  o.beam = beam; o.lattice_beam = lattice_beam; o.prune_interval = prune_interval;
  o.prune_scale = prune_scale;
  LatHandle *h = nullptr;
  if (Guard([&] { h = new LatHandle(*static_cast<Graph *>(graph), o); })) return nullptr;
  return h;
}
void kdo_latsimple_free(void *h) { delete static_cast<LatHandle *>(h); }
int kdo_latsimple_init(void *h) { return Guard([&] { static_cast<LatHandle *>(h)->dec.InitDecoding(); }); }
// returns 1/0 for Decode()'s bool, -1 on error
int kdo_latsimple_decode(void *h, const float *p, int32_t rows, int32_t cols, int32_t offset) {
  Decodable dec{p, rows, cols, offset};
  bool r = false;
  if (Guard([&] { r = static_cast<LatHandle *>(h)->dec.Decode(dec); })) return -1;
  return r ? 1 : 0;
}
int kdo_latsimple_decode_nofinal(void *h, const float *p, int32_t rows, int32_t cols, int32_t offset) {
  Decodable dec{p, rows, cols, offset};
  return Guard([&] { static_cast<LatHandle *>(h)->dec.DecodeNoFinal(dec); });
}
int kdo_latsimple_finalize(void *h) { return Guard([&] { static_cast<LatHandle *>(h)->dec.FinalizeDecoding(); }); }
int32_t kdo_latsimple_num_frames_decoded(void *h) { return static_cast<LatHandle *>(h)->dec.NumFramesDecoded(); }
float kdo_latsimple_final_relative_cost(void *h) { return static_cast<LatHandle *>(h)->dec.FinalRelativeCost(); }
int64_t kdo_latsimple_links_created(void *h) { return static_cast<LatHandle *>(h)->dec.links_created; }
// Builds the raw lattice inside the handle; sizes out: [num_states, num_arcs, num_final, start]
int kdo_latsimple_raw_lattice(void *hv, int32_t use_final_probs, int32_t *sizes4) {
  auto *h = static_cast<LatHandle *>(hv);
  bool ok = false;
  if (Guard([&] { ok = h->dec.GetRawLattice(&h->lat, use_final_probs != 0); })) return -1;
  sizes4[0] = h->lat.num_states;
  sizes4[1] = static_cast<int32_t>(h->lat.src.size());
  sizes4[2] = static_cast<int32_t>(h->lat.final_state.size());
  sizes4[3] = h->lat.start;
  return ok ? 1 : 0;
}
void kdo_latsimple_copy_lattice(void *hv, int32_t *src, int32_t *dst, int32_t *il, int32_t *ol,
                                float *gc, float *ac, int32_t *fs, float *fg, int32_t *state_frame) {
  const Lattice &l = static_cast<LatHandle *>(hv)->lat;
  size_t na = l.src.size(), nf = l.final_state.size();
  if (na) {
    std::memcpy(src, l.src.data(), na * 4); std::memcpy(dst, l.dst.data(), na * 4);
    std::memcpy(il, l.ilabel.data(), na * 4); std::memcpy(ol, l.olabel.data(), na * 4);
    std::memcpy(gc, l.graph_cost.data(), na * 4); std::memcpy(ac, l.ac_cost.data(), na * 4);
  }
  if (nf) { std::memcpy(fs, l.final_state.data(), nf * 4); std::memcpy(fg, l.final_graph.data(), nf * 4); }
  if (l.num_states) std::memcpy(state_frame, l.state_frame.data(), l.num_states * 4);
}
// GetBestPath = GetRawLattice + ShortestPath (lattice-simple-decoder.cc:574-580)
int kdo_latsimple_best_path(void *hv, int32_t use_final_probs, int32_t cap, int32_t *n, int32_t *il,
                            int32_t *ol, int32_t *st, float *gc, float *ac, float *final_w) {
  auto *h = static_cast<LatHandle *>(hv);
  Path p;
  if (Guard([&] {
        Lattice lat;
        h->dec.GetRawLattice(&lat, use_final_probs != 0);
        LatticeShortestPath(lat, &p);
      }))
    return -1;
  return CopyPath(p, cap, n, il, ol, st, gc, ac, final_w);
}


// -- canonical lattice decoder (policy 1) -----------------------------------------
struct CanonHandle {
  CanonLatticeDecoder dec;
  Lattice lat;
  CanonHandle(const Graph &g, const CanonLatticeOpts &o) : dec(g, o) {}
};
void *kdo_canonlat_create(void *graph, float beam, float lattice_beam, int32_t max_active,
                          int32_t min_active, float beam_delta) {
  CanonLatticeOpts o;
  o.beam = beam; o.lattice_beam = lattice_beam; o.max_active = max_active;
  o.min_active = min_active; o.beam_delta = beam_delta;
  CanonHandle *h = nullptr;
  if (Guard([&] { h = new CanonHandle(*static_cast<Graph *>(graph), o); })) return nullptr;
  return h;
}
void kdo_canonlat_free(void *h) { delete static_cast<CanonHandle *>(h); }
int kdo_canonlat_decode(void *h, const float *p, int32_t rows, int32_t cols, int32_t offset) {
  Decodable dec{p, rows, cols, offset};
  bool r = false;
  if (Guard([&] { r = static_cast<CanonHandle *>(h)->dec.Decode(dec); })) return -1;
  return r ? 1 : 0;
}
int32_t kdo_canonlat_num_frames_decoded(void *h) { return static_cast<CanonHandle *>(h)->dec.NumFramesDecoded(); }
float kdo_canonlat_final_relative_cost(void *h) { return static_cast<CanonHandle *>(h)->dec.final_relative_cost; }
int64_t kdo_canonlat_links_created(void *h) { return static_cast<CanonHandle *>(h)->dec.links_created; }
int32_t kdo_canonlat_frame_tokens(void *hv, int32_t f, int32_t cap, int32_t *state, float *tot, float *extra) {
  auto &fr = static_cast<CanonHandle *>(hv)->dec.toks[f];
  int32_t n = static_cast<int32_t>(fr.size());
  for (int32_t i = 0; i < n && i < cap; ++i) { state[i] = fr[i].state; tot[i] = fr[i].tot; extra[i] = fr[i].extra; }
  return n;
}
// Every forward link the decoder created, pruned or not, one row of 8 doubles each:
// (src frame, src graph state, dst frame, dst graph state, ilabel, olabel, graph, ac).
// Returns the number of links; writes at most `cap` rows.
int64_t kdo_canonlat_created_links(void *hv, int64_t cap, double *rows) {
  auto &d = static_cast<CanonHandle *>(hv)->dec;
  int64_t n = 0;
  for (size_t f = 0; f < d.links.size(); ++f)
    for (const auto &l : d.links[f]) {
      if (n < cap) {
        double *r = rows + 8 * n;
        r[0] = static_cast<double>(f);
        r[1] = d.toks[f][l.src].state;
        r[2] = l.dst_frame;
        r[3] = d.toks[l.dst_frame][l.dst].state;
        r[4] = d.g.ilabel[l.arc];
        r[5] = d.g.olabel[l.arc];
        r[6] = l.graph;
        r[7] = l.ac;
      }
      ++n;
    }
  return n;
}
int kdo_canonlat_raw_lattice(void *hv, int32_t *sizes4) {
  auto *h = static_cast<CanonHandle *>(hv);
  bool ok = false;
  if (Guard([&] { ok = h->dec.GetRawLattice(&h->lat); })) return -1;
  sizes4[0] = h->lat.num_states;
  sizes4[1] = static_cast<int32_t>(h->lat.src.size());
  sizes4[2] = static_cast<int32_t>(h->lat.final_state.size());
  sizes4[3] = h->lat.start;
  return ok ? 1 : 0;
}
static void CopyLattice(const Lattice &l, int32_t *src, int32_t *dst, int32_t *il, int32_t *ol,
                        float *gc, float *ac, int32_t *fs, float *fg, int32_t *state_frame,
                        int32_t *state_graph) {
  size_t na = l.src.size(), nf = l.final_state.size();
  if (na) {
    std::memcpy(src, l.src.data(), na * 4); std::memcpy(dst, l.dst.data(), na * 4);
    std::memcpy(il, l.ilabel.data(), na * 4); std::memcpy(ol, l.olabel.data(), na * 4);
    std::memcpy(gc, l.graph_cost.data(), na * 4); std::memcpy(ac, l.ac_cost.data(), na * 4);
  }
  if (nf) { std::memcpy(fs, l.final_state.data(), nf * 4); std::memcpy(fg, l.final_graph.data(), nf * 4); }
  if (l.num_states) {
    std::memcpy(state_frame, l.state_frame.data(), l.num_states * 4);
    std::memcpy(state_graph, l.state_graph.data(), l.num_states * 4);
  }
}
void kdo_canonlat_copy_lattice(void *hv, int32_t *src, int32_t *dst, int32_t *il, int32_t *ol,
                               float *gc, float *ac, int32_t *fs, float *fg, int32_t *state_frame,
                               int32_t *state_graph) {
  CopyLattice(static_cast<CanonHandle *>(hv)->lat, src, dst, il, ol, gc, ac, fs, fg, state_frame, state_graph);
}
void kdo_latsimple_copy_lattice2(void *hv, int32_t *src, int32_t *dst, int32_t *il, int32_t *ol,
                                 float *gc, float *ac, int32_t *fs, float *fg, int32_t *state_frame,
                                 int32_t *state_graph) {
  CopyLattice(static_cast<LatHandle *>(hv)->lat, src, dst, il, ol, gc, ac, fs, fg, state_frame, state_graph);
}
int kdo_canonlat_best_path(void *hv, int32_t cap, int32_t *n, int32_t *il, int32_t *ol, int32_t *st,
                           float *gc, float *ac, float *final_w) {
  auto *h = static_cast<CanonHandle *>(hv);
  Path p;
  if (Guard([&] {
        Lattice lat;
        h->dec.GetRawLattice(&lat);
        LatticeShortestPath(lat, &p);
      }))
    return -1;
  return CopyPath(p, cap, n, il, ol, st, gc, ac, final_w);
}

// -- HashList (for the pin against oracle/_ref) -------------------------------------
void *kdo_hashlist_create() { return new HashList; }
void kdo_hashlist_free(void *h) { delete static_cast<HashList *>(h); }
void kdo_hashlist_set_size(void *h, int64_t n) { static_cast<HashList *>(h)->SetSize(static_cast<size_t>(n)); }
// returns 1 if newly inserted, 0 if the key was present
int32_t kdo_hashlist_insert(void *hv, int32_t key, int32_t val) {
  auto *h = static_cast<HashList *>(hv);
  int32_t e = h->Insert(key, val);
  return h->pool[e].val == val ? 1 : 0;
}
int32_t kdo_hashlist_find(void *hv, int32_t key) {
  auto *h = static_cast<HashList *>(hv);
  int32_t e = h->Find(key);
  return e == HashList::kNil ? -1 : h->pool[e].val;
}
// Clear(): writes the handed-over list (keys, vals) and frees its elements
int32_t kdo_hashlist_clear(void *hv, int32_t cap, int32_t *keys, int32_t *vals) {
  auto *h = static_cast<HashList *>(hv);
  int32_t n = 0;
  for (int32_t e = h->Clear(), nx; e != HashList::kNil; e = nx) {
    nx = h->pool[e].next;
    if (n < cap) { keys[n] = h->pool[e].key; vals[n] = h->pool[e].val; }
    n++;
    h->Delete(e);
  }
  return n;
}
int32_t kdo_hashlist_list(void *hv, int32_t cap, int32_t *keys, int32_t *vals) {
  auto *h = static_cast<HashList *>(hv);
  int32_t n = 0;
  for (int32_t e = h->GetList(); e != HashList::kNil; e = h->pool[e].next) {
    if (n < cap) { keys[n] = h->pool[e].key; vals[n] = h->pool[e].val; }
    n++;
  }
  return n;
}

}  // extern "C"
