// kaldi-decoder_amd/csrc/kd_api.hip -- host side of the C ABI in include/kd_hip.h:
// graph upload, batch state in HBM, kernel launches, result read-back.
// No search arithmetic happens here; if there is no usable GPU every entry point
// that would compute fails with KD_E_HIP (there is no CPU fallback).

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "kd_device.h"
#include "kd_hip.h"

using namespace kd;

namespace {

thread_local std::string g_err;

int Fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define KD_HIP_TRY(expr)                                                            \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      return Fail(KD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t Alloc(size_t count) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count;
    return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
  }
  hipError_t Reserve(size_t count) { return count <= n && p ? hipSuccess : Alloc(count); }
  size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

struct kd_graph {
  int device = 0;
  int32_t num_states = 0, start = -1;
  int64_t num_arcs = 0;
  DevBuf<StateRec> states;
  DevBuf<ArcRec> arcs;
};

struct kd_batch {
  const kd_graph *g = nullptr;
  kd_faster_options o{};
  int32_t max_batch = 0, batch = 0;
  hipStream_t stream = nullptr;
  int g_log2_slots = 17;
  // device state
  DevBuf<int32_t> tok_state, tok_prev, tok_arc;
  DevBuf<double> tok_cost;
  DevBuf<unsigned long long> arena_top;
  DevBuf<long long> frame_base;
  DevBuf<int32_t> frame_count, frames_decoded;
  DevBuf<uint32_t> status;
  DevBuf<long long> stats;
  DevBuf<uint32_t> g_key, g_arc, g_used, g_wl, g_log_h, g_log_arc, g_log_src;
  DevBuf<unsigned long long> g_cost, g_log_nk;
  // lattice mode
  bool lattice = false, finalized = false, lat_stats_valid = false, lat_sizes_valid = false;
  bool simple = false;
  DevBuf<float> final_rel;
  std::vector<float> h_final_rel;
  int32_t lat_use_final = -1;
  kd_lattice_options lo{};
  int64_t link_cap = 0;  // per utterance
  DevBuf<float> g_log_ac, g_log_gr, tok_extra, link_ac, link_graph, link_delta, lat_frc, lat_fbc;
  DevBuf<int32_t> link_src, link_dst, link_arc, link_emit_count, link_eps_count, tok_newid,
      lat_any_final, lat_sizes;
  DevBuf<long long> link_emit_base, link_eps_base, lat_state_off, lat_arc_off, lat_final_off;
  DevBuf<unsigned long long> link_top;
  DevBuf<uint32_t> lat_scratch;
  // GetRawLattice output: the ten arrays back to back in one device buffer, read
  // back with one copy into pinned host memory
  DevBuf<int32_t> lo_blob;
  int32_t *lo_host = nullptr;
  size_t lo_host_words = 0;
  size_t lo_na = 0, lo_ns = 0, lo_nf = 0;  // layout of the blob: 6 x na, 2 x ns, 2 x nf words
  std::vector<int32_t> h_lat_sizes;
  DevBuf<UttDesc> utt;
  DevBuf<float> staging;
  DevBuf<const float *> unused_;
  // path outputs
  DevBuf<int32_t> reached_final, path_len;
  DevBuf<long long> best_tok, path_offset;
  DevBuf<float> final_weight;
  DevBuf<int32_t> o_ilabel, o_olabel, o_state;
  DevBuf<float> o_graph, o_ac;
  // host mirrors
  std::vector<int32_t> h_decoded, h_path_len, h_reached;
  std::vector<uint32_t> h_status;
  std::vector<float> h_final_weight;
  std::vector<UttDesc> h_utt;
  bool initialized = false, path_valid = false, pending = false;
  int32_t path_use_final = -1;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;

  ~kd_batch() {
    if (lo_host) (void)hipHostFree(lo_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }

  size_t DeviceBytes() const {
    return tok_state.bytes() + tok_prev.bytes() + tok_arc.bytes() + tok_cost.bytes() +
           frame_base.bytes() + frame_count.bytes() + g_key.bytes() + g_arc.bytes() +
           g_log_h.bytes() + g_log_arc.bytes() + g_log_src.bytes() + g_log_nk.bytes() +
           g_used.bytes() + g_wl.bytes() + g_cost.bytes() +
           staging.bytes() + stats.bytes();
  }

  SearchParams Params(int do_init) const {
    SearchParams p{};
    p.states = g->states.p;
    p.arcs = g->arcs.p;
    p.start = g->start;
    p.beam = o.beam;
    p.max_active = o.max_active;
    p.min_active = o.min_active;
    p.beam_delta = o.beam_delta;
    p.tok_state = tok_state.p;
    p.tok_cost = tok_cost.p;
    p.tok_prev = tok_prev.p;
    p.tok_arc = tok_arc.p;
    p.arena_cap = static_cast<unsigned long long>(o.arena_tokens / std::max(max_batch, 1));
    p.arena_top = arena_top.p;
    p.frame_base = frame_base.p;
    p.frame_count = frame_count.p;
    p.max_frames = o.max_frames;
    p.frames_decoded = frames_decoded.p;
    p.status = status.p;
    p.stats = stats.p;
    p.g_key = g_key.p;
    p.g_cost = g_cost.p;
    p.g_arc = g_arc.p;
    p.g_used = g_used.p;
    p.g_wl = g_wl.p;
    p.g_log_h = g_log_h.p;
    p.g_log_arc = g_log_arc.p;
    p.g_log_src = g_log_src.p;
    p.g_log_nk = g_log_nk.p;
    p.g_log2_slots = g_log2_slots;
    p.utt = utt.p;
    p.do_init = do_init;
    p.lattice = lattice ? 1 : 0;
    p.simple = simple ? 1 : 0;
    p.g_log_ac = g_log_ac.p;
    p.g_log_gr = g_log_gr.p;
    p.tok_extra = tok_extra.p;
    p.link_src = link_src.p;
    p.link_dst = link_dst.p;
    p.link_arc = link_arc.p;
    p.link_ac = link_ac.p;
    p.link_graph = link_graph.p;
    p.link_delta = link_delta.p;
    p.link_cap = static_cast<unsigned long long>(link_cap);
    p.link_top = link_top.p;
    p.link_emit_base = link_emit_base.p;
    p.link_emit_count = link_emit_count.p;
    p.link_eps_base = link_eps_base.p;
    p.link_eps_count = link_eps_count.p;
    return p;
  }

  LatticeParams LParams() const {
    LatticeParams p{};
    p.states = g->states.p;
    p.arcs = g->arcs.p;
    p.start = g->start;
    p.lattice_beam = lo.lattice_beam;
    p.tok_state = tok_state.p;
    p.tok_cost = tok_cost.p;
    p.tok_extra = tok_extra.p;
    p.tok_newid = tok_newid.p;
    p.frame_base = frame_base.p;
    p.frame_count = frame_count.p;
    p.max_frames = o.max_frames;
    p.frames_decoded = frames_decoded.p;
    p.link_src = link_src.p;
    p.link_dst = link_dst.p;
    p.link_arc = link_arc.p;
    p.link_ac = link_ac.p;
    p.link_graph = link_graph.p;
    p.link_delta = link_delta.p;
    p.link_emit_base = link_emit_base.p;
    p.link_emit_count = link_emit_count.p;
    p.link_eps_base = link_eps_base.p;
    p.link_eps_count = link_eps_count.p;
    p.scratch = lat_scratch.p;
    p.scratch_stride = static_cast<long long>(o.arena_tokens / std::max(max_batch, 1));
    p.final_relative_cost = lat_frc.p;
    p.final_best_cost = lat_fbc.p;
    p.any_final = lat_any_final.p;
    p.finalized = finalized ? 1 : 0;
    p.sizes = lat_sizes.p;
    p.out_state_off = lat_state_off.p;
    p.out_arc_off = lat_arc_off.p;
    p.out_final_off = lat_final_off.p;
    int32_t *w = lo_blob.p;
    p.o_src = w; p.o_dst = w + lo_na; p.o_ilabel = w + 2 * lo_na; p.o_olabel = w + 3 * lo_na;
    p.o_graph = reinterpret_cast<float *>(w + 4 * lo_na);
    p.o_ac = reinterpret_cast<float *>(w + 5 * lo_na);
    w += 6 * lo_na;
    p.o_state_frame = w; p.o_state_graph = w + lo_ns;
    w += 2 * lo_ns;
    p.o_final_state = w; p.o_final_graph = reinterpret_cast<float *>(w + lo_nf);
    return p;
  }

  PathParams PParams(int use_final) const {
    PathParams p{};
    p.states = g->states.p;
    p.arcs = g->arcs.p;
    p.tok_state = tok_state.p;
    p.tok_cost = tok_cost.p;
    p.tok_prev = tok_prev.p;
    p.tok_arc = tok_arc.p;
    p.frame_base = frame_base.p;
    p.frame_count = frame_count.p;
    p.max_frames = o.max_frames;
    p.frames_decoded = frames_decoded.p;
    p.use_final_probs = use_final;
    p.batch = batch;
    p.simple = simple ? 1 : 0;
    p.no_backptr = lattice ? 1 : 0;
    p.tok_ac = tok_extra.p;
    p.final_rel = final_rel.p;
    p.reached_final = reached_final.p;
    p.best_tok = best_tok.p;
    p.path_len = path_len.p;
    p.final_weight = final_weight.p;
    p.path_offset = path_offset.p;
    p.o_ilabel = o_ilabel.p;
    p.o_olabel = o_olabel.p;
    p.o_state = o_state.p;
    p.o_graph = o_graph.p;
    p.o_ac = o_ac.p;
    return p;
  }
};

namespace {

int CheckOptions(const kd_faster_options &o) {
  // faster-decoder.cc:24-28
  if (!(o.hash_ratio >= 1.0f)) return Fail(KD_E_INVALID, "Check failed: config_.hash_ratio >= 1.0");
  if (!(o.max_active > 1)) return Fail(KD_E_INVALID, "Check failed: config_.max_active > 1");
  if (!(o.min_active >= 0 && o.min_active < o.max_active))
    return Fail(KD_E_INVALID, "Check failed: config_.min_active >= 0 && config_.min_active < config_.max_active");
  return KD_OK;
}

int ReadBack(kd_batch *b) {
  KD_HIP_TRY(hipMemcpyAsync(b->h_decoded.data(), b->frames_decoded.p, b->batch * sizeof(int32_t),
                            hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->h_status.data(), b->status.p, b->batch * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  b->pending = false;
  for (int i = 0; i < b->batch; ++i) {
    if (b->h_status[i]) {
      std::string m = "utterance " + std::to_string(i) + " stopped at frame " +
                      std::to_string(b->h_decoded[i]) + ":";
      if (b->h_status[i] & KD_UTT_ARENA_FULL) m += " token arena full (raise arena_tokens)";
      if (b->h_status[i] & KD_UTT_TABLE_FULL) m += " token table full (raise table_slots)";
      if (b->h_status[i] & KD_UTT_FRAMES_FULL) m += " frame index full (raise max_frames)";
      return Fail(KD_E_CAPACITY, m);
    }
  }
  return KD_OK;
}

int Launch(kd_batch *b, int do_init, int32_t num_cols, int64_t row_stride) {
  SearchParams p = b->Params(do_init);
  p.num_cols = num_cols;
  p.row_stride = row_stride;
  KD_HIP_TRY(hipEventRecord(b->ev0, b->stream));
  KD_HIP_TRY(LaunchSearch(p, b->batch, b->stream));
  KD_HIP_TRY(hipEventRecord(b->ev1, b->stream));
  b->timed = true;
  b->pending = true;
  b->path_valid = false;
  b->lat_stats_valid = false;
  b->lat_sizes_valid = false;
  return KD_OK;
}

// Validates the per-utterance frame windows and fills b->h_utt (ll pointers are
// set by the caller).  faster-decoder.cc:126-144.
int PlanAdvance(kd_batch *b, const int32_t *num_rows, const int32_t *offset,
                int32_t max_num_frames) {
  if (!b->initialized)
    return Fail(KD_E_STATE, "Check failed: num_frames_decoded_ >= 0 && "
                            "\"You must call InitDecoding() before AdvanceDecoding()\"");
  if (b->finalized)
    return Fail(KD_E_STATE, "FinalizeDecoding() has been called: cannot decode further");
  for (int i = 0; i < b->batch; ++i) {
    const int32_t off = offset ? offset[i] : 0;
    const int32_t ready = off + num_rows[i];
    if (!(ready >= b->h_decoded[i]))
      return Fail(KD_E_INVALID, "Check failed: num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, b->h_decoded[i] + max_num_frames);
    if (target > b->h_decoded[i] && b->h_decoded[i] < off)
      return Fail(KD_E_INVALID, "decodable offset is beyond the next frame to decode");
    if (target > b->o.max_frames) return Fail(KD_E_CAPACITY, "frame index full (raise max_frames)");
    b->h_utt[i].num_rows = num_rows[i];
    b->h_utt[i].offset = off;
    b->h_utt[i].target = target;
  }
  return KD_OK;
}

}  // namespace

extern "C" {

const char *kd_last_error(void) { return g_err.c_str(); }

int kd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int kd_graph_create(int device, int32_t num_states, int32_t start, const int32_t *row_ptr,
                    const int32_t *ilabel, const int32_t *olabel, const float *weight,
                    const int32_t *nextstate, const float *final_cost, kd_graph **out) {
  if (!out) return Fail(KD_E_INVALID, "out is NULL");
  *out = nullptr;
  if (num_states <= 0 || !row_ptr) return Fail(KD_E_INVALID, "empty graph");
  if (start < 0 || start >= num_states)
    return Fail(KD_E_INVALID, "Check failed: start_state != fst::kNoStateId");
  const int64_t na = row_ptr[num_states];
  std::vector<uint8_t> has_eps(num_states, 0);
  for (int32_t s = 0; s < num_states; ++s) {
    if (row_ptr[s + 1] < row_ptr[s]) return Fail(KD_E_INVALID, "row_ptr not monotone");
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
      if (nextstate[a] < 0 || nextstate[a] >= num_states)
        return Fail(KD_E_INVALID, "arc nextstate out of range");
      if (ilabel[a] < 0) return Fail(KD_E_INVALID, "negative ilabel");
      if (ilabel[a] == 0) has_eps[s] = 1;
    }
  }
  std::vector<StateRec> st(num_states);
  std::vector<ArcRec> ar(std::max<int64_t>(na, 1));
  for (int32_t s = 0; s < num_states; ++s) {
    const int32_t b = row_ptr[s], e = row_ptr[s + 1];
    int32_t pe = b;
    for (int32_t a = b; a < e; ++a)
      if (ilabel[a] != 0) ++pe;
    st[s] = StateRec{b, pe, e, final_cost[s]};
    int32_t we = b, wn = pe;
    for (int32_t a = b; a < e; ++a) {
      ArcRec r{ilabel[a], olabel[a], weight[a],
               static_cast<uint32_t>(nextstate[a]) | (has_eps[nextstate[a]] ? kHasEps : 0u)};
      ar[ilabel[a] != 0 ? we++ : wn++] = r;
    }
  }
  KD_HIP_TRY(hipSetDevice(device));
  kd_graph *g = new (std::nothrow) kd_graph;
  if (!g) return Fail(KD_E_INVALID, "out of host memory");
  g->device = device;
  g->num_states = num_states;
  g->start = start;
  g->num_arcs = na;
  hipError_t e = g->states.Alloc(num_states);
  if (e == hipSuccess) e = g->arcs.Alloc(ar.size());
  if (e == hipSuccess) e = hipMemcpy(g->states.p, st.data(), st.size() * sizeof(StateRec), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->arcs.p, ar.data(), ar.size() * sizeof(ArcRec), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    delete g;
    return Fail(KD_E_HIP, std::string("graph upload: ") + hipGetErrorString(e));
  }
  *out = g;
  return KD_OK;
}

void kd_graph_destroy(kd_graph *g) { delete g; }
int32_t kd_graph_num_states(const kd_graph *g) { return g ? g->num_states : 0; }
int64_t kd_graph_num_arcs(const kd_graph *g) { return g ? g->num_arcs : 0; }
size_t kd_graph_device_bytes(const kd_graph *g) { return g ? g->states.bytes() + g->arcs.bytes() : 0; }

void kd_faster_options_default(kd_faster_options *o) {
  // faster-decoder.h:41-49
  o->beam = 16.0f;
  o->max_active = std::numeric_limits<int32_t>::max();
  o->min_active = 20;
  o->beam_delta = 0.5f;
  o->hash_ratio = 2.0f;
  o->arena_tokens = 0;
  o->table_slots = 0;
  o->max_frames = 0;
}

static int CreateBatch(const kd_graph *g, const kd_faster_options *opts,
                       const kd_lattice_options *lopts, int32_t max_batch, void *stream,
                       kd_batch **out) {
  KD_HIP_TRY(hipSetDevice(g->device));
  kd_batch *b = new (std::nothrow) kd_batch;
  if (!b) return Fail(KD_E_INVALID, "out of host memory");
  b->g = g;
  b->o = *opts;
  b->max_batch = max_batch;
  b->stream = static_cast<hipStream_t>(stream);
  if (b->o.arena_tokens <= 0)
    b->o.arena_tokens = std::min<int64_t>((1ll << 31) - 1, static_cast<int64_t>(max_batch) << 21);
  b->o.arena_tokens = std::min<int64_t>(b->o.arena_tokens, (1ll << 31) - 1);
  if (b->o.max_frames <= 0) b->o.max_frames = 8192;
  int slots = b->o.table_slots <= 0 ? (1 << 17) : b->o.table_slots;
  int lg = 14;
  while ((1 << lg) < slots && lg < 19) ++lg;
  b->g_log2_slots = lg;
  b->o.table_slots = 1 << lg;
  const size_t gs = static_cast<size_t>(max_batch) << lg;
  const size_t nf = static_cast<size_t>(max_batch) * (b->o.max_frames + 1);
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(b->tok_state.Alloc(b->o.arena_tokens));
  A(b->tok_prev.Alloc(b->o.arena_tokens));
  A(b->tok_arc.Alloc(b->o.arena_tokens));
  A(b->tok_cost.Alloc(b->o.arena_tokens));
  A(b->arena_top.Alloc(max_batch));
  A(b->frame_base.Alloc(nf));
  A(b->frame_count.Alloc(nf));
  A(b->frames_decoded.Alloc(max_batch));
  A(b->status.Alloc(max_batch));
  A(b->stats.Alloc(static_cast<size_t>(max_batch) * kStatStride));
  A(b->g_key.Alloc(gs));
  A(b->g_cost.Alloc(gs));
  A(b->g_arc.Alloc(gs));
  A(b->g_used.Alloc(gs));
  A(b->g_wl.Alloc(2 * gs));
  A(b->g_log_h.Alloc(2 * gs));
  A(b->g_log_arc.Alloc(2 * gs));
  A(b->g_log_src.Alloc(2 * gs));
  A(b->g_log_nk.Alloc(2 * gs));
  A(b->utt.Alloc(max_batch));
  A(b->reached_final.Alloc(max_batch));
  A(b->path_len.Alloc(max_batch));
  A(b->best_tok.Alloc(max_batch));
  A(b->path_offset.Alloc(max_batch));
  A(b->final_weight.Alloc(2 * static_cast<size_t>(max_batch)));
  A(b->final_rel.Alloc(max_batch));
  if (e == hipSuccess) e = hipMemset(b->g_key.p, 0xFF, b->g_key.bytes());
  if (e == hipSuccess) e = hipMemset(b->g_cost.p, 0xFF, b->g_cost.bytes());
  if (e == hipSuccess) e = hipMemset(b->g_arc.p, 0xFF, b->g_arc.bytes());
  if (e == hipSuccess) e = hipMemset(b->frames_decoded.p, 0xFF, b->frames_decoded.bytes());
  if (e == hipSuccess) e = hipMemset(b->status.p, 0, b->status.bytes());
  if (e == hipSuccess) e = hipMemset(b->stats.p, 0, b->stats.bytes());
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e != hipSuccess) {
    delete b;
    return Fail(KD_E_HIP, std::string("batch allocation: ") + hipGetErrorString(e));
  }
  b->h_decoded.assign(max_batch, -1);
  b->h_status.assign(max_batch, 0);
  b->h_path_len.assign(max_batch, -1);
  b->h_reached.assign(max_batch, 0);
  b->h_final_weight.assign(2 * static_cast<size_t>(max_batch), 0.f);
  b->h_final_rel.assign(max_batch, 0.f);
  b->h_utt.assign(max_batch, UttDesc{});
  if (lopts) {
    b->lattice = true;
    b->lo = *lopts;
    int64_t links = lopts->arena_links > 0 ? lopts->arena_links : 4 * b->o.arena_tokens;
    links = std::min<int64_t>(links, (1ll << 31) - 1);
    b->link_cap = links / max_batch;
    const size_t nl = static_cast<size_t>(b->link_cap) * max_batch;
    hipError_t e2 = hipSuccess;
    auto A2 = [&](hipError_t r) { if (e2 == hipSuccess) e2 = r; };
    A2(b->g_log_ac.Alloc(2 * gs));
    A2(b->g_log_gr.Alloc(2 * gs));
    A2(b->tok_extra.Alloc(b->o.arena_tokens));
    A2(b->tok_newid.Alloc(b->o.arena_tokens));
    A2(b->lat_scratch.Alloc(b->o.arena_tokens));
    A2(b->link_src.Alloc(nl));
    A2(b->link_dst.Alloc(nl));
    A2(b->link_arc.Alloc(nl));
    A2(b->link_ac.Alloc(nl));
    A2(b->link_graph.Alloc(nl));
    A2(b->link_delta.Alloc(nl));
    A2(b->link_top.Alloc(max_batch));
    A2(b->link_emit_base.Alloc(nf));
    A2(b->link_emit_count.Alloc(nf));
    A2(b->link_eps_base.Alloc(nf));
    A2(b->link_eps_count.Alloc(nf));
    A2(b->lat_frc.Alloc(max_batch));
    A2(b->lat_fbc.Alloc(max_batch));
    A2(b->lat_any_final.Alloc(max_batch));
    A2(b->lat_sizes.Alloc(4 * static_cast<size_t>(max_batch)));
    A2(b->lat_state_off.Alloc(max_batch));
    A2(b->lat_arc_off.Alloc(max_batch));
    A2(b->lat_final_off.Alloc(max_batch));
    if (e2 == hipSuccess) e2 = hipMemset(b->link_emit_count.p, 0, b->link_emit_count.bytes());
    if (e2 == hipSuccess) e2 = hipMemset(b->link_eps_count.p, 0, b->link_eps_count.bytes());
    if (e2 != hipSuccess) {
      delete b;
      return Fail(KD_E_HIP, std::string("lattice batch allocation: ") + hipGetErrorString(e2));
    }
    b->h_lat_sizes.assign(4 * static_cast<size_t>(max_batch), 0);
  }
  *out = b;
  return KD_OK;
}

int kd_batch_create(const kd_graph *g, const kd_faster_options *opts, int32_t max_batch,
                    void *stream, kd_batch **out) {
  if (!out) return Fail(KD_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!g || !opts || max_batch <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (int r = CheckOptions(*opts)) return r;
  return CreateBatch(g, opts, nullptr, max_batch, stream, out);
}

int kd_batch_create_simple(const kd_graph *g, float beam, int64_t arena_tokens,
                           int32_t table_slots, int32_t max_frames, int32_t max_batch,
                           void *stream, kd_batch **out) {
  if (!out) return Fail(KD_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!g || max_batch <= 0) return Fail(KD_E_INVALID, "bad argument");
  kd_faster_options fo;
  kd_faster_options_default(&fo);
  fo.beam = beam;
  fo.min_active = 0;
  fo.arena_tokens = arena_tokens;
  fo.table_slots = table_slots;
  fo.max_frames = max_frames;
  if (int r = CreateBatch(g, &fo, nullptr, max_batch, stream, out)) return r;
  kd_batch *b = *out;
  const size_t gs = static_cast<size_t>(max_batch) << b->g_log2_slots;
  hipError_t e = b->g_log_ac.Alloc(2 * gs);
  if (e == hipSuccess) e = b->tok_extra.Alloc(b->o.arena_tokens);
  if (e != hipSuccess) {
    delete b;
    *out = nullptr;
    return Fail(KD_E_HIP, std::string("simple batch allocation: ") + hipGetErrorString(e));
  }
  b->simple = true;
  return KD_OK;
}

void kd_lattice_options_default(kd_lattice_options *o) {
  // lattice-faster-decoder.h:48-66
  o->beam = 16.0f;
  o->max_active = std::numeric_limits<int32_t>::max();
  o->min_active = 200;
  o->lattice_beam = 10.0f;
  o->prune_interval = 25;
  o->beam_delta = 0.5f;
  o->hash_ratio = 2.0f;
  o->prune_scale = 0.1f;
  o->arena_tokens = 0;
  o->arena_links = 0;
  o->table_slots = 0;
  o->max_frames = 0;
}

int kd_batch_create_lattice(const kd_graph *g, const kd_lattice_options *lo, int32_t max_batch,
                            void *stream, kd_batch **out) {
  if (!out) return Fail(KD_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!g || !lo || max_batch <= 0) return Fail(KD_E_INVALID, "bad argument");
  // lattice-faster-decoder.h:128-133
  if (!(lo->beam > 0.0 && lo->max_active > 1 && lo->lattice_beam > 0.0 &&
        lo->min_active <= lo->max_active && lo->prune_interval > 0 && lo->beam_delta > 0.0 &&
        lo->hash_ratio >= 1.0 && lo->prune_scale > 0.0 && lo->prune_scale < 1.0))
    return Fail(KD_E_INVALID,
                "Check failed: beam > 0.0 && max_active > 1 && lattice_beam > 0.0 && "
                "min_active <= max_active && prune_interval > 0 && beam_delta > 0.0 && "
                "hash_ratio >= 1.0 && prune_scale > 0.0 && prune_scale < 1.0");
  if (lo->min_active < 0) return Fail(KD_E_INVALID, "Check failed: min_active >= 0");
  kd_faster_options fo;
  kd_faster_options_default(&fo);
  fo.beam = lo->beam;
  fo.max_active = lo->max_active;
  fo.min_active = lo->min_active;
  fo.beam_delta = lo->beam_delta;
  fo.hash_ratio = lo->hash_ratio;
  fo.arena_tokens = lo->arena_tokens;
  fo.table_slots = lo->table_slots;
  fo.max_frames = lo->max_frames;
  return CreateBatch(g, &fo, lo, max_batch, stream, out);
}

void kd_batch_destroy(kd_batch *b) {
  if (!b) return;
  (void)hipStreamSynchronize(b->stream);
  delete b;
}

int kd_batch_set_options(kd_batch *b, const kd_faster_options *opts) {
  if (!b || !opts) return Fail(KD_E_INVALID, "bad argument");
  // the reference's SetOptions does not re-validate (faster-decoder.h:78)
  b->o.beam = opts->beam;
  b->o.max_active = opts->max_active;
  b->o.min_active = opts->min_active;
  b->o.beam_delta = opts->beam_delta;
  b->o.hash_ratio = opts->hash_ratio;
  return KD_OK;
}

int kd_batch_init_decoding(kd_batch *b, int32_t batch) {
  if (!b || batch <= 0 || batch > b->max_batch) return Fail(KD_E_INVALID, "bad batch size");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  b->batch = batch;
  KD_HIP_TRY(hipMemsetAsync(b->arena_top.p, 0, b->arena_top.bytes(), b->stream));
  if (b->lattice) KD_HIP_TRY(hipMemsetAsync(b->link_top.p, 0, b->link_top.bytes(), b->stream));
  KD_HIP_TRY(hipMemsetAsync(b->frames_decoded.p, 0xFF, b->frames_decoded.bytes(), b->stream));
  KD_HIP_TRY(hipMemsetAsync(b->status.p, 0, b->status.bytes(), b->stream));
  KD_HIP_TRY(hipMemsetAsync(b->stats.p, 0, b->stats.bytes(), b->stream));
  b->finalized = false;
  for (auto &u : b->h_utt) u = UttDesc{};
  KD_HIP_TRY(hipMemcpyAsync(b->utt.p, b->h_utt.data(), batch * sizeof(UttDesc),
                            hipMemcpyHostToDevice, b->stream));
  if (int r = Launch(b, 1, 0, 0)) return r;
  if (int r = ReadBack(b)) return r;
  b->initialized = true;
  return KD_OK;
}

int kd_batch_advance_async(kd_batch *b, const float *const *loglikes_dev, const int32_t *num_rows,
                           const int32_t *offset, int32_t num_cols, int64_t row_stride,
                           int32_t max_num_frames) {
  if (!b || !loglikes_dev || !num_rows || num_cols <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (row_stride < num_cols) return Fail(KD_E_INVALID, "row_stride < num_cols");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (int r = PlanAdvance(b, num_rows, offset, max_num_frames)) return r;
  for (int i = 0; i < b->batch; ++i) b->h_utt[i].ll = loglikes_dev[i];
  KD_HIP_TRY(hipMemcpyAsync(b->utt.p, b->h_utt.data(), b->batch * sizeof(UttDesc),
                            hipMemcpyHostToDevice, b->stream));
  return Launch(b, 0, num_cols, row_stride);
}

int kd_batch_sync(kd_batch *b) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  return ReadBack(b);
}

int kd_batch_advance(kd_batch *b, const float *const *loglikes, const int32_t *num_rows,
                     const int32_t *offset, int32_t num_cols, int64_t row_stride, int mem_kind,
                     int32_t max_num_frames) {
  if (!b || !loglikes || !num_rows || num_cols <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (mem_kind == KD_MEM_DEVICE) {
    if (int r = kd_batch_advance_async(b, loglikes, num_rows, offset, num_cols, row_stride,
                                       max_num_frames))
      return r;
    return ReadBack(b);
  }
  if (mem_kind != KD_MEM_HOST) return Fail(KD_E_INVALID, "bad mem_kind");
  if (row_stride < num_cols) return Fail(KD_E_INVALID, "row_stride < num_cols");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (int r = PlanAdvance(b, num_rows, offset, max_num_frames)) return r;
  // stage only the rows this call will read, packed to num_cols
  size_t total = 0;
  std::vector<size_t> first(b->batch), count(b->batch);
  for (int i = 0; i < b->batch; ++i) {
    const UttDesc &u = b->h_utt[i];
    const int32_t f0 = b->h_decoded[i], f1 = u.target;
    first[i] = f1 > f0 ? static_cast<size_t>(f0 - u.offset) : 0;
    count[i] = f1 > f0 ? static_cast<size_t>(f1 - f0) : 0;
    total += count[i] * num_cols;
  }
  KD_HIP_TRY(b->staging.Reserve(total));
  size_t pos = 0;
  for (int i = 0; i < b->batch; ++i) {
    if (count[i]) {
      KD_HIP_TRY(hipMemcpy2DAsync(b->staging.p + pos, num_cols * sizeof(float),
                                  loglikes[i] + first[i] * row_stride, row_stride * sizeof(float),
                                  num_cols * sizeof(float), count[i], hipMemcpyHostToDevice,
                                  b->stream));
    }
    // row (frame - offset) of the caller's matrix == row (frame - f0) of the staged block
    b->h_utt[i].ll = b->staging.p + pos - first[i] * static_cast<size_t>(num_cols);
    pos += count[i] * num_cols;
  }
  KD_HIP_TRY(hipMemcpyAsync(b->utt.p, b->h_utt.data(), b->batch * sizeof(UttDesc),
                            hipMemcpyHostToDevice, b->stream));
  if (int r = Launch(b, 0, num_cols, num_cols)) return r;
  return ReadBack(b);
}

int kd_batch_num_frames_decoded(kd_batch *b, int32_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  for (int i = 0; i < b->batch; ++i) out[i] = b->h_decoded[i];
  return KD_OK;
}

int kd_batch_status(kd_batch *b, uint32_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (b->pending) (void)ReadBack(b);
  for (int i = 0; i < b->batch; ++i) out[i] = b->h_status[i];
  return KD_OK;
}

static int PreparePath(kd_batch *b, int32_t use_final_probs) {
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (b->path_valid && b->path_use_final == use_final_probs) return KD_OK;
  KD_HIP_TRY(LaunchPathPrepare(b->PParams(use_final_probs), b->batch, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->h_path_len.data(), b->path_len.p, b->batch * sizeof(int32_t),
                            hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->h_reached.data(), b->reached_final.p, b->batch * sizeof(int32_t),
                            hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->h_final_weight.data(), b->final_weight.p,
                            2 * b->batch * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->h_final_rel.data(), b->final_rel.p, b->batch * sizeof(float),
                            hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  b->path_valid = true;
  b->path_use_final = use_final_probs;
  return KD_OK;
}

int kd_batch_reached_final(kd_batch *b, int32_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (int r = PreparePath(b, b->path_valid ? b->path_use_final : 1)) return r;
  for (int i = 0; i < b->batch; ++i) out[i] = b->h_reached[i];
  return KD_OK;
}

int kd_batch_best_path_prepare(kd_batch *b, int32_t use_final_probs, int32_t *path_len) {
  if (!b || !path_len) return Fail(KD_E_INVALID, "bad argument");
  // lattice-simple-decoder.cc:574-580: the lattice decoders' best path is the shortest
  // path through the raw lattice; their tokens carry forward links, not back-pointers
  if (b->lattice)
    return Fail(KD_E_STATE, "a lattice batch has no back-pointers: use kd_batch_raw_lattice_*");
  if (int r = PreparePath(b, use_final_probs ? 1 : 0)) return r;
  for (int i = 0; i < b->batch; ++i) path_len[i] = b->h_path_len[i];
  return KD_OK;
}

int kd_batch_best_path_copy(kd_batch *b, const int64_t *path_offset, int32_t *ilabel,
                            int32_t *olabel, int32_t *state, float *graph_cost, float *ac_cost,
                            float *final_weight) {
  if (!b || !path_offset) return Fail(KD_E_INVALID, "bad argument");
  if (!b->path_valid) return Fail(KD_E_STATE, "call kd_batch_best_path_prepare first");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  // device-side packing is dense in utterance order; scatter to the caller's offsets on the host
  std::vector<long long> doff(b->batch);
  size_t total = 0;
  for (int i = 0; i < b->batch; ++i) {
    doff[i] = static_cast<long long>(total);
    total += b->h_path_len[i] > 0 ? b->h_path_len[i] : 0;
  }
  if (final_weight)
    std::memcpy(final_weight, b->h_final_weight.data(), 2 * b->batch * sizeof(float));
  if (total == 0) return KD_OK;
  KD_HIP_TRY(b->o_ilabel.Reserve(total));
  KD_HIP_TRY(b->o_olabel.Reserve(total));
  KD_HIP_TRY(b->o_state.Reserve(total));
  KD_HIP_TRY(b->o_graph.Reserve(total));
  KD_HIP_TRY(b->o_ac.Reserve(total));
  KD_HIP_TRY(hipMemcpyAsync(b->path_offset.p, doff.data(), b->batch * sizeof(long long),
                            hipMemcpyHostToDevice, b->stream));
  KD_HIP_TRY(LaunchPathFill(b->PParams(b->path_use_final), b->batch, b->stream));
  std::vector<int32_t> hi(total), ho(total), hs(total);
  std::vector<float> hg(total), ha(total);
  KD_HIP_TRY(hipMemcpyAsync(hi.data(), b->o_ilabel.p, total * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(ho.data(), b->o_olabel.p, total * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(hs.data(), b->o_state.p, total * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(hg.data(), b->o_graph.p, total * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(ha.data(), b->o_ac.p, total * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  for (int i = 0; i < b->batch; ++i) {
    const int32_t n = b->h_path_len[i];
    if (n <= 0) continue;
    const size_t s = static_cast<size_t>(doff[i]), d = static_cast<size_t>(path_offset[i]);
    if (ilabel) std::memcpy(ilabel + d, hi.data() + s, n * 4);
    if (olabel) std::memcpy(olabel + d, ho.data() + s, n * 4);
    if (state) std::memcpy(state + d, hs.data() + s, n * 4);
    if (graph_cost) std::memcpy(graph_cost + d, hg.data() + s, n * 4);
    if (ac_cost) std::memcpy(ac_cost + d, ha.data() + s, n * 4);
  }
  return KD_OK;
}

int kd_batch_frame_tokens(kd_batch *b, int32_t utt, int32_t frame_plus_one, int32_t cap,
                          int32_t *count, int32_t *state, double *cost) {
  if (!b || !count || utt < 0 || utt >= b->batch) return Fail(KD_E_INVALID, "bad argument");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (frame_plus_one < 0 || frame_plus_one > b->h_decoded[utt])
    return Fail(KD_E_INVALID, "frame not decoded");
  const size_t fi = static_cast<size_t>(utt) * (b->o.max_frames + 1) + frame_plus_one;
  long long base = 0;
  int32_t n = 0;
  KD_HIP_TRY(hipMemcpy(&base, b->frame_base.p + fi, sizeof(base), hipMemcpyDeviceToHost));
  KD_HIP_TRY(hipMemcpy(&n, b->frame_count.p + fi, sizeof(n), hipMemcpyDeviceToHost));
  *count = n;
  if (cap == 0 || n == 0) return KD_OK;
  if (cap < n || !state || !cost) return Fail(KD_E_INVALID, "buffer too small");
  std::vector<int32_t> s(n);
  std::vector<double> c(n);
  KD_HIP_TRY(hipMemcpy(s.data(), b->tok_state.p + base, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  KD_HIP_TRY(hipMemcpy(c.data(), b->tok_cost.p + base, n * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int32_t> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return s[x] < s[y]; });
  for (int i = 0; i < n; ++i) { state[i] = s[order[i]]; cost[i] = c[order[i]]; }
  return KD_OK;
}

static int PreparePath(kd_batch *b, int32_t use_final_probs);

static int LatticeStats(kd_batch *b) {
  if (b->lat_stats_valid) return KD_OK;
  LatticeParams p = b->LParams();
  p.stats_only = 1;
  KD_HIP_TRY(LaunchLatticePrune(p, b->batch, b->stream));
  b->lat_stats_valid = true;
  return KD_OK;
}

int kd_batch_finalize_decoding(kd_batch *b) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  if (b->finalized) return Fail(KD_E_STATE, "Check failed: !decoding_finalized_");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  LatticeParams p = b->LParams();
  p.stats_only = 0;
  KD_HIP_TRY(LaunchLatticePrune(p, b->batch, b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  b->finalized = true;
  b->lat_stats_valid = true;
  b->lat_sizes_valid = false;
  return KD_OK;
}

int kd_batch_final_relative_cost(kd_batch *b, float *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  if (!b->lattice) {  // SimpleDecoder::FinalRelativeCost, simple-decoder.cc:78-100
    if (int r = PreparePath(b, b->path_valid ? b->path_use_final : 1)) return r;
    for (int i = 0; i < b->batch; ++i) out[i] = b->h_final_rel[i];
    return KD_OK;
  }
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (int r = LatticeStats(b)) return r;
  KD_HIP_TRY(hipMemcpyAsync(out, b->lat_frc.p, b->batch * sizeof(float), hipMemcpyDeviceToHost,
                            b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  return KD_OK;
}

int kd_batch_raw_lattice_prepare(kd_batch *b, int32_t use_final_probs, int32_t *sizes) {
  if (!b || !sizes) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  if (b->finalized && !use_final_probs)  // lattice-simple-decoder.cc:594-597
    return Fail(KD_E_STATE, "You cannot call FinalizeDecoding() and then call "
                            "GetRawLattice() with use_final_probs == false");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  for (int i = 0; i < b->batch; ++i)
    if (!(b->h_decoded[i] > 0)) return Fail(KD_E_STATE, "Check failed: num_frames > 0");
  if (int r = LatticeStats(b)) return r;
  if (!(b->lat_sizes_valid && b->lat_use_final == (use_final_probs ? 1 : 0))) {
    LatticeParams p = b->LParams();
    p.use_final_probs = use_final_probs ? 1 : 0;
    KD_HIP_TRY(LaunchLatticeCount(p, b->batch, b->stream));
    KD_HIP_TRY(hipMemcpyAsync(b->h_lat_sizes.data(), b->lat_sizes.p, 4 * b->batch * sizeof(int32_t),
                              hipMemcpyDeviceToHost, b->stream));
    KD_HIP_TRY(hipStreamSynchronize(b->stream));
    b->lat_sizes_valid = true;
    b->lat_use_final = use_final_probs ? 1 : 0;
  }
  std::memcpy(sizes, b->h_lat_sizes.data(), 4 * b->batch * sizeof(int32_t));
  return KD_OK;
}

int kd_batch_raw_lattice_copy(kd_batch *b, const int64_t *state_off, const int64_t *arc_off,
                              const int64_t *final_off, int32_t *src, int32_t *dst,
                              int32_t *ilabel, int32_t *olabel, float *graph_cost,
                              float *ac_cost, int32_t *state_frame, int32_t *state_graph,
                              int32_t *final_state, float *final_graph) {
  if (!b || !state_off || !arc_off || !final_off) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice || !b->lat_sizes_valid)
    return Fail(KD_E_STATE, "call kd_batch_raw_lattice_prepare first");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  std::vector<long long> so(b->batch), ao(b->batch), fo(b->batch);
  size_t ns = 0, na = 0, nfn = 0;
  for (int i = 0; i < b->batch; ++i) {
    so[i] = static_cast<long long>(ns);
    ao[i] = static_cast<long long>(na);
    fo[i] = static_cast<long long>(nfn);
    ns += b->h_lat_sizes[4 * i];
    na += b->h_lat_sizes[4 * i + 1];
    nfn += b->h_lat_sizes[4 * i + 2];
  }
  if (ns == 0) return KD_OK;
  const size_t words = 6 * na + 2 * ns + 2 * nfn;
  b->lo_na = na;
  b->lo_ns = ns;
  b->lo_nf = nfn;
  if (hipError_t e = b->lo_blob.Reserve(words); e != hipSuccess)
    return Fail(KD_E_HIP, std::string("lattice output: ") + hipGetErrorString(e));
  if (b->lo_host_words < words) {
    if (b->lo_host) (void)hipHostFree(b->lo_host);
    b->lo_host = nullptr;
    b->lo_host_words = 0;
    const size_t want = words + words / 2;
    if (hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&b->lo_host), want * 4, hipHostMallocDefault);
        e != hipSuccess)
      return Fail(KD_E_HIP, std::string("lattice output (host): ") + hipGetErrorString(e));
    b->lo_host_words = want;
  }
  KD_HIP_TRY(hipMemcpyAsync(b->lat_state_off.p, so.data(), b->batch * 8, hipMemcpyHostToDevice, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->lat_arc_off.p, ao.data(), b->batch * 8, hipMemcpyHostToDevice, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->lat_final_off.p, fo.data(), b->batch * 8, hipMemcpyHostToDevice, b->stream));
  LatticeParams p = b->LParams();
  p.use_final_probs = b->lat_use_final;
  KD_HIP_TRY(LaunchLatticeFill(p, b->batch, b->stream));
  KD_HIP_TRY(hipMemcpyAsync(b->lo_host, b->lo_blob.p, words * 4, hipMemcpyDeviceToHost, b->stream));
  KD_HIP_TRY(hipStreamSynchronize(b->stream));
  const int32_t *hw = b->lo_host;
  const int32_t *hsrc = hw, *hdst = hw + na, *hil = hw + 2 * na, *hol = hw + 3 * na;
  const int32_t *hgc = hw + 4 * na, *hac = hw + 5 * na;
  const int32_t *hsf = hw + 6 * na, *hsg = hsf + ns, *hfs = hsg + ns, *hfg = hfs + nfn;
  for (int i = 0; i < b->batch; ++i) {
    const size_t s1 = b->h_lat_sizes[4 * i], a1 = b->h_lat_sizes[4 * i + 1],
                 f1 = b->h_lat_sizes[4 * i + 2];
    auto C4 = [&](void *d, const void *s2, size_t n) { if (d && n) std::memcpy(d, s2, n * 4); };
    C4(src ? src + arc_off[i] : nullptr, hsrc + ao[i], a1);
    C4(dst ? dst + arc_off[i] : nullptr, hdst + ao[i], a1);
    C4(ilabel ? ilabel + arc_off[i] : nullptr, hil + ao[i], a1);
    C4(olabel ? olabel + arc_off[i] : nullptr, hol + ao[i], a1);
    C4(graph_cost ? graph_cost + arc_off[i] : nullptr, hgc + ao[i], a1);
    C4(ac_cost ? ac_cost + arc_off[i] : nullptr, hac + ao[i], a1);
    C4(state_frame ? state_frame + state_off[i] : nullptr, hsf + so[i], s1);
    C4(state_graph ? state_graph + state_off[i] : nullptr, hsg + so[i], s1);
    C4(final_state ? final_state + final_off[i] : nullptr, hfs + fo[i], f1);
    C4(final_graph ? final_graph + final_off[i] : nullptr, hfg + fo[i], f1);
  }
  return KD_OK;
}

int kd_batch_stats(kd_batch *b, int64_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  static_assert(sizeof(long long) == sizeof(int64_t), "");
  std::vector<long long> h(static_cast<size_t>(b->batch) * kStatStride);
  KD_HIP_TRY(hipMemcpy(h.data(), b->stats.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  for (int i = 0; i < b->batch; ++i)
    for (int k = 0; k < KD_NUM_STATS; ++k) out[i * KD_NUM_STATS + k] = h[i * kStatStride + k];
  return KD_OK;
}

int kd_batch_phase_cycles(kd_batch *b, int64_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
#ifndef KD_PROFILE
  return Fail(KD_E_STATE, "kd_batch_phase_cycles: not a diagnostic (KD_PROFILE) build");
#else
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  std::vector<long long> h(static_cast<size_t>(b->batch) * kStatStride);
  KD_HIP_TRY(hipMemcpy(h.data(), b->stats.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  for (int i = 0; i < b->batch; ++i)
    for (int k = 0; k < kProfSlots; ++k) out[i * kProfSlots + k] = h[i * kStatStride + KD_NUM_STATS + k];
  return KD_OK;
#endif
}

int kd_batch_last_kernel_ms(kd_batch *b, float *ms) {
  if (!b || !ms) return Fail(KD_E_INVALID, "bad argument");
  if (!b->timed) return Fail(KD_E_STATE, "no kernel launched yet");
  KD_HIP_TRY(hipEventSynchronize(b->ev1));
  KD_HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
  return KD_OK;
}

size_t kd_batch_device_bytes(const kd_batch *b) { return b ? b->DeviceBytes() : 0; }

}  // extern "C"
