// kaldi-decoder_amd/csrc/kd_api.hip -- host side of the C ABI in include/kd_hip.h:
// graph upload, batch state in HBM, kernel launches, result read-back.
// No search arithmetic happens here; if there is no usable GPU every entry point
// that would compute fails with KD_E_HIP (there is no CPU fallback).

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kd_device.h"
#include "kd_graph_check.h"
#include "kd_label_index.h"
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

// ---- two kernels side by side (round 6; kd_device.h SchedCtl) ----------
// The large and the small geometry's persistent workgroups must be resident TOGETHER and on
// CUs of their own: a large workgroup fills a CU's LDS, three small ones fill another's, and a
// CU that holds one small workgroup cannot take a large one.  Left to itself the dispatcher
// spreads the small kernel one workgroup per CU over the whole chip (tools/ubench/cumask.hip);
// with a CU mask per stream -- hipExtStreamCreateWithCUMask: the mask is a property of the
// hardware queue -- the two kernels land on disjoint CUs whatever the launch order.  Mask bits
// are dealt round-robin over the XCDs (bit i: XCD i % 8 -- measured, same tool), so "the first X
// bits" is X / 8 CUs of every XCD.  A few pairs of masked streams per device are created once
// and shared by the batches (round-robin): a masked stream is a hardware queue of its own, and
// batches that are in flight together should not share one.
struct SplitStreams {
  hipStream_t large = nullptr, small = nullptr;
};
struct SplitDevice {
  int large_cus = 0;
  bool failed = false;
  std::vector<SplitStreams> pairs;
  unsigned next = 0;
};
std::mutex g_split_mu;
std::map<int, SplitDevice> g_split;

int EnvInt(const char *name, int def) {
  const char *v = std::getenv(name);
  return v && *v ? std::atoi(v) : def;
}

// A pair of streams masked to the first `large_cus` CU bits / the rest (false: not available).
bool TakeSplitStreams(int device, int cus, int large_cus, SplitStreams *out) {
  std::lock_guard<std::mutex> lock(g_split_mu);
  SplitDevice &d = g_split[device];
  if (d.failed) return false;
  if (d.pairs.empty()) {
    d.large_cus = large_cus;
    const int n_pairs = std::max(1, std::min(8, EnvInt("KD_SPLIT_PAIRS", 4)));
    const size_t words = static_cast<size_t>(cus + 31) / 32;
    std::vector<uint32_t> ml(words, 0u), ms(words, 0u);
    for (int i = 0; i < cus; ++i) (i < large_cus ? ml : ms)[i / 32] |= 1u << (i % 32);
    for (int k = 0; k < n_pairs; ++k) {
      SplitStreams sp;
      if (hipExtStreamCreateWithCUMask(&sp.large, static_cast<uint32_t>(words), ml.data()) != hipSuccess ||
          hipExtStreamCreateWithCUMask(&sp.small, static_cast<uint32_t>(words), ms.data()) != hipSuccess) {
        (void)hipGetLastError();
        if (sp.large) (void)hipStreamDestroy(sp.large);
        if (d.pairs.empty()) { d.failed = true; return false; }
        break;
      }
      d.pairs.push_back(sp);
    }
  }
  if (d.large_cus != large_cus) return false;  // (one partition per device and process)
  *out = d.pairs[d.next++ % d.pairs.size()];
  return true;
}

}  // namespace

struct kd_graph {
  int device = 0;
  int32_t num_states = 0, start = -1;
  int64_t num_arcs = 0;
  int32_t max_ilabel = 0;  // a log-likelihood matrix needs at least this many columns
  DevBuf<StateRec> states;
  DevBuf<ArcPacked> arcs;
  DevBuf<int32_t> olabels;  // [arcs], in the arcs' device order (emitting first per state)
  DevBuf<uint32_t> lab_idx; // [arcs], label index of the emitting ranges (kd_device.h)
  DevBuf<int32_t> big_ilabel;  // [arcs] when the graph holds an input label of 2^20 - 1 or more (else empty)
  float emit_w_min = 0.0f;  // lowest emitting-arc weight (NaN if a weight is NaN)
};

struct kd_batch {
  const kd_graph *g = nullptr;
  kd_faster_options o{};
  int32_t max_batch = 0, batch = 0;
  hipStream_t stream = nullptr;
  int g_log2_slots = 17;
  int workgroups = 1;  // persistent workgroups of a launch == owners of the scratch regions
  // ---- pools (shared by the utterances; grown on demand, see GrowPools)
  uint64_t tok_cap = 0, link_cap = 0, out_cap = 0;
  // records an utterance takes from a pool at a time (fixed when the batch is created:
  // freed chunks are reused as chunks of exactly this size)
  int32_t tok_chunk = 8192, link_chunk = 16384;
  DevBuf<int32_t> tok_state, tok_prev, tok_arc;
  DevBuf<double> tok_cost;
  DevBuf<float> tok_extra;
  DevBuf<int32_t> tok_newid;
  DevBuf<uint32_t> lat_scratch;
  DevBuf<int32_t> link_src, link_dst, link_arc;
  DevBuf<float> link_ac, link_graph, link_delta;
  unsigned long long *out_pool = nullptr;  // pinned host memory, written by the kernel
  // ... and, if the caller has lent a DEVICE buffer for the exported lattices
  // (kd_batch_set_output_buffer), there too, word for word (the staged export)
  unsigned long long *out_ext = nullptr;
  uint64_t out_ext_cap = 0;
  std::vector<uint8_t> h_lat_arena;  // the export of utterance i is in arena order (general export)
  bool canon_done = true;
  bool mirrored = true;    // arena-order exports have been copied into the lent buffer (out_ext)
  // ---- per utterance
  DevBuf<long long> frame_base, link_emit_base, link_eps_base;
  DevBuf<int32_t> frame_count, link_emit_count, link_eps_count, frame_state_off, frame_arc_off;
  DevBuf<int32_t> frames_decoded;
  DevBuf<uint32_t> status;
  DevBuf<UttMem> mem;
  DevBuf<long long> stats;
  DevBuf<float> lat_fbc, lat_frel;
  DevBuf<int32_t> lat_any_final, lat_finalized;
  // ---- per workgroup: spill table, candidate queue, live list, worklists
  DevBuf<uint32_t> g_slot, g_used, g_wl, g_log_h, g_log_arc, g_log_src;
  DevBuf<unsigned long long> g_log_nk;
  DevBuf<float> g_log_ac, g_log_gr;
  // ---- the two kinds of workgroups (kd_device.h SchedCtl): queues, the frame loop's state of an
  // utterance that changes hands, the CU-masked streams the two kernels run on
  bool split_possible = false;    // not SimpleDecoder, a device of at least 16 CUs
  bool split_ok = false;          // the batch is prepared for split launches (PrepareSplit) ...
  bool split_on = false;          // ... and does (kd_batch_set_split)
  int split_min_batch = 0;        // ... and does for calls of at least this many utterances
  int cus = 0, large_cus = 0;
  int wgs_single = 1, wgs_large = 0, wgs_small = 0;  // workgroups of an unsplit launch / of a split one, at most
  int mig_lo = 0, mig_hi = 0, prune_small = 0, stay = 0;
  SplitStreams split_streams;
  hipEvent_t ev_fork = nullptr, ev_join_l = nullptr, ev_join_s = nullptr;
  DevBuf<SchedCtl> sched_ctl;
  DevBuf<SchedProf> sched_prof;
  DevBuf<uint32_t> sched_ring;
  DevBuf<UttResume> resume;
  int ring_cap = 1;
  // ---- launch control: UttCtl[B] | order[B] | UttOut[B] | HostTops in pinned host memory
  // the GPU reads and writes in place; BatchCtl (the atomic counters) in device memory
  DevBuf<BatchCtl> bctl;
  unsigned char *ctl_host = nullptr;
  size_t off_order = 0, off_out = 0, off_tops = 0, ctl_bytes = 0;
  UttCtl *h_ctl() const { return reinterpret_cast<UttCtl *>(ctl_host); }
  int32_t *h_order() const { return reinterpret_cast<int32_t *>(ctl_host + off_order); }
  UttOut *h_out() const { return reinterpret_cast<UttOut *>(ctl_host + off_out); }
  HostTops *h_tops() const { return reinterpret_cast<HostTops *>(ctl_host + off_tops); }
  // BatchCtl's counters as of the last read-back (raw: they only ever grow), and where
  // each pool's index 0 currently is; cursor = raw - base
  uint32_t work_raw = 0, done_raw = 0;
  uint64_t tok_raw = 0, link_raw = 0, out_raw = 0, tok_base = 0, link_base = 0, out_base = 0;
  uint64_t h_tok_top = 0, h_link_top = 0, h_out_top = 0;
  // the launch in flight (what a relaunch after growing a pool repeats)
  int32_t l_num_cols = 0;
  int64_t l_row_stride = 0;
  uint32_t l_tail_flags = 0;
  // lattice state
  bool lattice = false, finalized = false, simple = false, log_softmax = false;
  bool lat_stats_valid = false;  // h_final_rel is current
  bool lazy_prune = true;        // PruneActiveTokens owed, not run, until it shows (kd_batch_set_lazy_prune)
  uint64_t gc_tok_budget = 1ull << 29, gc_link_budget = 1ull << 30;  // pools of this size collect, not grow
  int32_t exported_use_final = -1;  // the output pool holds GetRawLattice(use_final_probs = this)
  kd_lattice_options lo{};
  std::vector<float> h_final_rel;
  std::vector<int32_t> h_lat_sizes;
  std::vector<long long> h_lat_off;
  DevBuf<float> staging;
  // host mirrors
  std::vector<int32_t> h_decoded, h_path_len, h_reached;
  std::vector<uint32_t> h_status;
  std::vector<float> h_final_weight, h_path_final_rel;
  bool initialized = false, path_valid = false, pending = false;
  int32_t path_use_final = -1;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_staged = nullptr;  // behind the last host-to-device copy of a staged call
  bool timed = false;
  int grow_events = 0;

  ~kd_batch() {
    if (out_pool) (void)hipHostFree(out_pool);
    if (ctl_host) (void)hipHostFree(ctl_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_staged) (void)hipEventDestroy(ev_staged);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join_l) (void)hipEventDestroy(ev_join_l);
    if (ev_join_s) (void)hipEventDestroy(ev_join_s);
  }

  size_t DeviceBytes() const {
    return tok_state.bytes() + tok_prev.bytes() + tok_arc.bytes() + tok_cost.bytes() +
           tok_extra.bytes() + tok_newid.bytes() + lat_scratch.bytes() + link_src.bytes() +
           link_dst.bytes() + link_arc.bytes() + link_ac.bytes() + link_graph.bytes() +
           link_delta.bytes() + frame_base.bytes() + frame_count.bytes() +
           link_emit_base.bytes() + link_eps_base.bytes() + link_emit_count.bytes() +
           link_eps_count.bytes() + frame_state_off.bytes() + frame_arc_off.bytes() +
           g_slot.bytes() + g_log_h.bytes() + g_log_arc.bytes() +
           g_log_src.bytes() + g_log_nk.bytes() + g_log_ac.bytes() + g_log_gr.bytes() +
           g_used.bytes() + g_wl.bytes() + staging.bytes() + stats.bytes() +
           bctl.bytes();
  }

  SearchParams Params() const {
    SearchParams p{};
    p.states = g->states.p;
    p.arcs = g->arcs.p;
    p.arc_olabel = g->olabels.p;
    p.lab_idx = g->lab_idx.p;
    p.big_ilabel = g->big_ilabel.n ? g->big_ilabel.p : nullptr;
    p.emit_w_min = g->emit_w_min;
    p.start = g->start;
    p.beam = o.beam;
    p.max_active = o.max_active;
    p.min_active = o.min_active;
    p.beam_delta = o.beam_delta;
    p.tok_state = tok_state.p;
    p.tok_cost = tok_cost.p;
    p.tok_prev = tok_prev.p;
    p.tok_arc = tok_arc.p;
    p.tok_cap = tok_cap;
    // a chunk is what an utterance takes from a pool at a time: large enough that the
    // atomic is rare, small enough that what is left over when the utterance ends is
    // nothing against the pool
    p.tok_chunk = tok_chunk;
    p.frame_base = frame_base.p;
    p.frame_count = frame_count.p;
    p.max_frames = o.max_frames;
    p.frames_decoded = frames_decoded.p;
    p.status = status.p;
    p.mem = mem.p;
    p.stats = stats.p;
    p.g_slot = g_slot.p;
    p.g_used = g_used.p;
    p.g_wl = g_wl.p;
    p.g_log_h = g_log_h.p;
    p.g_log_arc = g_log_arc.p;
    p.g_log_src = g_log_src.p;
    p.g_log_nk = g_log_nk.p;
    p.g_log2_slots = g_log2_slots;
    p.batch = batch;
    p.utt = h_ctl();     // (pinned host memory: the same address on the device)
    p.order = h_order();
    p.ctl = bctl.p;
    p.out = h_out();
    p.tops = h_tops();
    p.work_base = work_raw;
    p.done_base = done_raw;
    p.tok_base = tok_base;
    p.link_base = link_base;
    p.out_base = out_base;
    p.lattice = lattice ? 1 : 0;
    p.simple = simple ? 1 : 0;
    p.log_softmax = log_softmax ? 1 : 0;
    p.g_log_ac = g_log_ac.p;
    p.g_log_gr = g_log_gr.p;
    p.tok_extra = tok_extra.p;
    p.link_src = link_src.p;
    p.link_dst = link_dst.p;
    p.link_arc = link_arc.p;
    p.link_ac = link_ac.p;
    p.link_graph = link_graph.p;
    p.link_delta = link_delta.p;
    p.link_cap = link_cap;
    p.link_chunk = link_chunk;
    p.link_emit_base = link_emit_base.p;
    p.link_emit_count = link_emit_count.p;
    p.link_eps_base = link_eps_base.p;
    p.link_eps_count = link_eps_count.p;
    p.lat.lattice_beam = lo.lattice_beam;
    p.lat.prune_interval = lattice && lo.prune_interval > 0 ? lo.prune_interval : 0x7FFFFFFF;
    p.lat.prune_scale = lo.prune_scale;
    p.lat.prune_lazy = lazy_prune ? 1 : 0;
    // Owed prunes are a pool's garbage collection only when it is not to grow: no_grow, or the
    // batch's budget reached (KD_GC_TOKENS / KD_GC_LINKS records; default 2^29 / 2^30: 12.9 GB of
    // token and 25.8 GB of link records -- a 1024-utterance call of the large-LM workload un-pruned).
    p.lat.gc_tokens = (o.no_grow || tok_cap >= gc_tok_budget) ? 1 : 0;
    p.lat.gc_links = (o.no_grow || link_cap >= gc_link_budget) ? 1 : 0;
    p.lat.tok_extra = tok_extra.p;
    p.lat.tok_newid = tok_newid.p;
    p.lat.scratch = lat_scratch.p;
    p.lat.final_best_cost = lat_fbc.p;
    p.lat.final_rel = lat_frel.p;
    p.lat.any_final = lat_any_final.p;
    p.lat.finalized = lat_finalized.p;
    p.lat.frame_state_off = frame_state_off.p;
    p.lat.frame_arc_off = frame_arc_off.p;
    p.lat.out = out_pool;
    p.lat.out2 = out_ext;
    p.lat.out_cap = out_ext ? std::min<uint64_t>(out_cap, out_ext_cap) : out_cap;
    p.sched.ctl = sched_ctl.p;
    p.sched.prof = sched_prof.p;
    p.sched.ring = sched_ring.p;
    p.sched.resume = resume.p;
    p.sched.ring_cap = ring_cap;
    p.sched.wg_base = 0;
    p.sched.total_wgs = 0;  // (Enqueue)
    p.sched.mig_lo = mig_lo;
    p.sched.mig_hi = mig_hi;
    p.sched.prune_small = prune_small;
    p.sched.stay = stay;
    p.sched.split = 0;
    return p;
  }
};

namespace {

constexpr uint32_t kRecoverable =
    KD_UTT_ARENA_FULL | KD_UTT_LINKS_FULL | KD_UTT_TABLE_FULL | KD_UTT_OUT_FULL | KD_UTT_PATH_QUEUE_FULL;
constexpr uint64_t kIndexLimit = (1ull << 31) - 1;  // tokens and links are indexed with int32
constexpr int kMaxLog2Slots = 19;  // the spill table's worklist bitmask lives in 64 KiB of LDS
// A call is split over the two kinds of workgroups when it brings at least as many utterances
// as the chip has CUs: below that every utterance has a CU of its own anyway.  (Test builds:
// every call, so that fuzz-sized cases change hands.)
#ifndef KD_SPLIT_MIN_BATCH
#define KD_SPLIT_MIN_BATCH_DEFAULT(cus) (cus)
#else
#define KD_SPLIT_MIN_BATCH_DEFAULT(cus) (KD_SPLIT_MIN_BATCH)
#endif
#ifndef KD_SPLIT_DEFAULT
#define KD_SPLIT_DEFAULT 0  // (1 in the test builds: every batch is split unless told otherwise)
#endif

int CheckOptions(const kd_faster_options &o) {
  // faster-decoder.cc:24-28
  if (!(o.hash_ratio >= 1.0f)) return Fail(KD_E_INVALID, "Check failed: config_.hash_ratio >= 1.0");
  if (!(o.max_active > 1)) return Fail(KD_E_INVALID, "Check failed: config_.max_active > 1");
  if (!(o.min_active >= 0 && o.min_active < o.max_active))
    return Fail(KD_E_INVALID, "Check failed: config_.min_active >= 0 && config_.min_active < config_.max_active");
  return KD_OK;
}

// ---- capacity: everything below grows on demand ---------------------------------
// A pool, the spill tables or the frame index that turn out too small do not fail the
// decode (the reference keeps going until the host runs out of memory): the utterances
// that could not commit a frame stop at that frame boundary with a KD_UTT_* bit, the
// host doubles what was exhausted and launches again; the search resumes from the
// arena.  `no_grow` restores the hard limits (KD_E_CAPACITY).

template <class T>
hipError_t GrowBuf(DevBuf<T> &b, size_t new_count, size_t keep, hipStream_t st) {
  if (!b.p) return hipSuccess;  // (not allocated in this mode)
  T *np = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&np), std::max<size_t>(new_count, 1) * sizeof(T));
  if (e != hipSuccess) return e;
  if (keep) e = hipMemcpyAsync(np, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(np); return e; }
  (void)hipFree(b.p);
  b.p = np;
  b.n = new_count;
  return hipSuccess;
}

int GrowTokens(kd_batch *b) {
  if (b->tok_cap >= kIndexLimit)
    return Fail(KD_E_CAPACITY, "token pool full at 2^31 - 1 records (32-bit token indices)");
  const uint64_t cap = std::min<uint64_t>(kIndexLimit, b->tok_cap * 2);
  const size_t keep = static_cast<size_t>(std::min<uint64_t>(b->h_tok_top, b->tok_cap));
  hipError_t e = GrowBuf(b->tok_state, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->tok_cost, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->tok_prev, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->tok_arc, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->tok_extra, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->tok_newid, cap, 0, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->lat_scratch, cap, 0, b->stream);
  if (e != hipSuccess)
    return Fail(KD_E_CAPACITY, std::string("token pool full and cannot grow: ") + hipGetErrorString(e));
  b->tok_cap = cap;
  b->grow_events++;
  return KD_OK;
}

int GrowLinks(kd_batch *b) {
  if (b->link_cap >= kIndexLimit)
    return Fail(KD_E_CAPACITY, "link pool full at 2^31 - 1 links (32-bit link indices)");
  const uint64_t cap = std::min<uint64_t>(kIndexLimit, b->link_cap * 2);
  const size_t keep = static_cast<size_t>(std::min<uint64_t>(b->h_link_top, b->link_cap));
  hipError_t e = GrowBuf(b->link_src, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->link_dst, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->link_arc, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->link_ac, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->link_graph, cap, keep, b->stream);
  if (e == hipSuccess) e = GrowBuf(b->link_delta, cap, keep, b->stream);
  if (e != hipSuccess)
    return Fail(KD_E_CAPACITY, std::string("link pool full and cannot grow: ") + hipGetErrorString(e));
  b->link_cap = cap;
  b->grow_events++;
  return KD_OK;
}

constexpr uint64_t kMaxOutWords = 1ull << 33;  // 64 GiB of pinned host memory
int GrowOut(kd_batch *b) {
  if (b->out_ext && b->out_ext_cap <= b->out_cap)
    return Fail(KD_E_CAPACITY, "the output buffer lent with kd_batch_set_output_buffer is full (" +
                               std::to_string(b->out_ext_cap) + " words): lend a larger one");
  if (b->out_cap >= kMaxOutWords)
    return Fail(KD_E_CAPACITY, "lattice output pool full at 2^33 words");
  const uint64_t cap = b->out_cap * 2;
  // (what the pool holds is exported again after growing: nothing to keep)
  unsigned long long *np = nullptr;
  if (hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&np), cap * 8, hipHostMallocDefault); e != hipSuccess)
    return Fail(KD_E_CAPACITY, std::string("lattice output pool cannot grow: ") + hipGetErrorString(e));
  (void)hipHostFree(b->out_pool);
  b->out_pool = np;
  b->out_cap = cap;
  b->grow_events++;
  return KD_OK;
}

hipError_t AllocTables(kd_batch *b, int lg) {
  // The new set is allocated before the old one is let go: if an allocation fails mid-way
  // the batch keeps its old tables (and g_log2_slots), and stays usable at its old size.
  const size_t gs = static_cast<size_t>(b->workgroups) << lg;
  DevBuf<uint32_t> slot, used, wl, log_h, log_arc, log_src;
  DevBuf<unsigned long long> log_nk;
  DevBuf<float> log_ac, log_gr;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(slot.Alloc(4 * gs));
  A(used.Alloc(gs));
  A(wl.Alloc(2 * gs));
  A(log_h.Alloc(2 * gs));
  A(log_arc.Alloc(2 * gs));
  A(log_src.Alloc(2 * gs));
  A(log_nk.Alloc(2 * gs));
  if (b->lattice || b->simple) A(log_ac.Alloc(2 * gs));
  if (b->lattice) A(log_gr.Alloc(2 * gs));
  if (e == hipSuccess) e = hipMemsetAsync(slot.p, 0xFF, slot.bytes(), b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e != hipSuccess) return e;  // (the locals free what was allocated)
  auto S = [](auto &dst, auto &src) { std::swap(dst.p, src.p); std::swap(dst.n, src.n); };
  S(b->g_slot, slot); S(b->g_used, used); S(b->g_wl, wl);
  S(b->g_log_h, log_h); S(b->g_log_arc, log_arc); S(b->g_log_src, log_src); S(b->g_log_nk, log_nk);
  S(b->g_log_ac, log_ac); S(b->g_log_gr, log_gr);
  b->g_log2_slots = lg;
  return hipSuccess;
}

int GrowTables(kd_batch *b, bool for_path = false) {
  if (b->g_log2_slots >= kMaxLog2Slots)
    return Fail(KD_E_CAPACITY, for_path
                    ? "a best path of more than 2^20 arcs does not fit the staging queue at its "
                      "largest (2 x 2^19 entries): decode the utterance in pieces"
                    : "token table full at 2^19 slots per utterance: a frame holds more "
                      "than 393216 tokens (tighten beam / max_active)");
  // the tables are empty between frames: nothing to carry over
  if (hipError_t e = AllocTables(b, b->g_log2_slots + 1); e != hipSuccess)
    return Fail(KD_E_CAPACITY, std::string("token table full and cannot grow: ") + hipGetErrorString(e));
  b->o.table_slots = 1 << b->g_log2_slots;
  b->grow_events++;
  return KD_OK;
}

// Per-utterance, per-frame index arrays: rows of (max_frames + 1) entries.
template <class T>
hipError_t GrowRows(DevBuf<T> &buf, int rows, int old_w, int new_w, hipStream_t st) {
  if (!buf.p) return hipSuccess;
  T *np = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&np), static_cast<size_t>(rows) * new_w * sizeof(T));
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(np, 0, static_cast<size_t>(rows) * new_w * sizeof(T), st);
  if (e == hipSuccess)
    e = hipMemcpy2DAsync(np, new_w * sizeof(T), buf.p, old_w * sizeof(T), old_w * sizeof(T), rows,
                         hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(np); return e; }
  (void)hipFree(buf.p);
  buf.p = np;
  buf.n = static_cast<size_t>(rows) * new_w;
  return hipSuccess;
}

int GrowFrames(kd_batch *b, int32_t need) {
  if (b->o.no_grow) return Fail(KD_E_CAPACITY, "frame index full (raise max_frames)");
  int32_t nm = b->o.max_frames;
  while (nm < need) nm = nm < (1 << 29) ? nm * 2 : need;
  const int ow = b->o.max_frames + 1, nw = nm + 1, rows = b->max_batch;
  hipError_t e = GrowRows(b->frame_base, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->frame_count, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->link_emit_base, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->link_emit_count, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->link_eps_base, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->link_eps_count, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->frame_state_off, rows, ow, nw, b->stream);
  if (e == hipSuccess) e = GrowRows(b->frame_arc_off, rows, ow, nw, b->stream);
  if (e != hipSuccess)
    return Fail(KD_E_CAPACITY, std::string("frame index full and cannot grow: ") + hipGetErrorString(e));
  b->o.max_frames = nm;
  b->grow_events++;
  return KD_OK;
}

// ---- launch -----------------------------------------------------------------------
// The utterances' UttCtl entries (ll, num_rows, offset, target, flags) are in the pinned
// control block.  A launch is ONE command on the stream -- the kernel: it reads its
// inputs from, and leaves its results (per-utterance outcomes, exported lattices, the
// pools' cursors) in, pinned host memory, so there is nothing to copy either way.
int Enqueue(kd_batch *b) {
  // work queue order: most frames to decode first, so the launch does not end on a
  // long utterance started late
  int32_t *order = b->h_order();
  for (int i = 0; i < b->batch; ++i) order[i] = i;
  const UttCtl *uc = b->h_ctl();
  const std::vector<int32_t> &dec = b->h_decoded;
  std::stable_sort(order, order + b->batch, [&](int32_t x, int32_t y) {
    const int fx = uc[x].target - ((uc[x].flags & kUttInit) ? 0 : std::max(dec[x], 0));
    const int fy = uc[y].target - ((uc[y].flags & kUttInit) ? 0 : std::max(dec[y], 0));
    return fx > fy;
  });
  SearchParams p = b->Params();
  p.num_cols = b->l_num_cols;
  p.row_stride = b->l_row_stride;
  if (b->log_softmax && b->l_num_cols > SearchMaxLdsCols())
    return Fail(KD_E_INVALID, "log-softmax is fused into the staging of a row in LDS: at most " +
                              std::to_string(SearchMaxLdsCols()) + " columns");
  if (b->log_softmax && b->split_ok && b->l_num_cols > SearchMaxLdsColsSmall())
    return Fail(KD_E_INVALID, "log-softmax is fused into the staging of a row in LDS: at most " +
                              std::to_string(SearchMaxLdsColsSmall()) + " columns");
  int wgs = 0;
  KD_HIP_TRY(hipEventRecord(b->ev0, b->stream));
  if (b->split_ok && b->split_on && b->batch >= b->split_min_batch) {
    // Both kernels, each on its CU-masked stream, between a fork and a join on the batch's own
    // stream.  Either kernel can finish the call alone (a large workgroup runs any frame; a
    // small one hands a large frame over only to large workgroups that have started), so the
    // order in which the hardware starts them decides nothing but speed.
    const int wl = std::min(b->batch, b->wgs_large), ws = std::min(b->batch, b->wgs_small);
    wgs = wl + ws;
    p.sched.split = 1;
    p.sched.total_wgs = wgs;
    KD_HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
    KD_HIP_TRY(hipStreamWaitEvent(b->split_streams.large, b->ev_fork, 0));
    KD_HIP_TRY(hipStreamWaitEvent(b->split_streams.small, b->ev_fork, 0));
    p.sched.wg_base = 0;
    KD_HIP_TRY(LaunchSearch(p, wl, b->split_streams.large));
    p.sched.wg_base = wl;
    KD_HIP_TRY(LaunchSearchSmall(p, ws, b->split_streams.small));
    KD_HIP_TRY(hipEventRecord(b->ev_join_l, b->split_streams.large));
    KD_HIP_TRY(hipEventRecord(b->ev_join_s, b->split_streams.small));
    KD_HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join_l, 0));
    KD_HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join_s, 0));
  } else {
    wgs = std::min(b->batch, b->wgs_single);
    p.sched.total_wgs = wgs;
    KD_HIP_TRY(LaunchSearch(p, wgs, b->stream));
  }
  KD_HIP_TRY(hipEventRecord(b->ev1, b->stream));
  // every utterance is handed out once and every workgroup makes one failed attempt
  b->work_raw += static_cast<uint32_t>(b->batch + wgs);
  b->done_raw += static_cast<uint32_t>(wgs);
  b->timed = true;
  b->pending = true;
  b->path_valid = false;
  return KD_OK;
}

std::string StatusMessage(const kd_batch *b, int i) {
  std::string m = "utterance " + std::to_string(i) + " stopped at frame " +
                  std::to_string(b->h_decoded[i]) + ":";
  const uint32_t st = b->h_status[i];
  if (st & KD_UTT_ARENA_FULL) m += " token pool full (raise arena_tokens)";
  if (st & KD_UTT_LINKS_FULL) m += " link pool full (raise arena_links)";
  if (st & KD_UTT_TABLE_FULL) m += " token table full (raise table_slots)";
  if (st & KD_UTT_FRAMES_FULL) m += " frame index full (raise max_frames)";
  if (st & KD_UTT_OUT_FULL) m += " lattice output pool full";
  if (st & KD_UTT_PATH_QUEUE_FULL)
    m += " best path longer than the staging queue of 2 x table_slots arcs (raise table_slots)";
  return m;
}

// An exported lattice that the kernel left in arena order (the general export: an
// un-finalized lattice, or one that outgrew the sweep's lists): states inside a frame and
// arcs come in an order that depends on wave scheduling.  Here -- in the pinned output pool,
// which is host memory -- it is put into the order the staged export produces on the device:
// states by (frame, graph state), arcs by (source state id, graph arc), finals by state id;
// the arcs shrink from four words to three.  Same input, same arrays (ref: GetRawLattice,
// lattice-simple-decoder.cc:584-657, is deterministic too).
void CanonicalizeExport(unsigned long long *region, int32_t *sizes) {
  const size_t ns = static_cast<size_t>(sizes[0]), na = static_cast<size_t>(sizes[1]), nf = static_cast<size_t>(sizes[2]);
  if (ns == 0) return;
  unsigned long long *o_state = region, *o_final = region + ns, *o_arc = o_final + nf;
  std::vector<int32_t> order(ns), newid(ns);
  for (size_t i = 0; i < ns; ++i) order[i] = static_cast<int32_t>(i);
  // a state word is {frame, graph state}: one token per state and frame, so the key is unique
  auto key = [&](int32_t i) {
    const unsigned long long w = o_state[i];
    return (w & 0xFFFFFFFFull) << 32 | w >> 32;
  };
  std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return key(x) < key(y); });
  std::vector<unsigned long long> tmp(ns);
  for (size_t k = 0; k < ns; ++k) { newid[order[k]] = static_cast<int32_t>(k); tmp[k] = o_state[order[k]]; }
  std::copy(tmp.begin(), tmp.end(), o_state);
  if (sizes[3] >= 0) sizes[3] = newid[sizes[3]];
  for (size_t k = 0; k < nf; ++k) {
    const uint32_t id = static_cast<uint32_t>(o_final[k]);
    o_final[k] = (o_final[k] & 0xFFFFFFFF00000000ull) | static_cast<uint32_t>(newid[id]);
  }
  std::sort(o_final, o_final + nf, [](unsigned long long x, unsigned long long y) {
    return static_cast<uint32_t>(x) < static_cast<uint32_t>(y);
  });
  struct Arc { unsigned long long ends, labels, costs; uint32_t arc; };
  std::vector<Arc> arcs(na);
  for (size_t k = 0; k < na; ++k) {
    const unsigned long long *w = o_arc + 4 * k;
    const uint32_t src = static_cast<uint32_t>(w[0]), dst = static_cast<uint32_t>(w[0] >> 32);
    arcs[k] = Arc{static_cast<unsigned long long>(static_cast<uint32_t>(newid[dst])) << 32 |
                      static_cast<uint32_t>(newid[src]),
                  w[1], w[2], static_cast<uint32_t>(w[3])};
  }
  std::sort(arcs.begin(), arcs.end(), [](const Arc &x, const Arc &y) {
    const uint32_t sx = static_cast<uint32_t>(x.ends), sy = static_cast<uint32_t>(y.ends);
    return sx != sy ? sx < sy : x.arc < y.arc;
  });
  for (size_t k = 0; k < na; ++k) {
    o_arc[3 * k] = arcs[k].ends;
    o_arc[3 * k + 1] = arcs[k].labels;
    o_arc[3 * k + 2] = arcs[k].costs;
  }
}

// The exported results as the host reads them: arena-ordered exports are put in order, once.
int HostResults(kd_batch *b) {
  if (!b->canon_done) {
    for (int i = 0; i < b->batch; ++i)
      if (b->h_lat_arena[i]) CanonicalizeExport(b->out_pool + b->h_lat_off[i], &b->h_lat_sizes[4 * i]);
    b->canon_done = true;
  }
  return KD_OK;
}

// Waits for the launch in flight and takes its results; grows whatever a stopped
// utterance ran out of and launches again until every utterance is where it was sent.
int ReadBack(kd_batch *b) {
  while (true) {
    KD_HIP_TRY(hipStreamSynchronize(b->stream));
    b->pending = false;
    const HostTops tops = *b->h_tops();
    const UttOut *uo = b->h_out();
    b->tok_raw = tops.tok_top;
    b->link_raw = tops.link_top;
    b->out_raw = tops.out_top;
    b->h_tok_top = b->tok_raw - b->tok_base;
    b->h_link_top = b->link_raw - b->link_base;
    b->h_out_top = b->out_raw - b->out_base;
    uint32_t any = 0;
    int first_bad = -1;
    for (int i = 0; i < b->batch; ++i) {
      b->h_decoded[i] = uo[i].frames_decoded;
      b->h_status[i] = uo[i].status;
      if (uo[i].status && first_bad < 0) first_bad = i;
      any |= uo[i].status;
    }
    if (b->lattice && (b->l_tail_flags & (kUttFinalStats | kUttFinalize | kUttExport | kUttExportNoFinal)) && !any) {
      for (int i = 0; i < b->batch; ++i) b->h_final_rel[i] = uo[i].final_rel;
      b->lat_stats_valid = true;
      if (b->l_tail_flags & kUttFinalize) b->finalized = true;
      if (b->l_tail_flags & (kUttExport | kUttExportNoFinal)) {
        for (int i = 0; i < b->batch; ++i) {
          for (int k = 0; k < 4; ++k) b->h_lat_sizes[4 * i + k] = uo[i].sizes[k];
          b->h_lat_off[i] = uo[i].out_off;
          b->h_lat_arena[i] = (uo[i].phase & kPhaseArenaOrder) && uo[i].out_off >= 0;
        }
        b->canon_done = false;
        b->mirrored = false;
        b->exported_use_final = (b->l_tail_flags & kUttExport) ? 1 : 0;
      }
    }
    if (!b->lattice && (b->l_tail_flags & kUttBestPath) && !any) {
      for (int i = 0; i < b->batch; ++i) {
        b->h_path_len[i] = uo[i].sizes[0];
        b->h_reached[i] = uo[i].sizes[1];
        std::memcpy(&b->h_final_weight[2 * i], &uo[i].sizes[2], 4);
        b->h_final_weight[2 * i + 1] = 0.0f;
        b->h_path_final_rel[i] = uo[i].final_rel;
        b->h_lat_off[i] = uo[i].out_off;
      }
      b->path_valid = true;
      b->path_use_final = (b->l_tail_flags & kUttPathNoFinal) ? 0 : 1;
    }
    if (!any) return KD_OK;
    if (b->o.no_grow || (any & ~kRecoverable)) return Fail(KD_E_CAPACITY, StatusMessage(b, first_bad));
    // A pool's cursor only moves forward (kd_search.hip, PoolChunk): the take that did not fit,
    // and every take after it, left it BEYOND the capacity.  Nothing was handed out past the
    // first failed take, which started below the capacity: pull the cursor back to the
    // capacity before the launch resumes, or the overshoot -- one chunk per workgroup that ran
    // into the end together -- is lost for the rest of the decode and may force a doubling more.
    auto pull_back = [&](uint64_t *raw, uint64_t base, uint64_t cap, size_t word) -> int {
      if (*raw - base <= cap) return KD_OK;
      const unsigned long long v = base + cap;
      // (on the batch's own stream, which is idle here: the launch that overshot has been waited for)
      KD_HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(b->bctl.p) + word, &v, sizeof(v), hipMemcpyHostToDevice,
                                b->stream));
      KD_HIP_TRY(hipStreamSynchronize(b->stream));
      *raw = v;
      return KD_OK;
    };
    if (int r = pull_back(&b->tok_raw, b->tok_base, b->tok_cap, offsetof(BatchCtl, tok_top))) return r;
    if (int r = pull_back(&b->link_raw, b->link_base, b->link_cap, offsetof(BatchCtl, link_top))) return r;
    b->h_tok_top = b->tok_raw - b->tok_base;
    b->h_link_top = b->link_raw - b->link_base;
    // (a pool that cannot grow -- the device is out of memory, or the index limit -- is
    // garbage-collected from here on if prunes may be owed: Params() tells the kernel so; a pool
    // that could not grow AND is already collected is the end)
    auto grow_or_collect = [&](int (*grow)(kd_batch *), uint64_t *budget, uint64_t cap) -> int {
      const int r = grow(b);
      if (r == KD_OK) return KD_OK;
      if (!b->lattice || !b->lazy_prune || b->o.no_grow || cap >= *budget) return r;
      *budget = cap;  // (gc_tokens / gc_links from the next launch on)
      return KD_OK;
    };
    if (any & KD_UTT_ARENA_FULL)
      if (int r = grow_or_collect(GrowTokens, &b->gc_tok_budget, b->tok_cap)) return r;
    if (any & KD_UTT_LINKS_FULL)
      if (int r = grow_or_collect(GrowLinks, &b->gc_link_budget, b->link_cap)) return r;
    if (any & KD_UTT_TABLE_FULL) {
      if (int r = GrowTables(b)) return r;
    } else if (any & KD_UTT_PATH_QUEUE_FULL) {
      if (int r = GrowTables(b, true)) return r;
    }
    if (any & KD_UTT_OUT_FULL) {
      if (int r = GrowOut(b)) return r;
    }
    // every utterance's tail runs again in the launch that resumes the stopped ones, and exports
    // again: from the start of the output pool (grown or not), not behind what this launch
    // exported -- a call that is resumed three times must not need four times the pool (round 6:
    // a lent buffer sized for one export of the call was "full" after two resumptions)
    b->out_base = b->out_raw;
    b->h_out_top = 0;
    // resume: utterances that never got past InitDecoding start over, the others go on
    // from their last committed frame
    UttCtl *uc = b->h_ctl();
    for (int i = 0; i < b->batch; ++i)
      if (b->h_decoded[i] >= 0) uc[i].flags &= ~kUttInit;
    if (int r = Enqueue(b)) return r;
  }
}

// Validates the per-utterance frame windows and fills the UttCtl entries (ll pointers
// are set by the caller).  faster-decoder.cc:126-144.
int PlanAdvance(kd_batch *b, const int32_t *num_rows, const int32_t *offset,
                int32_t max_num_frames, bool with_init) {
  if (!with_init && !b->initialized)
    return Fail(KD_E_STATE, "Check failed: num_frames_decoded_ >= 0 && "
                            "\"You must call InitDecoding() before AdvanceDecoding()\"");
  if (!with_init && b->finalized)
    return Fail(KD_E_STATE, "FinalizeDecoding() has been called: cannot decode further");
  int32_t need = 0;
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < b->batch; ++i) {
    const int32_t done = with_init ? 0 : b->h_decoded[i];
    const int32_t off = offset ? offset[i] : 0;
    const int32_t ready = off + num_rows[i];
    if (!(ready >= done))
      return Fail(KD_E_INVALID, "Check failed: num_frames_ready >= num_frames_decoded_");
    int32_t target = ready;
    if (max_num_frames >= 0) target = std::min(target, done + max_num_frames);
    if (target > done && done < off)
      return Fail(KD_E_INVALID, "decodable offset is beyond the next frame to decode");
    need = std::max(need, target);
    uc[i].num_rows = num_rows[i];
    uc[i].offset = off;
    uc[i].target = target;
    uc[i].flags = with_init ? kUttInit : 0u;
  }
  if (need > b->o.max_frames)
    if (int r = GrowFrames(b, need)) return r;
  return KD_OK;
}

// A launch that decodes nothing: only the lattice tail (`tail`) of every utterance.
int LaunchTail(kd_batch *b, uint32_t tail) {
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < b->batch; ++i) {
    uc[i].ll = nullptr;
    uc[i].num_rows = 0;
    uc[i].offset = 0;
    uc[i].target = b->h_decoded[i];
    uc[i].flags = tail;
  }
  b->l_num_cols = 1;
  b->l_row_stride = 1;
  b->l_tail_flags = tail;
  if (int r = Enqueue(b)) return r;
  return ReadBack(b);
}

void ResetForInit(kd_batch *b, int32_t batch) {
  b->batch = batch;
  b->finalized = false;
  b->lat_stats_valid = false;
  b->exported_use_final = -1;
  // the pools are empty again: index 0 is wherever their counters stand
  b->tok_base = b->tok_raw;
  b->link_base = b->link_raw;
  b->out_base = b->out_raw;
  b->h_tok_top = 0;
  b->h_link_top = 0;
  b->h_out_top = 0;
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
  if (row_ptr[0] != 0) return Fail(KD_E_INVALID, "row_ptr[0] != 0");
  const int64_t na = row_ptr[num_states];
  if (na > 0 && (!ilabel || !olabel || !weight || !nextstate)) return Fail(KD_E_INVALID, "arc arrays are NULL");
  if (!final_cost) return Fail(KD_E_INVALID, "final_cost is NULL");
  std::vector<uint8_t> has_eps(num_states, 0);
  int32_t max_ilabel = 0;
  for (int32_t s = 0; s < num_states; ++s) {
    if (row_ptr[s + 1] < row_ptr[s]) return Fail(KD_E_INVALID, "row_ptr not monotone");
    for (int32_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
      if (nextstate[a] < 0 || nextstate[a] >= num_states)
        return Fail(KD_E_INVALID, "arc nextstate out of range");
      if (ilabel[a] < 0) return Fail(KD_E_INVALID, "negative ilabel");
      if (ilabel[a] == 0) has_eps[s] = 1;
      max_ilabel = std::max(max_ilabel, ilabel[a]);
    }
  }
  {
    // epsilon cycles: positive ones decode like the reference; a zero- or negative-weight one
    // is refused here (kd_graph_check.h says why)
    std::string why;
    if (!EpsilonCyclesArePositive(num_states, row_ptr, ilabel, weight, nextstate, &why))
      return Fail(KD_E_INVALID, why);
  }
  // (input labels of 2^20 - 1 and more do not fit the arc record's 20-bit field: it holds the
  // escape code and the label stands in a side array -- kd_device.h kIlabelEscape)
  const bool big_labels = max_ilabel >= kIlabelEscape;
  std::vector<int32_t> big(big_labels ? std::max<int64_t>(na, 1) : 0, 0);
  std::vector<StateRec> st(num_states);
  std::vector<ArcPacked> ar(std::max<int64_t>(na, 1));
  std::vector<int32_t> ol(std::max<int64_t>(na, 1), 0);
  for (int32_t s = 0; s < num_states; ++s) {
    const int32_t b = row_ptr[s], e = row_ptr[s + 1];
    int32_t pe = b;
    for (int32_t a = b; a < e; ++a)
      if (ilabel[a] != 0) ++pe;
    st[s] = StateRec{b, pe, e, final_cost[s]};
  }
  for (int32_t s = 0; s < num_states; ++s) {
    const int32_t b = row_ptr[s], e = row_ptr[s + 1];
    int32_t we = b, wn = st[s].emit_end;
    for (int32_t a = b; a < e; ++a) {
      // the arc carries its destination's arc range (first arc, emitting and epsilon degrees:
      // what the search would otherwise fetch from the destination's state record)
      const StateRec &d = st[nextstate[a]];
      const uint32_t degs = PackDegs(d.emit_end - d.arc_begin, d.arc_end - d.emit_end);
      const uint32_t lab20 = static_cast<uint32_t>(ilabel[a] >= kIlabelEscape ? kIlabelEscape : ilabel[a]);
      ArcPacked r{lab20 | degs << kIlabelBits, static_cast<uint32_t>(d.arc_begin),
                  weight[a], static_cast<uint32_t>(nextstate[a]) | (has_eps[nextstate[a]] ? kHasEps : 0u)};
      const int32_t at = ilabel[a] != 0 ? we++ : wn++;
      ar[at] = r;
      ol[at] = olabel[a];
      if (big_labels) big[at] = ilabel[a];
    }
  }
  // Label index (kd_label_index.h): per state, the keys (ilabel, offset) of its emitting arcs as an
  // implicit 5-ary search tree, in an array parallel to the arcs -- the arcs themselves keep their
  // order.  A token whose cost leaves room for the frame's likeliest label only looks THAT label
  // up (kd_search.hip, IndexedArcs) instead of reading every arc: the loop of
  // faster-decoder.cc:204-211 / lattice-simple-decoder.cc:375-399 visits arcs whose cost is known
  // to be beyond the cutoff from a lower bound (this frame's second-lowest acoustic cost, the
  // graph's lowest emitting weight).  (+ 4 entries: a node read at the end of the array is 16 bytes.)
  std::vector<uint32_t> li(static_cast<size_t>(std::max<int64_t>(na, 1)) + 4, kIdxNone);
  float w_min = std::numeric_limits<float>::infinity();
  bool w_nan = false;
  {
    std::vector<uint32_t> labs, tmp;
    for (int32_t s = 0; s < num_states; ++s) {
      const int32_t b = st[s].arc_begin, e = st[s].emit_end;
      for (int32_t a = b; a < e; ++a) {
        const float w = ar[a].weight;
        if (w != w) w_nan = true;
        else if (w < w_min) w_min = w;
      }
      if (e - b >= kIdxMaxDeg) continue;
      labs.resize(e - b);
      tmp.resize(e - b);
      for (int32_t a = b; a < e; ++a) labs[a - b] = ar[a].lab & static_cast<uint32_t>(kMaxIlabel);
      LabelIndexBuild(labs.data(), e - b, li.data() + b, tmp.data());
    }
  }
  KD_HIP_TRY(hipSetDevice(device));
  kd_graph *g = new (std::nothrow) kd_graph;
  if (!g) return Fail(KD_E_INVALID, "out of host memory");
  g->emit_w_min = w_nan ? std::numeric_limits<float>::quiet_NaN() : w_min;
  g->device = device;
  g->num_states = num_states;
  g->start = start;
  g->num_arcs = na;
  g->max_ilabel = max_ilabel;
  hipError_t e = g->states.Alloc(num_states);
  if (e == hipSuccess) e = g->arcs.Alloc(ar.size());
  if (e == hipSuccess) e = g->olabels.Alloc(ol.size());
  if (e == hipSuccess) e = g->lab_idx.Alloc(li.size());
  if (e == hipSuccess && big_labels) e = g->big_ilabel.Alloc(big.size());
  if (e == hipSuccess && big_labels)
    e = hipMemcpy(g->big_ilabel.p, big.data(), big.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->lab_idx.p, li.data(), li.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->states.p, st.data(), st.size() * sizeof(StateRec), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->arcs.p, ar.data(), ar.size() * sizeof(ArcPacked), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(g->olabels.p, ol.data(), ol.size() * sizeof(int32_t), hipMemcpyHostToDevice);
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
size_t kd_graph_device_bytes(const kd_graph *g) {
  return g ? g->states.bytes() + g->arcs.bytes() + g->olabels.bytes() + g->lab_idx.bytes() +
                 (g->big_ilabel.p ? g->big_ilabel.bytes() : 0)
           : 0;
}

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
  o->no_grow = 0;
}

// Two kinds of workgroups (kd_device.h SchedCtl): most CUs run one large workgroup each, the
// rest three small ones each, and an utterance is run by the kind its frames' sizes ask for.
// A batch is prepared for it -- CU-masked streams, queues, scratch sets for both kernels'
// workgroups -- when it is first asked to (kd_batch_set_split, KD_SPLIT=1): calls are NOT dealt
// out that way by default.  Measured on the bench workload a split call that runs alone is
// 10-12 % shorter, but two calls kept in flight (how the throughput figure is reached: one call's
// tail is the other's middle) are 30 % slower split than not -- DESIGN section 5.  SimpleDecoder
// has no LDS-resident tier and is never split.  Returns KD_OK with split_ok false when the device
// gives no CU-masked streams.
static int PrepareSplit(kd_batch *b) {
  if (b->split_ok || !b->split_possible) return KD_OK;
  const int cus = b->cus, max_batch = b->max_batch;
  int x = EnvInt("KD_SPLIT_LARGE_CUS", cus * 7 / 8) / 8 * 8;
  x = std::max(8, std::min(cus - 8, x));
  if (!TakeSplitStreams(b->g->device, cus, x, &b->split_streams)) return KD_OK;
  const int wl = std::min<int>(max_batch, x);
  // (a few more than fit: the dispatcher does not deal them out evenly -- tools/ubench/cumask.hip
  // -- and one that finds no room starts when the launch is over, and leaves)
  const int ws = std::min<int>(max_batch, (cus - x) * SearchWorkgroupsPerCuSmall() + (cus - x) / 8);
  if (b->pending) (void)ReadBack(b);
  if (wl + ws > b->workgroups) {  // (the tables are empty between launches: nothing to carry over)
    const int was = b->workgroups;
    b->workgroups = wl + ws;
    if (hipError_t e = AllocTables(b, b->g_log2_slots); e != hipSuccess) {
      b->workgroups = was;
      return Fail(KD_E_CAPACITY, std::string("scratch sets for the second kernel's workgroups: ") + hipGetErrorString(e));
    }
  }
  while (b->ring_cap < max_batch) b->ring_cap *= 2;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(b->sched_ctl.Alloc(1));
  A(b->sched_prof.Alloc(1));
  A(b->sched_ring.Alloc(static_cast<size_t>(kMaxXcd) * 2 * b->ring_cap));
  A(b->resume.Alloc(max_batch));
  if (e == hipSuccess) e = hipMemset(b->sched_prof.p, 0, sizeof(SchedProf));
  if (e == hipSuccess) e = hipMemset(b->sched_ctl.p, 0, sizeof(SchedCtl));
  if (e == hipSuccess) e = hipMemset(b->sched_ring.p, 0, b->sched_ring.bytes());
  if (e == hipSuccess) e = hipMemset(b->resume.p, 0, b->resume.bytes());
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_join_l, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_join_s, hipEventDisableTiming);
  if (e != hipSuccess) return Fail(KD_E_HIP, std::string("preparing the second kernel: ") + hipGetErrorString(e));
  b->large_cus = x;
  b->wgs_large = wl;
  b->wgs_small = ws;
  b->split_min_batch = std::max(1, EnvInt("KD_SPLIT_MIN_BATCH", KD_SPLIT_MIN_BATCH_DEFAULT(cus)));
  b->mig_hi = std::max(1, std::min(EnvInt("KD_MIG_HI", std::min(64, SearchT1FrontSmall())), 1 << 20));
  b->mig_lo = std::max(0, std::min(EnvInt("KD_MIG_LO", std::min(32, b->mig_hi)), b->mig_hi));
  b->prune_small = EnvInt("KD_PRUNE_SMALL", 0) != 0 ? 1 : 0;
  b->stay = EnvInt("KD_SPLIT_STAY", 0) != 0 ? 1 : 0;
  b->split_ok = true;
  return KD_OK;
}

static int CreateBatch(const kd_graph *g, const kd_faster_options *opts,
                       const kd_lattice_options *lopts, bool simple, int32_t max_batch,
                       void *stream, kd_batch **out) {
  KD_HIP_TRY(hipSetDevice(g->device));
  kd_batch *b = new (std::nothrow) kd_batch;
  if (!b) return Fail(KD_E_INVALID, "out of host memory");
  b->g = g;
  b->o = *opts;
  b->max_batch = max_batch;
  b->stream = static_cast<hipStream_t>(stream);
  b->lattice = lopts != nullptr;
  b->simple = simple;
  if (lopts) b->lo = *lopts;
  // one persistent workgroup per CU takes utterances from the work queue
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g->device) != hipSuccess || cus <= 0)
    cus = 256;
  b->cus = cus;
  b->wgs_single = std::min<int>(max_batch, cus * SearchWorkgroupsPerCu());
  b->workgroups = b->wgs_single;
  b->split_possible = !simple && cus >= 16;
  b->lazy_prune = EnvInt("KD_PRUNE_LAZY", 1) != 0;
  if (const char *v = std::getenv("KD_GC_TOKENS")) b->gc_tok_budget = std::strtoull(v, nullptr, 10);
  if (const char *v = std::getenv("KD_GC_LINKS")) b->gc_link_budget = std::strtoull(v, nullptr, 10);
  // pools: 2^17 token records per utterance to begin with (an HLG decode at beam 15 writes
  // about 10^5 per 10 s utterance), two links per token; both grow on demand
  uint64_t tok = b->o.arena_tokens > 0 ? static_cast<uint64_t>(b->o.arena_tokens)
                                        : std::max<uint64_t>(1ull << 20, static_cast<uint64_t>(max_batch) << 17);
  b->tok_cap = std::min<uint64_t>(tok, kIndexLimit);
  b->o.arena_tokens = static_cast<int64_t>(b->tok_cap);
  b->tok_chunk = static_cast<int32_t>(std::min<uint64_t>(8192, std::max<uint64_t>(64, b->tok_cap / (8ull * b->workgroups))));
  if (b->o.max_frames <= 0) b->o.max_frames = 2048;
  int slots = b->o.table_slots <= 0 ? (1 << 17) : b->o.table_slots;
  int lg = 14;
  while ((1 << lg) < slots && lg < kMaxLog2Slots) ++lg;
  b->o.table_slots = 1 << lg;
  const size_t nf = static_cast<size_t>(max_batch) * (b->o.max_frames + 1);
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(b->tok_state.Alloc(b->tok_cap));
  A(b->tok_cost.Alloc(b->tok_cap));
  if (!b->lattice) {  // (lattice tokens keep forward links, not back-pointers)
    A(b->tok_prev.Alloc(b->tok_cap));
    A(b->tok_arc.Alloc(b->tok_cap));
  }
  if (b->lattice || b->simple) A(b->tok_extra.Alloc(b->tok_cap));
  A(b->frame_base.Alloc(nf));
  A(b->frame_count.Alloc(nf));
  A(b->frames_decoded.Alloc(max_batch));
  A(b->status.Alloc(max_batch));
  A(b->mem.Alloc(max_batch));
  A(b->stats.Alloc(static_cast<size_t>(max_batch) * kStatStride));
  A(AllocTables(b, lg));
  // launch control: the pinned block the GPU addresses directly, and the device counters
  b->off_order = static_cast<size_t>(max_batch) * sizeof(UttCtl);
  b->off_out = (b->off_order + static_cast<size_t>(max_batch) * sizeof(int32_t) + 15) & ~size_t(15);
  b->off_tops = (b->off_out + static_cast<size_t>(max_batch) * sizeof(UttOut) + 15) & ~size_t(15);
  b->ctl_bytes = b->off_tops + sizeof(HostTops);
  A(b->bctl.Alloc(1));
  if (e == hipSuccess) e = hipMemset(b->bctl.p, 0, sizeof(BatchCtl));
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void **>(&b->ctl_host), b->ctl_bytes, hipHostMallocDefault);
  if (e == hipSuccess) std::memset(b->ctl_host, 0, b->ctl_bytes);
  if (e == hipSuccess) e = hipMemset(b->frames_decoded.p, 0xFF, b->frames_decoded.bytes());
  if (e == hipSuccess) e = hipMemset(b->status.p, 0, b->status.bytes());
  if (e == hipSuccess) e = hipMemset(b->stats.p, 0, b->stats.bytes());
  if (e == hipSuccess) e = hipMemset(b->mem.p, 0, b->mem.bytes());
  // output pool (pinned): exported lattices, or best paths (three words an arc)
  b->out_cap = std::max<uint64_t>(1ull << 16, static_cast<uint64_t>(max_batch) << (lopts ? 13 : 11));
  A(hipHostMalloc(reinterpret_cast<void **>(&b->out_pool), b->out_cap * 8, hipHostMallocDefault));
  b->h_lat_off.assign(max_batch, -1);
  b->h_lat_arena.assign(max_batch, 0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev0);
  if (e == hipSuccess) e = hipEventCreate(&b->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_staged, hipEventDisableTiming);
  if (e == hipSuccess && lopts) {
    uint64_t links = lopts->arena_links > 0 ? static_cast<uint64_t>(lopts->arena_links) : 2 * b->tok_cap;
    b->link_cap = std::min<uint64_t>(links, kIndexLimit);
    b->lo.arena_links = static_cast<int64_t>(b->link_cap);
    b->link_chunk = static_cast<int32_t>(std::min<uint64_t>(16384, std::max<uint64_t>(64, b->link_cap / (8ull * b->workgroups))));
    A(b->tok_newid.Alloc(b->tok_cap));
    A(b->lat_scratch.Alloc(b->tok_cap));
    A(b->link_src.Alloc(b->link_cap));
    A(b->link_dst.Alloc(b->link_cap));
    A(b->link_arc.Alloc(b->link_cap));
    A(b->link_ac.Alloc(b->link_cap));
    A(b->link_graph.Alloc(b->link_cap));
    A(b->link_delta.Alloc(b->link_cap));
    A(b->link_emit_base.Alloc(nf));
    A(b->link_emit_count.Alloc(nf));
    A(b->link_eps_base.Alloc(nf));
    A(b->link_eps_count.Alloc(nf));
    A(b->frame_state_off.Alloc(nf));
    A(b->frame_arc_off.Alloc(nf));
    A(b->lat_fbc.Alloc(max_batch));
    A(b->lat_frel.Alloc(max_batch));
    A(b->lat_any_final.Alloc(max_batch));
    A(b->lat_finalized.Alloc(max_batch));
    if (e == hipSuccess) e = hipMemset(b->link_emit_count.p, 0, b->link_emit_count.bytes());
    if (e == hipSuccess) e = hipMemset(b->link_eps_count.p, 0, b->link_eps_count.bytes());
    if (e == hipSuccess) e = hipMemset(b->lat_finalized.p, 0, b->lat_finalized.bytes());
    b->h_lat_sizes.assign(4 * static_cast<size_t>(max_batch), 0);
  }
  if (e != hipSuccess) {
    delete b;
    return Fail(KD_E_HIP, std::string("batch allocation: ") + hipGetErrorString(e));
  }
  if (std::getenv("KD_DEBUG_ALLOC")) {  // (diagnostic: which array does a faulting address belong to?)
    auto D = [&](const char *name, const void *p, size_t bytes) {
      std::fprintf(stderr, "KD_ALLOC %-16s %p .. %p (%zu B)\n", name, p,
                   static_cast<const void *>(static_cast<const char *>(p) + bytes), bytes);
    };
#define KD_D(x) D(#x, b->x.p, b->x.bytes())
    KD_D(tok_state); KD_D(tok_prev); KD_D(tok_arc); KD_D(tok_cost); KD_D(tok_extra); KD_D(tok_newid);
    KD_D(lat_scratch); KD_D(link_src); KD_D(link_dst); KD_D(link_arc); KD_D(link_ac); KD_D(link_graph);
    KD_D(link_delta); KD_D(frame_base); KD_D(link_emit_base); KD_D(link_eps_base); KD_D(frame_count);
    KD_D(link_emit_count); KD_D(link_eps_count); KD_D(frame_state_off); KD_D(frame_arc_off);
    KD_D(frames_decoded); KD_D(status); KD_D(mem); KD_D(stats); KD_D(lat_fbc); KD_D(lat_frel);
    KD_D(lat_any_final); KD_D(lat_finalized); KD_D(g_slot); KD_D(g_used); KD_D(g_wl);
    KD_D(g_log_h); KD_D(g_log_arc); KD_D(g_log_src); KD_D(g_log_nk); KD_D(g_log_ac);
    KD_D(g_log_gr); KD_D(bctl);
#undef KD_D
    D("out_pool", b->out_pool, b->out_cap * 8);
    D("ctl_host", b->ctl_host, b->ctl_bytes);
    D("graph.states", g->states.p, g->states.bytes());
    D("graph.arcs", g->arcs.p, g->arcs.bytes());
  }
  b->h_decoded.assign(max_batch, -1);
  b->h_status.assign(max_batch, 0);
  b->h_path_len.assign(max_batch, -1);
  b->h_reached.assign(max_batch, 0);
  b->h_final_weight.assign(2 * static_cast<size_t>(max_batch), 0.f);
  b->h_path_final_rel.assign(max_batch, 0.f);
  b->h_final_rel.assign(max_batch, 0.f);
  // (KD_SPLIT=1: every batch deals its calls out over the two kinds of workgroups, as if
  // kd_batch_set_split(b, 1, ...) had been called; the test builds do by default)
  if (b->split_possible && EnvInt("KD_SPLIT", KD_SPLIT_DEFAULT) > 0) {
    if (int r = PrepareSplit(b)) {
      delete b;
      return r;
    }
    b->split_on = b->split_ok;
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
  return CreateBatch(g, opts, nullptr, false, max_batch, stream, out);
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
  return CreateBatch(g, &fo, nullptr, true, max_batch, stream, out);
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
  o->no_grow = 0;
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
  fo.no_grow = lo->no_grow;
  return CreateBatch(g, &fo, lo, false, max_batch, stream, out);
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
  if (b->pending) (void)ReadBack(b);  // (results of a launch nobody waited for are dropped)
  ResetForInit(b, batch);
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < batch; ++i) uc[i] = UttCtl{nullptr, 0, 0, 0, kUttInit};
  b->l_num_cols = 1;
  b->l_row_stride = 1;
  b->l_tail_flags = 0;
  if (int r = Enqueue(b)) return r;
  if (int r = ReadBack(b)) return r;
  b->initialized = true;
  return KD_OK;
}

static uint32_t TailFlags(const kd_batch *b, int32_t want) {
  uint32_t t = 0;
  if (!b->lattice) return (want & KD_DECODE_BEST_PATH) ? kUttBestPath : 0u;
  if (want & KD_DECODE_FINALIZE) t |= kUttFinalize;
  if (want & KD_DECODE_LATTICE) t |= kUttExport;
  return t;
}

// Host matrices: only the rows this call will read are staged (packed to num_cols), and
// the utterances' ll pointers are set so that row (frame - offset) of the caller's matrix
// is row (frame - first decoded frame) of the staged block.  `done` = frames decoded so far.
static int StageHost(kd_batch *b, const float *const *loglikes, int32_t num_cols, int64_t row_stride,
                     const int32_t *done) {
  UttCtl *uc = b->h_ctl();
  size_t total = 0;
  std::vector<size_t> first(b->batch), count(b->batch);
  for (int i = 0; i < b->batch; ++i) {
    const int32_t f0 = done ? std::max(done[i], 0) : 0, f1 = uc[i].target;
    first[i] = f1 > f0 ? static_cast<size_t>(f0 - uc[i].offset) : 0;
    count[i] = f1 > f0 ? static_cast<size_t>(f1 - f0) : 0;
    total += count[i] * num_cols;
  }
  if (total > 0 && num_cols < b->g->max_ilabel)
    return Fail(KD_E_INVALID, "log-likelihood matrix has " + std::to_string(num_cols) +
                              " columns but the graph's largest ilabel is " + std::to_string(b->g->max_ilabel));
  KD_HIP_TRY(b->staging.Reserve(total));
  size_t pos = 0;
  for (int i = 0; i < b->batch; ++i) {
    if (count[i]) {
      KD_HIP_TRY(hipMemcpy2DAsync(b->staging.p + pos, num_cols * sizeof(float),
                                  loglikes[i] + first[i] * row_stride, row_stride * sizeof(float),
                                  num_cols * sizeof(float), count[i], hipMemcpyHostToDevice,
                                  b->stream));
    }
    uc[i].ll = b->staging.p + pos - first[i] * static_cast<size_t>(num_cols);
    pos += count[i] * num_cols;
  }
  // The contract of every staged call: when it returns, the rows have LEFT the caller's
  // memory (ref: DecodableCtc copies its matrix, csrc/decodable-ctc.h:15-16).  From pageable
  // memory HIP's copy blocks the host anyway; from pinned memory (hipHostMalloc,
  // hipHostRegister, torch pin_memory) it is truly asynchronous, and a caller that refills its
  // buffer for the next call would overwrite rows still on their way.  Wait for the last copy
  // -- not for the launch, which is enqueued behind it.
  // (only then -- a call fed pageable matrices returns as soon as HIP's own
  // staging lets it, as before round 4.)
  // (a handful of driver look-ups per call, not one per utterance -- the first
  // and last non-empty matrices and two between; callers hand over matrices of one kind)
  bool pinned = false;
  std::vector<int> rows;
  for (int i = 0; i < b->batch; ++i)
    if (count[i]) rows.push_back(i);
  const size_t nr = rows.size();
  for (size_t k = 0; k < std::min<size_t>(nr, 4) && !pinned; ++k) {
    const int i = rows[nr <= 4 ? k : k * (nr - 1) / 3];
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, loglikes[i]) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // (plain malloc'ed memory: "invalid value", not an error of ours)
  }
  if (total > 0 && pinned) {
    KD_HIP_TRY(hipEventRecord(b->ev_staged, b->stream));
    KD_HIP_TRY(hipEventSynchronize(b->ev_staged));
  }
  return KD_OK;
}

int kd_batch_decode_async(kd_batch *b, int32_t batch, const float *const *loglikes_dev,
                          const int32_t *num_rows, const int32_t *offset, int32_t num_cols,
                          int64_t row_stride, int32_t want) {
  if (!b || !loglikes_dev || !num_rows || num_cols <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (batch <= 0 || batch > b->max_batch) return Fail(KD_E_INVALID, "bad batch size");
  if (row_stride < num_cols) return Fail(KD_E_INVALID, "row_stride < num_cols");
  if (num_cols < b->g->max_ilabel)
    return Fail(KD_E_INVALID, "log-likelihood matrix has " + std::to_string(num_cols) +
                              " columns but the graph's largest ilabel is " + std::to_string(b->g->max_ilabel));
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending) (void)ReadBack(b);
  ResetForInit(b, batch);
  if (int r = PlanAdvance(b, num_rows, offset, -1, true)) return r;
  const uint32_t tail = TailFlags(b, want);
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < batch; ++i) {
    uc[i].ll = loglikes_dev[i];
    uc[i].flags |= tail;
  }
  b->l_num_cols = num_cols;
  b->l_row_stride = row_stride;
  b->l_tail_flags = tail;
  b->initialized = true;
  return Enqueue(b);
}

int kd_batch_decode_host_async(kd_batch *b, int32_t batch, const float *const *loglikes,
                               const int32_t *num_rows, const int32_t *offset, int32_t num_cols,
                               int64_t row_stride, int32_t want) {
  if (!b || !loglikes || !num_rows || num_cols <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (batch <= 0 || batch > b->max_batch) return Fail(KD_E_INVALID, "bad batch size");
  if (row_stride < num_cols) return Fail(KD_E_INVALID, "row_stride < num_cols");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending) (void)ReadBack(b);  // (the staging block is this batch's: its last launch is over)
  ResetForInit(b, batch);
  if (int r = PlanAdvance(b, num_rows, offset, -1, true)) return r;
  // the rows go to the device on the batch's stream, ahead of the launch: while they are
  // copied (the call returns when the last row has left the caller's memory) the launches
  // of OTHER batches keep the CUs busy
  if (int r = StageHost(b, loglikes, num_cols, row_stride, nullptr)) return r;
  const uint32_t tail = TailFlags(b, want);
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < batch; ++i) uc[i].flags |= tail;
  b->l_num_cols = num_cols;
  b->l_row_stride = num_cols;
  b->l_tail_flags = tail;
  b->initialized = true;
  return Enqueue(b);
}

int kd_batch_decode(kd_batch *b, int32_t batch, const float *const *loglikes, const int32_t *num_rows,
                    const int32_t *offset, int32_t num_cols, int64_t row_stride, int mem_kind,
                    int32_t want) {
  if (mem_kind == KD_MEM_DEVICE) {
    if (int r = kd_batch_decode_async(b, batch, loglikes, num_rows, offset, num_cols, row_stride, want))
      return r;
    return ReadBack(b);
  }
  if (mem_kind != KD_MEM_HOST) return Fail(KD_E_INVALID, "bad mem_kind");
  if (int r = kd_batch_decode_host_async(b, batch, loglikes, num_rows, offset, num_cols, row_stride, want))
    return r;
  return ReadBack(b);
}

int kd_batch_advance_async(kd_batch *b, const float *const *loglikes_dev, const int32_t *num_rows,
                           const int32_t *offset, int32_t num_cols, int64_t row_stride,
                           int32_t max_num_frames) {
  if (!b || !loglikes_dev || !num_rows || num_cols <= 0) return Fail(KD_E_INVALID, "bad argument");
  if (row_stride < num_cols) return Fail(KD_E_INVALID, "row_stride < num_cols");
  if (num_cols < b->g->max_ilabel)
    return Fail(KD_E_INVALID, "log-likelihood matrix has " + std::to_string(num_cols) +
                              " columns but the graph's largest ilabel is " + std::to_string(b->g->max_ilabel));
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (int r = PlanAdvance(b, num_rows, offset, max_num_frames, false)) return r;
  UttCtl *uc = b->h_ctl();
  for (int i = 0; i < b->batch; ++i) uc[i].ll = loglikes_dev[i];
  b->l_num_cols = num_cols;
  b->l_row_stride = row_stride;
  b->l_tail_flags = 0;
  b->lat_stats_valid = false;
  b->exported_use_final = -1;
  return Enqueue(b);
}

int kd_batch_sync(kd_batch *b) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (!b->pending) return KD_OK;
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
  if (int r = PlanAdvance(b, num_rows, offset, max_num_frames, false)) return r;
  UttCtl *uc = b->h_ctl();
  if (int r = StageHost(b, loglikes, num_cols, row_stride, b->h_decoded.data())) return r;
  b->l_num_cols = num_cols;
  b->l_row_stride = num_cols;
  b->l_tail_flags = 0;
  b->lat_stats_valid = false;
  b->exported_use_final = -1;
  if (int r = Enqueue(b)) return r;
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
  // (lattice tokens carry forward links, not back-pointers: the path tail is not theirs,
  // and rewinding the output pool here would orphan an exported lattice)
  if (b->lattice) return Fail(KD_E_STATE, "a lattice batch has no back-pointers: use kd_batch_raw_lattice_*");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (b->path_valid && b->path_use_final == use_final_probs) return KD_OK;
  // a launch in which every utterance only runs its tail: best token, traceback, arcs
  b->out_base = b->out_raw;  // the pool is rewritten from its start
  b->h_out_top = 0;
  return LaunchTail(b, kUttBestPath | (use_final_probs ? 0u : kUttPathNoFinal));
}

int kd_batch_reached_final(kd_batch *b, int32_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (b->lattice) {
    // lattice-simple-decoder.h:142-147: "FinalRelativeCost() serves the same purpose as
    // ReachedFinal()": infinity iff no final state is active.  A tail launch that only
    // computes the final costs; an exported lattice in the output pool stays valid.
    std::vector<float> rel(static_cast<size_t>(b->batch));
    if (int r = kd_batch_final_relative_cost(b, rel.data())) return r;
    for (int i = 0; i < b->batch; ++i) out[i] = std::isinf(rel[i]) ? 0 : 1;
    return KD_OK;
  }
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
  if (!b->path_valid || b->pending) return Fail(KD_E_STATE, "call kd_batch_best_path_prepare first");
  if (int r = HostResults(b)) return r;
  if (final_weight)
    std::memcpy(final_weight, b->h_final_weight.data(), 2 * b->batch * sizeof(float));
  // an utterance's region of the (pinned) output pool: {ilabel, olabel} {state, graph} {ac, -}
  for (int i = 0; i < b->batch; ++i) {
    const int32_t n = b->h_path_len[i];
    if (n <= 0 || b->h_lat_off[i] < 0) continue;
    const int32_t *w = reinterpret_cast<const int32_t *>(b->out_pool + b->h_lat_off[i]);
    const size_t d = static_cast<size_t>(path_offset[i]);
    for (int32_t k = 0; k < n; ++k) {
      const int32_t *a = w + 6 * k;
      if (ilabel) ilabel[d + k] = a[0];
      if (olabel) olabel[d + k] = a[1];
      if (state) state[d + k] = a[2];
      if (graph_cost) std::memcpy(graph_cost + d + k, a + 3, 4);
      if (ac_cost) std::memcpy(ac_cost + d + k, a + 4, 4);
    }
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

int kd_batch_finalize_decoding(kd_batch *b) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (b->finalized) return Fail(KD_E_STATE, "Check failed: !decoding_finalized_");
  b->exported_use_final = -1;
  return LaunchTail(b, kUttFinalize);
}

int kd_batch_final_relative_cost(kd_batch *b, float *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  if (!b->lattice) {  // SimpleDecoder::FinalRelativeCost, simple-decoder.cc:78-100
    if (int r = PreparePath(b, b->path_valid ? b->path_use_final : 1)) return r;
    for (int i = 0; i < b->batch; ++i) out[i] = b->h_path_final_rel[i];
    return KD_OK;
  }
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (!b->lat_stats_valid)
    if (int r = LaunchTail(b, kUttFinalStats)) return r;
  for (int i = 0; i < b->batch; ++i) out[i] = b->h_final_rel[i];
  return KD_OK;
}

int kd_batch_raw_lattice_prepare(kd_batch *b, int32_t use_final_probs, int32_t *sizes) {
  if (!b || !sizes) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  if (!b->initialized) return Fail(KD_E_STATE, "InitDecoding() has not been called");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (b->finalized && !use_final_probs)  // lattice-simple-decoder.cc:594-597
    return Fail(KD_E_STATE, "You cannot call FinalizeDecoding() and then call "
                            "GetRawLattice() with use_final_probs == false");
  for (int i = 0; i < b->batch; ++i)
    if (!(b->h_decoded[i] > 0)) return Fail(KD_E_STATE, "Check failed: num_frames > 0");
  const int want = use_final_probs ? 1 : 0;
  if (b->exported_use_final != want) {
    b->out_base = b->out_raw;  // the pool is rewritten from its start
    b->h_out_top = 0;
    if (int r = LaunchTail(b, use_final_probs ? kUttExport : kUttExportNoFinal)) return r;
  }
  if (int r = HostResults(b)) return r;  // (ordering an arena-order export moves its start state)
  std::memcpy(sizes, b->h_lat_sizes.data(), 4 * b->batch * sizeof(int32_t));
  return KD_OK;
}

int kd_batch_raw_lattice_copy(kd_batch *b, const int64_t *state_off, const int64_t *arc_off,
                              const int64_t *final_off, int32_t *src, int32_t *dst,
                              int32_t *ilabel, int32_t *olabel, float *graph_cost,
                              float *ac_cost, int32_t *state_frame, int32_t *state_graph,
                              int32_t *final_state, float *final_graph) {
  if (!b || !state_off || !arc_off || !final_off) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice || b->exported_use_final < 0 || b->pending)
    return Fail(KD_E_STATE, "call kd_batch_raw_lattice_prepare first");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (int r = HostResults(b)) return r;
  // an utterance's region: states {frame, graph state} | finals {state, weight} |
  // arcs {src, dst, ilabel, olabel, graph, ac}
  for (int i = 0; i < b->batch; ++i) {
    const size_t ns = b->h_lat_sizes[4 * i], na = b->h_lat_sizes[4 * i + 1], nf = b->h_lat_sizes[4 * i + 2];
    if (ns == 0 || b->h_lat_off[i] < 0) continue;
    const int32_t *w = reinterpret_cast<const int32_t *>(b->out_pool + b->h_lat_off[i]);
    for (size_t k = 0; k < ns; ++k) {
      if (state_frame) state_frame[state_off[i] + k] = w[2 * k];
      if (state_graph) state_graph[state_off[i] + k] = w[2 * k + 1];
    }
    w += 2 * ns;
    for (size_t k = 0; k < nf; ++k) {
      if (final_state) final_state[final_off[i] + k] = w[2 * k];
      if (final_graph) std::memcpy(final_graph + final_off[i] + k, w + 2 * k + 1, 4);
    }
    w += 2 * nf;
    for (size_t k = 0; k < na; ++k) {
      const int32_t *a = w + 6 * k;
      const size_t at = static_cast<size_t>(arc_off[i]) + k;
      if (src) src[at] = a[0];
      if (dst) dst[at] = a[1];
      if (ilabel) ilabel[at] = a[2];
      if (olabel) olabel[at] = a[3];
      if (graph_cost) std::memcpy(graph_cost + at, a + 4, 4);
      if (ac_cost) std::memcpy(ac_cost + at, a + 5, 4);
    }
  }
  return KD_OK;
}

int kd_batch_set_log_softmax(kd_batch *b, int32_t on) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  b->log_softmax = on != 0;
  return KD_OK;
}

int kd_batch_set_output_buffer(kd_batch *b, void *device_words, int64_t cap_words) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  if (device_words && cap_words <= 0) return Fail(KD_E_INVALID, "bad capacity");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  b->out_ext = static_cast<unsigned long long *>(device_words);
  b->out_ext_cap = device_words ? static_cast<uint64_t>(cap_words) : 0;
  b->exported_use_final = -1;  // (what was exported is not in the new buffer)
  b->mirrored = false;
  return KD_OK;
}

int kd_batch_raw_lattice_table(kd_batch *b, int64_t *used_words, int64_t *table) {
  if (!b || !used_words || !table) return Fail(KD_E_INVALID, "bad argument");
  if (!b->lattice) return Fail(KD_E_STATE, "not a lattice batch");
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending)
    if (int r = ReadBack(b)) return r;
  if (b->exported_use_final < 0)
    return Fail(KD_E_STATE, "nothing exported: decode with KD_DECODE_LATTICE or call kd_batch_raw_lattice_prepare");
  // An export in arena order (an un-finalized lattice, one exported apart from the decode, one
  // that outgrew the staging lists) is written to the pinned pool only and put in order by the
  // host: mirror those utterances' ordered regions into the lent buffer, so that buffer prefix +
  // table are complete whichever export ran.  (Rare; the staged export of a decode call writes
  // both copies itself.)
  // Once per export, on the batch's stream, waited for here: the lent buffer must not be read by
  // work of the caller's that is still in flight when this call is made (include/kd_hip.h).
  if (b->out_ext && !b->mirrored) {
    bool any = false;
    for (int i = 0; i < b->batch; ++i) any |= b->h_lat_arena[i] && b->h_lat_off[i] >= 0;
    if (any) {
      if (int r = HostResults(b)) return r;
      for (int i = 0; i < b->batch; ++i) {
        if (!b->h_lat_arena[i] || b->h_lat_off[i] < 0) continue;
        const int32_t *sz = &b->h_lat_sizes[4 * i];
        const size_t words = static_cast<size_t>(sz[0]) + sz[2] + 3 * static_cast<size_t>(sz[1]);
        KD_HIP_TRY(hipMemcpyAsync(b->out_ext + b->h_lat_off[i], b->out_pool + b->h_lat_off[i], words * 8,
                                  hipMemcpyHostToDevice, b->stream));
      }
      KD_HIP_TRY(hipStreamSynchronize(b->stream));
    }
    b->mirrored = true;
  }
  *used_words = static_cast<int64_t>(b->h_out_top);
  for (int i = 0; i < b->batch; ++i) {
    table[5 * i] = b->h_lat_off[i];
    for (int k = 0; k < 4; ++k) table[5 * i + 1 + k] = b->h_lat_sizes[4 * i + k];
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
#if !defined(KD_PROFILE) && !defined(KD_DEBUG_WAVE)
  return Fail(KD_E_STATE, "kd_batch_phase_cycles: not a diagnostic (KD_PROFILE) build");
#else
  KD_HIP_TRY(hipSetDevice(b->g->device));
  if (b->pending) (void)ReadBack(b);  // (a diagnostic read-out: also of a launch that failed)
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

int kd_batch_capacity(const kd_batch *b, int64_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  out[0] = static_cast<int64_t>(b->tok_cap);
  out[1] = static_cast<int64_t>(b->link_cap);
  out[2] = 1ll << b->g_log2_slots;
  out[3] = b->o.max_frames;
  // (a take that did not fit leaves the cursor beyond the capacity: kd_search.hip PoolChunk)
  out[4] = static_cast<int64_t>(std::min<uint64_t>(b->h_tok_top, b->tok_cap));
  out[5] = static_cast<int64_t>(std::min<uint64_t>(b->h_link_top, b->link_cap));
  out[6] = b->grow_events;
  out[7] = b->workgroups;
  return KD_OK;
}

int kd_batch_set_lazy_prune(kd_batch *b, int32_t lazy) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (b->pending) (void)ReadBack(b);
  b->lazy_prune = lazy != 0;
  return KD_OK;
}

int kd_batch_split_info(const kd_batch *b, int32_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  for (int i = 0; i < 8; ++i) out[i] = 0;
  if (!b->split_ok) return KD_OK;
  out[0] = b->split_on ? 1 : 0;
  out[1] = b->large_cus;
  out[2] = b->wgs_large;
  out[3] = b->wgs_small;
  out[4] = b->mig_lo;
  out[5] = b->mig_hi;
  out[6] = b->prune_small | (b->stay << 1);
  out[7] = b->split_min_batch;
  return KD_OK;
}

int kd_batch_sched_profile(kd_batch *b, int64_t *out) {
  if (!b || !out) return Fail(KD_E_INVALID, "bad argument");
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (!b->split_ok) return KD_OK;
  if (b->pending) {
    if (int r = ReadBack(b)) return r;
  }
  static_assert(sizeof(SchedProf) == 16 * sizeof(int64_t), "sixteen counters");
  KD_HIP_TRY(hipMemcpy(out, b->sched_prof.p, sizeof(SchedProf), hipMemcpyDeviceToHost));
  return KD_OK;
}

int kd_batch_set_split(kd_batch *b, int32_t enable, int32_t min_batch, int32_t mig_lo, int32_t mig_hi,
                       int32_t prune_small, int32_t stay) {
  if (!b) return Fail(KD_E_INVALID, "bad argument");
  if (b->pending) (void)ReadBack(b);
  if (!b->split_ok && enable > 0) {
    KD_HIP_TRY(hipSetDevice(b->g->device));
    if (int r = PrepareSplit(b)) return r;
  }
  if (!b->split_ok) {
    if (enable > 0) return Fail(KD_E_STATE, "this batch has no second kernel (SimpleDecoder, or CU-masked "
                                            "streams are not available)");
    return KD_OK;
  }
  const int hi = mig_hi >= 0 ? mig_hi : b->mig_hi, lo = mig_lo >= 0 ? mig_lo : std::min(b->mig_lo, hi);
  if (hi < 1 || lo > hi) return Fail(KD_E_INVALID, "need 0 <= mig_lo <= mig_hi, mig_hi >= 1");
  b->mig_hi = hi;
  b->mig_lo = lo;
  if (enable >= 0) b->split_on = enable != 0;
  if (min_batch >= 1) b->split_min_batch = min_batch;
  if (prune_small >= 0) b->prune_small = prune_small != 0 ? 1 : 0;
  if (stay >= 0) b->stay = stay != 0 ? 1 : 0;
  return KD_OK;
}

}  // extern "C"
