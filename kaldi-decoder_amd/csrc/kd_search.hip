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
// Mapping: one workgroup owns one utterance for the whole launch and loops
// over its frames; utterances never interact, so the only synchronisation is
// s_barrier inside the workgroup and a frame costs no kernel launch.  A frame:
//   cutoff    best cost comes from the previous frame's compaction; max_active
//             / min_active need one counting pass and, when they bind, a 4-pass
//             radix select.  The best token's arcs are scored first to bound
//             the next cutoff early (faster-decoder.cc:176-189).
//   filter    one token per lane.  Tokens with <= 4 arcs are scored by their
//             own lane; the rest are queued in LDS and taken kGroup at a time
//             by whole waves, so consecutive lanes read consecutive 16-byte
//             arcs.  The log-likelihood row sits in LDS.  Arcs that beat the
//             running cutoff are appended to a candidate queue (state, arc,
//             source, cost key) -- this phase never touches the token table.
//             A token whose cost leaves room for the row's likeliest label only
//             does not read its arcs: it finds the one of that label in the
//             label index (kd_label_index.h; exact: a lower bound of every
//             other arc's cost is at or beyond the running cutoff), and those
//             arcs' costs tighten the cutoff before anything is queued.
//   insert    one queue entry per lane (dense): entries below the now exact
//             cutoff do one ds_cmpst (claim) and one 64-bit ds_min (relax) on
//             the 8192-slot LDS token table; a state's first cost appends its
//             slot to the frame's live list.
//   epsilon   filter + insert again over epsilon arc ranges, iterated to a
//             fixpoint with a worklist of improved slots.
//   resolve   queue entries that still hold their slot's cost compete for the
//             lowest arc index (deterministic back-pointer), then the live
//             list becomes one contiguous arena block (state, cost, prev, arc)
//             and only the live slots are cleared.
// The table (key, 64-bit cost key, winning arc) is 128 KiB of the CU's 160 KiB
// LDS.  A frame whose states do not fit replays its queue into a larger table
// in global memory (template parameter LDS=false); the filter is not repeated.
//
// Semantics are the reference's, made order-free (DESIGN.md "Order
// dependence"): costs are double and use the reference's expressions, the
// emitting cutoff is the value the reference's running cutoff ends at
// (the running value only limits the queue), strict improvement replaces, and
// ties pick the lowest arc index.  The oracle's policy 1 states the same
// semantics on the CPU and the two are compared bit for bit.

#include <float.h>
#include <limits.h>

#include <atomic>
#include <type_traits>

#include "kd_device.h"
#include "kd_hip.h"
#include "kd_label_index.h"

// Timing-perturbation build (make jitter; -DKD_JITTER=<unit>): behind EVERY workgroup barrier
// -- __syncthreads, LdsBarrier, Team::Sync -- each wave sleeps a pseudo-random time of its own
// (seeded by the clock and the wave's id: another draw at every barrier of every call), mostly
// up to 31 units of <unit> x 64 cycles, one time in sixteen sixteen times that.  Results must
// not change: a kernel whose waves may only differ in WHEN they pass a barrier gives the same
// answer however late any of them is.  What it shakes out is an LDS region that one phase reads
// and the next overwrites with no barrier in between: in the shipped build the reader is
// rarely late enough to see it (round 3's SweepWindow race bit one bench run in six and was
// found by reading); here a wave lags by thousands of cycles at every barrier.  Never shipped,
// never timed: tools/fuzz_gpu.py, the bench-scale tests and tools/soak.sh run against it.
#ifdef KD_JITTER
__device__ __forceinline__ void KdJitter() {
  uint32_t x = static_cast<uint32_t>(__builtin_amdgcn_s_memtime()) * 2654435761u;
  x ^= (threadIdx.x >> 6) * 0x9E3779B9u + blockIdx.x * 0x85EBCA6Bu;
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  x = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x)));
  int n = static_cast<int>(x & 31u);
  if (((x >> 5) & 15u) == 0u) n *= 16;
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(KD_JITTER);
}
#define __syncthreads()  \
  do {                   \
    ::__syncthreads();   \
    KdJitter();          \
  } while (0)
#else
__device__ __forceinline__ void KdJitter() {}
#endif

// This file is compiled twice into the library (kaldi-decoder_amd/Makefile): as it stands -- the
// LARGE geometry: one 768-thread workgroup per CU with the whole CU's LDS, the geometry every
// frame fits -- and with -DKD_GEOM_SMALL: 256 threads, a third of the LDS, three workgroups to a
// CU, whose chains of dependent round trips overlap (round 5 measured 1.7-1.8 x on workloads of
// small frames, and a loss on frames that want the CU's LDS).  A call launches both kernels side
// by side on disjoint CUs and utterances change hands at frame boundaries (RunUtterance,
// SearchKernel; kd_device.h SchedCtl).  Everything below is in an anonymous namespace: the two
// copies share nothing but the launchers' names, which differ.
#ifdef KD_GEOM_SMALL
#define KD_GEOM_KIND 0
#ifndef KD_BLOCK
#define KD_BLOCK 256
#endif
#ifndef KD_LOG2_SLOTS
#define KD_LOG2_SLOTS 11
#endif
#ifndef KD_T1_FRONT
#define KD_T1_FRONT 256
#endif
#ifndef KD_WG_PER_CU
#define KD_WG_PER_CU 3
#endif
// (the lattice tail's collector keeps its tables in the LDS key array: kd_lattice_tail.h)
#ifndef KD_MAX_PIECES
#define KD_MAX_PIECES 1536
#endif
#ifndef KD_COLLECT_FRAMES
#define KD_COLLECT_FRAMES 64
#endif
#else
#define KD_GEOM_KIND 1
#endif

#ifdef KD_GEOM_SMALL
#define SearchKernel SearchKernelSmall  // (one name per geometry in traces and code-object listings)
#endif

namespace kd {
namespace {

constexpr int kKind = KD_GEOM_KIND;  // kKindSmall / kKindLarge (kd_device.h)
static_assert(KD_NUM_STATS + kProfSlots <= kStatStride, "a row of counters");

#ifndef KD_BLOCK
#define KD_BLOCK 768
#endif
// Threads per utterance.  768 = three waves per SIMD gives each thread 168 vector
// registers; at 1024 (128 registers) the search spills to scratch memory in its hot
// loops, and every reload there is an exposed memory round trip.
constexpr int kBlock = KD_BLOCK;
constexpr int kWaves = kBlock / 64;
// Geometry knobs (the defaults are the shipped configuration; tools/ builds others):
//   KD_LOG2_SLOTS  slots of the LDS token table (tier 2); tier 1 uses half of them
//   KD_T1_FRONT    largest frontier the LDS-resident tier takes
//   KD_VMAX        widest log-likelihood row staged in LDS
//   KD_WG_PER_CU   workgroups that must fit a CU together (LDS and registers)
#ifndef KD_LOG2_SLOTS
#define KD_LOG2_SLOTS 13
#endif
#ifndef KD_T1_FRONT
#define KD_T1_FRONT 1024
#endif
#ifndef KD_VMAX
#define KD_VMAX 2048
#endif
#ifndef KD_GT_SIZE_X4
#define KD_GT_SIZE_X4 8  // global table: at least this many quarter slots per emitting candidate
#endif
#ifndef KD_WG_PER_CU
#define KD_WG_PER_CU 1
#endif
constexpr int kLog2Slots = KD_LOG2_SLOTS;
constexpr int kSlots = 1 << kLog2Slots;
constexpr int kLdsFillLimit = kSlots * 3 / 4;
constexpr int kVmax = KD_VMAX;  // log-likelihood row staged in LDS when V <= kVmax
// the next frame's row is carried in registers through the emitting filter
constexpr int kPfRegs = (kVmax + KD_BLOCK - 1) / KD_BLOCK;
#ifndef KD_SMALL_DEG
#define KD_SMALL_DEG 4
#endif
#ifndef KD_GROUP
#define KD_GROUP 2  // (round 6, on the register allocation of that round: 40.9 against 40.5 M, 43.1 against 42.4; 4 until then)
#endif
// Label index (IndexedArcs): on (1) / off (0), and the smallest frontier of an LDS-resident frame
// that uses it -- below that a frame's few high-degree tokens are a handful of wave rows whose
// arcs arrive in ONE round trip, against the index's chain of dependent probes.
#ifndef KD_LABEL_INDEX
#define KD_LABEL_INDEX 1
#endif
// Noted tokens a lane looks up side by side in the big tiers' label-index pass (Filter).
#ifndef KD_IDX_NOTE_ILP
#define KD_IDX_NOTE_ILP 4
#endif
// Room for notes in that pass: 0 = what the two epsilon worklists hold; test builds set a few
// dozen so that a frame meets the full list (the token then looks its arc up inline).
#ifndef KD_IDX_NOTE_CAP
#define KD_IDX_NOTE_CAP 0
#endif
#ifndef KD_IDX_MIN_FRONT
#define KD_IDX_MIN_FRONT 192
#endif
constexpr int kIdxMinFront = KD_IDX_MIN_FRONT;
#ifndef KD_TOP_MIN_QUEUE
#define KD_TOP_MIN_QUEUE (2 * KD_IDX_MIN_FRONT / 3)
#endif
constexpr int kTopMinQueue = KD_TOP_MIN_QUEUE;  // (see the row swap of a tier-1 frame)
// The best token's row is scored ahead of the others (a tight running cutoff early: half as many
// candidates queued) in frames of at least this many tokens; a smaller frame queues the little
// it has either way, and the row's round trip ahead of the work items is the dearer.
#ifndef KD_LEAD_MIN_FRONT
#define KD_LEAD_MIN_FRONT 0
#endif
constexpr int kLeadMinFront = KD_LEAD_MIN_FRONT;
constexpr int kSmallDeg = KD_SMALL_DEG;  // tokens with at most this many arcs stay on their lane
constexpr int kGroup = KD_GROUP;  // queued tokens a wave scores at a time
#ifndef KD_BIG_GROUP
#define KD_BIG_GROUP KD_GROUP
#endif
constexpr int kBigGroup = KD_BIG_GROUP;  // ... in the big tiers' filter (BigLoop)
// Tier 1 (frames whose frontier is at most one token per lane -- 97% of the frames
// of an HLG decode): the lower half of the table arrays is a 4096-slot table, the
// upper half holds the first kQL entries of the candidate queue, and the frontier
// itself (state, cost, arc range) stays in LDS from one frame to the next.
constexpr int kT1Log2 = kLog2Slots - 1;
constexpr int kT1Slots = 1 << kT1Log2;
constexpr int kT1Front = KD_T1_FRONT;
constexpr int kQL = kT1Slots / 2;
#ifndef KD_T1_SOFTQ
#define KD_T1_SOFTQ (3 * (1 << (KD_LOG2_SLOTS - 2)))  // three times the LDS part of the queue (kQL)
#endif
constexpr int kT1SoftQueue = KD_T1_SOFTQ;         // (0: off) see RunUtterance
constexpr int kT1SoftFront = KD_T1_FRONT * 3 / 4;
static_assert(KD_T1_SOFTQ == 0 || KD_T1_SOFTQ >= 2 * kQL, "the soft limit is a multiple of the LDS queue part");
constexpr int kT1W16 = 512;    // tokens of at most 16 arcs (in Shared::hist)
constexpr int kSecondPart = 0x8000;  // 16-lane list entry: arcs 32..47 of a token of 33..48 arcs
static_assert(KD_T1_FRONT <= kSecondPart, "work-list entries are frontier indices below the flag");
static_assert(kBlock >= 128 && kT1Front >= kBlock && kSlots == 2 * kT1Slots && 2 * kQL == kT1Slots,
              "tier-1 LDS overlay");
// Cost keys: order-preserving 64-bit images of costs, so that one integer atomicMin
// relaxes a state.  FasterDecoder / SimpleDecoder costs are doubles (CostKey).  The
// lattice decoders' costs are floats: their key is the float's 32-bit key in the
// high word (low word zero) -- no f64 conversion and no 64-bit arithmetic per arc
// (MI355X issues those at a fraction of the 32-bit rate).  Keys never leave this
// file; a launch uses one mapping throughout.
template <int M>
__device__ __forceinline__ unsigned long long CK(double d) {
  if constexpr (M == 1) return static_cast<unsigned long long>(FloatKey(static_cast<float>(d))) << 32;
  else return CostKey(d);
}
template <int M>
__device__ __forceinline__ unsigned long long CKf(float f) {  // (lattice: the cost as it was computed)
  if constexpr (M == 1) return static_cast<unsigned long long>(FloatKey(f)) << 32;
  else return CostKey(static_cast<double>(f));
}
template <int M>
__device__ __forceinline__ float KCf(unsigned long long k) {
  if constexpr (M == 1) return KeyFloat(static_cast<uint32_t>(k >> 32));
  else return static_cast<float>(KeyCost(k));
}
template <int M>
__device__ __forceinline__ double KC(unsigned long long k) {
  if constexpr (M == 1) return static_cast<double>(KeyFloat(static_cast<uint32_t>(k >> 32)));
  else return KeyCost(k);
}
// key of +infinity
template <int M>
constexpr unsigned long long kCostInf = M == 1 ? 0xFF80000000000000ull : 0xFFF0000000000000ull;
// Global-table tier: tokens / queue entries a lane reads at a time in the passes that
// only read the table.  Every read there is an L2 round trip (agent-scope loads), so
// a lane keeps kSpillIlp independent ones in flight instead of one.  (The insert
// pass gains nothing from the same treatment -- measured: it runs at the rate the CU
// issues scattered L2 atomics, about four cycles an entry.)
constexpr int kSpillIlp = 4;
// Frame results besides 0 and the KD_UTT_* bits: run the table half again in the
// next larger table / run the whole frame again in the big path.
constexpr uint32_t kRetry = 0x100u, kRedo = 0x200u;

struct Shared {
  // token table (LDS mode); in spill mode `cost` is reused as the worklist bitmask
  unsigned long long cost[kSlots];
  uint32_t key[kSlots];
  uint32_t arc[kSlots];
  // queue of high-degree tokens of the current chunk; between tier-1 frames the
  // same arrays hold the frontier: cost, emitting arc range, state
  double it_cost[kT1Front];
  int32_t it_abeg[kT1Front];
  int32_t it_deg[kT1Front];
  uint32_t it_src[kT1Front];
  float ll[kVmax];
  uint32_t bitmask[kSlots / 32];
  int32_t hist[256];
  unsigned long long red64[kWaves];
  unsigned long long min_new;   // min cost key seen by the emitting filter
  unsigned long long min_stored;  // simple mode: min stored cost after the emitting insert
  unsigned long long f_best;    // CostKey of the best cost of the frontier to expand
  unsigned long long f_best_next;
  long long arena_next;         // next free record of this utterance's token-pool chunk
  long long arena_end;          // ... and the chunk's end
  long long base_out;
  double best_cost;             // the frontier's best token ...
  int32_t best_abeg, best_deg;  // ... and its emitting arc range
  long long link_next;          // lattice: next free link of this utterance's link-pool chunk
  long long link_end;
  unsigned long long ud_ll;     // the utterance's UttCtl::ll and ::offset (read once from
  int32_t ud_offset, ud_pad;    // pinned host memory when the utterance is taken up)
  // The row in `ll`: each wave's share of its two highest log-likelihoods (top_a0 >= top_a1)
  // and the column of the highest -- written by whoever stages the row, combined by every
  // wave that needs them (LoadRowTop): the label index asks "can any label but the likeliest
  // still pass this token's cutoff?"
  float top_a0[kWaves], top_a1[kWaves];
  int32_t top_i0[kWaves];
  int32_t top_valid;            // ... have been noted for the row in `ll` (0: the frame does without the index)
  int32_t alloc_ok;             // result of the last chunk allocation (0 ok, else a KD_UTT_* bit)
  // what a tier-1 frame leaves for the waves that sat it out, by frame parity:
  // result, new frontier (size, arena base), and what a replay in the big table needs
  struct {
    double cutoff;
    long long base_out;
    uint32_t r;
    int32_t n_in;
    int32_t q_emit;
    int32_t pad;
  } ctl[2];
  // ---- per-frame counters: one contiguous run of words, zeroed together when a frame
  // ends (ResetFrame; kFrameZeroWords of them, from f_nlt)
  int32_t f_nlt, f_nle;
  int32_t n_live;               // slots holding a token this frame (== live-list length)
  int32_t q_n;                  // candidate queue length
  int32_t n_emit_links;         // lattice: emitting entries below the final cutoff
  int32_t n_eps_links;          // lattice: epsilon links written so far
  int32_t n_emit_written;       // lattice: emitting links written so far
  int32_t n_act, n_inact;       // simple: tokens kept / dropped by PruneToks
  int32_t overflow;
  int32_t wl_count[2];
  int32_t it_count[2], it_next[2];
  int32_t it_n32, it_n16;       // tier 1: tokens in the 32-lane and 16-lane lists
  int32_t it_next32, it_next16;
  // ---- (end of the zeroed run)
  int32_t sel_bin, sel_k;
  int32_t idx_n;                // big tiers: tokens noted for the label-index pass (Filter)
  int32_t frame_cnt[4];         // this frame: expanded, eps_exp, arcs, eps_arcs
  union {  // whole launch (all values are non-negative)
    long long stat[KD_NUM_STATS];
    unsigned long long ustat[KD_NUM_STATS];
  };
#ifdef KD_PROFILE
  long long cyc[kProfSlots];
  long long lap_last;
  int lap_on, lap_pad;  // (KD_PROF_NMIN / KD_PROF_NMAX: only frames of that frontier size are timed)
#endif
};
static_assert(sizeof(Shared) + 128 <= 163840 / KD_WG_PER_CU, "LDS budget (the block and the scheduler's words)");

// The workgroup's LDS block.  Functions that are not inlined into the kernel reach
// it through this (an LDS pointer passed as an argument would become a flat one).
// (A static array, not `extern __shared__`: with the block's address and size known
// at compile time the field addresses are literals; as offsets from a dynamic base
// they were sixty more scalar registers held -- and spilled -- across the frame loop.
// Measured: 12.2 -> 11.9 ms per launch.)
__shared__ __attribute__((aligned(16))) unsigned char kd_smem[sizeof(Shared)];
__device__ __forceinline__ Shared &Lds() { return *reinterpret_cast<Shared *>(kd_smem); }

// Tier-1 views of the upper table halves: candidate queue entries [0, kQL) and
// the emitting filter's queue of high-degree frontier tokens.
__device__ __forceinline__ unsigned long long *T1QNk(Shared &s) { return &s.cost[kT1Slots]; }
__device__ __forceinline__ float *T1QAc(Shared &s) { return reinterpret_cast<float *>(&s.cost[kT1Slots + kQL]); }
__device__ __forceinline__ float *T1QGr(Shared &s) { return reinterpret_cast<float *>(&s.cost[kT1Slots + kQL]) + kQL; }
__device__ __forceinline__ uint32_t *T1QH(Shared &s) { return &s.key[kT1Slots]; }
__device__ __forceinline__ uint32_t *T1QArc(Shared &s) { return &s.key[kT1Slots + kQL]; }
__device__ __forceinline__ uint32_t *T1QSrc(Shared &s) { return &s.arc[kT1Slots]; }
// ... the live list (slots holding a token, in arena order) and the filter's work items
__device__ __forceinline__ uint16_t *T1Live(Shared &s) { return reinterpret_cast<uint16_t *>(&s.arc[kT1Slots + kQL]); }
__device__ __forceinline__ uint16_t *T1Heads(Shared &s) { return reinterpret_cast<uint16_t *>(&s.arc[kSlots - kT1Front / 2]); }
static_assert(2 * (kT1Slots * 3 / 4) + 2 * kT1Front <= 4 * kQL, "tier-1 spare region");
__device__ __forceinline__ uint16_t *T1W16(Shared &s) { return reinterpret_cast<uint16_t *>(s.hist); }


// Pointers that arrive inside the by-value parameter struct are generic
// ("flat") to the compiler; casting them to the global address space at the use
// site turns flat_load/flat_atomic into global_load/global_atomic.
#define KD_G __attribute__((address_space(1)))
// The launch parameters are read in place from the kernel-argument segment
// (constant address space: scalar loads on demand, no copy held in registers).
typedef const __attribute__((address_space(4))) SearchParams CParams;
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
// The first half of a state record (arc_begin, emit_end: the emitting arc range) with one
// global_load_dwordx2.  Where only the range is wanted this is not a saving of bandwidth
// but of a wait: hipcc reuses the dead upper half of a dwordx4 destination at once, and
// must then wait for the load to complete before it may overwrite those registers.
typedef int v2i __attribute__((ext_vector_type(2)));
struct ArcRange { int32_t arc_begin, emit_end; };
static_assert(offsetof(StateRec, arc_begin) == 0 && offsetof(StateRec, emit_end) == 4 && sizeof(StateRec) == 16,
              "LoadArcRange reads the first two words of a state record");
__device__ __forceinline__ ArcRange LoadArcRange(const StateRec *base, uint32_t i) {
  const v2i v = reinterpret_cast<const KD_G v2i *>(ToG(base))[2 * static_cast<size_t>(i)];
  ArcRange r;
  r.arc_begin = v.x; r.emit_end = v.y;
  return r;
}
__device__ __forceinline__ ArcRec LoadArcRec(const ArcPacked *base, uint32_t i) {
  const v4i v = reinterpret_cast<const KD_G v4i *>(ToG(base))[i];
  ArcRec r;
  const uint32_t lab = static_cast<uint32_t>(v.x);
  r.ilabel = static_cast<int32_t>(lab & static_cast<uint32_t>(kMaxIlabel));
  r.dst_degs = lab >> kIlabelBits;
  r.dst_arcs = static_cast<uint32_t>(v.y);
  r.weight = __int_as_float(v.z);
  r.next = static_cast<uint32_t>(v.w);
  return r;
}

// A load that is issued whether or not the lane wants it (`on` false: arc `safe`,
// which the caller knows to be a valid index).  hipcc puts a conditional load in a
// branch of its own and waits for it there, so loads written as
// "if (on) x = load" complete one after the other; written this way a group of them
// is in flight together.
__device__ __forceinline__ ArcRec LoadArcRecIf(const ArcPacked *base, uint32_t i, bool on, uint32_t safe) {
  return LoadArcRec(base, on ? i : safe);
}

// The input label of arc `aidx` as a column of a row that is read IN PLACE (rows of graphs with
// escaped labels are never in LDS: kd_device.h kIlabelEscape).
__device__ __forceinline__ int32_t FullIlabel(CParams &p, const ArcRec &a, uint32_t aidx) {
  int32_t il = a.ilabel;
  if (il == kIlabelEscape) il = ToG(p.big_ilabel)[aidx];
  return il;
}

// Per-frame, per-lane work counters, folded into LDS at the end of the frame.
struct Counters {
  int tokens_expanded = 0, eps_exp = 0, arcs_scanned = 0, eps_arcs = 0;
  // label index: arcs of expanded tokens that were never fetched, index entries read
  int arcs_skipped = 0, idx_probes = 0;
};

// Diagnostic build only (make prof): a lap timer on lane 0.  KD_LAP(s, i) adds
// the shader-clock cycles since the previous lap to bucket i.  The shipping
// build executes no stamps.
// -DKD_PROF_T1 details the LDS-resident small frames and lumps the others into
// bucket 14; the default details the others and lumps the small frames into 19.
// -DKD_PROF_SPILL details the global-table halves of the spill frames instead.
#if defined(KD_PROF_T1)
constexpr int kBucketSmall = -1, kBucketBig = 14, kBucketSpill = 14;
#elif defined(KD_PROF_SPILL)
constexpr int kBucketSmall = 19, kBucketBig = 19, kBucketSpill = -1;
#else
constexpr int kBucketSmall = 19, kBucketBig = -1, kBucketSpill = 14;
#endif
#if defined(KD_PROFILE) && !defined(KD_PROF_SIZE)
#define KD_LAP(s, i)                                                              \
  do {                                                                            \
    if (threadIdx.x == 0) {                                                       \
      const long long now_ = static_cast<long long>(__builtin_amdgcn_s_memtime()); \
      if ((s).lap_on) (s).cyc[KD_BUCKET < 0 ? (i) : KD_BUCKET] += now_ - (s).lap_last;      \
      (s).lap_last = now_;                                                        \
    }                                                                             \
  } while (0)
#else
#define KD_LAP(s, i)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// every outstanding global store of the wave (s_waitcnt vmcnt(0)), which costs a
// memory round trip whenever stores are in flight; where the next phase only
// needs the other waves' LDS writes, this lets the stores keep draining.
__device__ __forceinline__ void LdsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  KdJitter();
}

// Barrier for phases whose cross-wave traffic is the token table: LDS in the LDS
// mode, global memory in the spill mode.
template <bool LDS>
__device__ __forceinline__ void TableBarrier() {
  if constexpr (LDS) LdsBarrier(); else __syncthreads();
}

// The threads that run a frame: the whole workgroup.  (Rounds 3-5 could hand a frame of few
// tokens to the first four waves while the other eight executed only its barriers,
// KD_T1_NARROW: never faster, never shipped -- docs/HISTORY.md, round 6 took it out.)
struct Team {
  static constexpr int nt = kBlock;
  // LDS traffic only, or (with_global) global stores too
  __device__ __forceinline__ void Sync(bool with_global) const {
    if (with_global) __syncthreads(); else LdsBarrier();
  }
};

// Values that are the same in every lane (read from LDS, say) live in vector
// registers unless told otherwise.  Uni() moves them to scalar registers, whose
// spills go to lanes of a vector register instead of scratch memory.
__device__ __forceinline__ int Uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t Uni(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
__device__ __forceinline__ unsigned long long Uni(unsigned long long v) {
  const uint32_t lo = Uni(static_cast<uint32_t>(v)), hi = Uni(static_cast<uint32_t>(v >> 32));
  return static_cast<unsigned long long>(hi) << 32 | lo;
}
__device__ __forceinline__ long long Uni(long long v) {
  return static_cast<long long>(Uni(static_cast<unsigned long long>(v)));
}
__device__ __forceinline__ double Uni(double v) {
  return __longlong_as_double(Uni(__double_as_longlong(v)));
}
__device__ __forceinline__ float Uni(float v) { return __int_as_float(Uni(__float_as_int(v))); }

// Per-frame counters, reset at the end of a frame for the next one: by the lanes of
// one wave (`lane0` is that wave's first thread), one word each -- not word after
// word by a single thread while the workgroup waits at the frame barrier.
constexpr int kFrameZeroWords = 20;
// ... of which the work-item lists' counters and cursors: it_count[2], it_next[2], it_n32, it_n16,
// it_next32, it_next16 -- one contiguous run too, reset between a pass over the lists and the next
constexpr int kItemWords = 8;
static_assert(offsetof(Shared, it_next16) - offsetof(Shared, it_count) == 4 * (kItemWords - 1),
              "the work-item counters are one contiguous run");
template <int M>
__device__ __forceinline__ void ResetFrame(Shared &s, int lane0 = 0) {
  static_assert(offsetof(Shared, it_next16) - offsetof(Shared, f_nlt) == 4 * (kFrameZeroWords - 1),
                "the per-frame counters are one contiguous run");
  const int l = static_cast<int>(threadIdx.x) - lane0;
  if (l >= 0 && l < kFrameZeroWords) (&s.f_nlt)[l] = 0;
  if (l == kFrameZeroWords) s.min_new = kCostInf<M>;
  if (l == kFrameZeroWords + 1) s.min_stored = kCostInf<M>;
  if (l == kFrameZeroWords + 2) s.f_best_next = kCostDead;
}

// ---- wave-wide primitives: DPP row operations and v_readlane -- no LDS -----------------
// A reduction over the wave's 64 lanes is four DPP steps inside each row of 16 lanes (quad
// permutes, then the mirror of 8 and of 16 lanes: after step k every lane holds the
// aggregate of its group of 2^k), then the four rows' values are read into scalar registers
// (v_readlane) and combined there: the result is wave-uniform.  Until round 5 these were
// __shfl_xor loops, which hipcc compiles to ds_bpermute_b32 -- the LDS crossbar: six dependent
// round trips behind s_waitcnt lgkmcnt on the queue the token table's atomics use (234
// ds_bpermute in SearchKernel<1>), on the critical path of the running cutoff and of every
// counter.  `old` is what a lane whose source lane is inactive keeps: the operation's identity.
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;          // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int kDppHalfMirror = 0x141, kDppMirror = 0x140;  // lane i <- 7 - i of 8, 15 - i of 16
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int Dpp(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long DppMin64(unsigned long long v) {
  const uint32_t lo = static_cast<uint32_t>(Dpp<CTRL>(-1, static_cast<int>(static_cast<uint32_t>(v))));
  const uint32_t hi = static_cast<uint32_t>(Dpp<CTRL>(-1, static_cast<int>(static_cast<uint32_t>(v >> 32))));
  const unsigned long long y = static_cast<unsigned long long>(hi) << 32 | lo;
  return y < v ? y : v;
}
__device__ __forceinline__ unsigned long long ReadLane64(unsigned long long v, int l) {
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v)), l));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v >> 32)), l));
  return static_cast<unsigned long long>(hi) << 32 | lo;
}

// Sum over the wave (wave-uniform result).  (An atomicAdd of a per-lane value makes hipcc
// serialise the lanes in a waterfall loop, so reduce first and let one lane add.)
__device__ __forceinline__ int WaveSumInt(int v) {
  v += Dpp<kDppXor1>(0, v);
  v += Dpp<kDppXor2>(0, v);
  v += Dpp<kDppHalfMirror>(0, v);
  v += Dpp<kDppMirror>(0, v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
         __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ unsigned long long WaveMinU64(unsigned long long v) {
  v = DppMin64<kDppXor1>(v);
  v = DppMin64<kDppXor2>(v);
  v = DppMin64<kDppHalfMirror>(v);
  v = DppMin64<kDppMirror>(v);
  const unsigned long long a = ReadLane64(v, 0), b = ReadLane64(v, 16), c = ReadLane64(v, 32),
                           d = ReadLane64(v, 48);
  const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}

__device__ __forceinline__ float WaveMaxF(float v) {
  v = fmaxf(v, __int_as_float(Dpp<kDppXor1>(__float_as_int(v), __float_as_int(v))));
  v = fmaxf(v, __int_as_float(Dpp<kDppXor2>(__float_as_int(v), __float_as_int(v))));
  v = fmaxf(v, __int_as_float(Dpp<kDppHalfMirror>(__float_as_int(v), __float_as_int(v))));
  v = fmaxf(v, __int_as_float(Dpp<kDppMirror>(__float_as_int(v), __float_as_int(v))));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)),
              b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
              c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)),
              d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// Sum of floats over the wave in a FIXED association (pairs, quads, eights, rows, then the
// four rows left to right): the same bits whichever wave computes it.
__device__ __forceinline__ float WaveSumF(float v) {
  v += __int_as_float(Dpp<kDppXor1>(0, __float_as_int(v)));
  v += __int_as_float(Dpp<kDppXor2>(0, __float_as_int(v)));
  v += __int_as_float(Dpp<kDppHalfMirror>(0, __float_as_int(v)));
  v += __int_as_float(Dpp<kDppMirror>(0, __float_as_int(v)));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)),
              b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
              c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)),
              d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return ((a + b) + c) + d;
}

// Inclusive prefix sum over the wave's lanes: doubling steps inside a row (row_shr 1, 2, 4, 8:
// a lane without a source adds 0), then lane 15 of rows 0 and 2 into the row above
// (row_bcast15 to rows 1 and 3), then lane 31 into the upper half (row_bcast31).
__device__ __forceinline__ int WaveInclusiveSumInt(int x) {
  x += Dpp<0x111>(0, x);
  x += Dpp<0x112>(0, x);
  x += Dpp<0x114>(0, x);
  x += Dpp<0x118>(0, x);
  x += Dpp<0x142, 0xA>(0, x);
  x += Dpp<0x143, 0xC>(0, x);
  return x;
}

// The two largest values of a set and where the largest is: (a0, i0, a1), a0 >= a1.  Ties go to
// the lower index, so the result does not depend on how the set was split.  NaNs are never taken
// (every comparison with one is false): a label whose log-likelihood is NaN cannot pass a cutoff.
struct Top2 {
  float a0, a1;
  int i0;
};
__device__ __forceinline__ Top2 Top2Empty() {
  return Top2{-__int_as_float(0x7F800000), -__int_as_float(0x7F800000), 0x7FFFFFFF};
}
__device__ __forceinline__ void Top2Add(Top2 &t, float v, int i) {
  if (v > t.a0 || (v == t.a0 && i < t.i0)) { t.a1 = t.a0; t.a0 = v; t.i0 = i; }
  else if (v > t.a1) t.a1 = v;
}
__device__ __forceinline__ Top2 Top2Merge(const Top2 &x, const Top2 &y) {
  const bool take_y = y.a0 > x.a0 || (y.a0 == x.a0 && y.i0 < x.i0);
  Top2 r;
  r.a1 = take_y ? (x.a0 > y.a1 ? x.a0 : y.a1) : (x.a1 > y.a0 ? x.a1 : y.a0);
  r.a0 = take_y ? y.a0 : x.a0;
  r.i0 = take_y ? y.i0 : x.i0;
  return r;
}
template <int CTRL>
__device__ __forceinline__ Top2 Top2Step(const Top2 &t) {
  Top2 y;
  y.a0 = __int_as_float(Dpp<CTRL>(__float_as_int(t.a0), __float_as_int(t.a0)));
  y.a1 = __int_as_float(Dpp<CTRL>(__float_as_int(t.a1), __float_as_int(t.a1)));
  y.i0 = Dpp<CTRL>(t.i0, t.i0);
  return Top2Merge(t, y);
}
__device__ __forceinline__ Top2 Top2ReadLane(const Top2 &t, int l) {
  Top2 r;
  r.a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.a0), l));
  r.a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.a1), l));
  r.i0 = __builtin_amdgcn_readlane(t.i0, l);
  return r;
}
// ... of the wave's lanes' sets (wave-uniform).  Call with the whole wave active.
__device__ __forceinline__ Top2 WaveTop2(Top2 t) {
  t = Top2Step<kDppXor1>(t);
  t = Top2Step<kDppXor2>(t);
  t = Top2Step<kDppHalfMirror>(t);
  t = Top2Step<kDppMirror>(t);
  return Top2Merge(Top2Merge(Top2ReadLane(t, 0), Top2ReadLane(t, 16)),
                   Top2Merge(Top2ReadLane(t, 32), Top2ReadLane(t, 48)));
}

// This wave's share of the top two of the row in `ll` (whole workgroup; the row is complete
// in LDS; a barrier must follow before LoadRowTop).
__device__ __forceinline__ void StoreRowTop(Shared &s, int V) {
  Top2 t = Top2Empty();
  for (int v = threadIdx.x; v < V; v += kBlock) Top2Add(t, s.ll[v], v);
  t = WaveTop2(t);
  if ((threadIdx.x & 63) == 0) {
    s.top_a0[threadIdx.x >> 6] = t.a0;
    s.top_a1[threadIdx.x >> 6] = t.a1;
    s.top_i0[threadIdx.x >> 6] = t.i0;
  }
}
// The row's lowest acoustic cost m0 (= minus its highest log-likelihood), that label's column
// l0, and the second lowest m1 (+inf for a row of one column): wave-uniform.
static_assert(kWaves <= 16, "LoadRowTop combines the waves' shares in one row of 16 lanes");
__device__ __forceinline__ void LoadRowTop(Shared &s, float &m0, float &m1, int &l0) {
  const int lane = threadIdx.x & 63;
  Top2 t = Top2Empty();
  const int w = lane < kWaves ? lane : 0;
  const float a0 = s.top_a0[w], a1 = s.top_a1[w];
  const int i0 = s.top_i0[w];
  if (lane < kWaves) { t.a0 = a0; t.a1 = a1; t.i0 = i0; }
  t = Top2Step<kDppXor1>(t);
  t = Top2Step<kDppXor2>(t);
  t = Top2Step<kDppHalfMirror>(t);
  t = Top2Step<kDppMirror>(t);
  t = Top2ReadLane(t, 0);
  m0 = -t.a0;
  m1 = -t.a1;
  l0 = t.i0;
}

// atomicMin(p, v) from the lanes with `on`, to one LDS word.  hipcc turns such a
// same-address atomic of per-lane values into a loop over the lanes (some fifteen
// instructions each); reduce over the wave first and let one lane go to LDS.  The
// wave takes the short way out when no lane has anything to contribute.
__device__ __forceinline__ void LaneMin(unsigned long long *p, unsigned long long v, bool on) {
  if (__ballot(on) == 0) return;
  const unsigned long long w = WaveMinU64(on ? v : ~0ull);
  if ((threadIdx.x & 63) == 0) atomicMin(p, w);
}

// counter++ by every lane that is active here: one LDS atomic for the wave (the
// build runs without the compiler's atomic optimizer, see the Makefile).  AggInc
// returns each lane's own old value, AggAdd1 nothing.
__device__ __forceinline__ int AggInc(int32_t *ctr) {
  const unsigned long long m = __ballot(1);
  const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
  int base = 0;
  if (rank == 0) base = atomicAdd(ctr, __popcll(m));
  return __builtin_amdgcn_readfirstlane(base) + rank;
}
__device__ __forceinline__ void AggAdd1(int32_t *ctr) {
  const unsigned long long m = __ballot(1);
  const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
  if (rank == 0) atomicAdd(ctr, __popcll(m));
}

__device__ __forceinline__ unsigned long long BlockMinU64(Shared &s, unsigned long long v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = WaveMinU64(v);
  if (lane == 0) s.red64[wave] = v;
  LdsBarrier();
  unsigned long long r = s.red64[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) r = s.red64[w] < r ? s.red64[w] : r;
  LdsBarrier();
  return r;
}

// ---- pools -----------------------------------------------------------------------
// Token records and forward links come from two pools shared by all utterances of the
// batch.  An utterance owns a chunk of each ([arena_next, arena_end), [link_next,
// link_end) in LDS) and cuts one contiguous block per frame from it; when a block does
// not fit, ONE thread takes a new chunk with one atomic on the pool cursor (the rest of
// the old chunk is abandoned: at most one frame's worth).  Returns 0 or the KD_UTT_* bit
// of the pool that is exhausted -- the frame is then not committed, and the host grows
// the pool and relaunches (kd_api.hip).  Call from one thread.
// `want` units from a pool cursor, or -1 when they do not fit below `cap`.  Compare-and-
// swap, not add-then-take-back: an allocator that adds first and subtracts on failure
// moves, for a moment, the base of every allocation that succeeds in between, and its
// subtraction then pulls the cursor back UNDER them (two utterances end up with the same
// records -- seen as rare wrong lattices when a pool ran out mid-launch).
__device__ __forceinline__ long long PoolTake(unsigned long long *top, unsigned long long seen_top,
                                              unsigned long long base, unsigned long long cap,
                                              unsigned long long want) {
  unsigned long long old = seen_top;  // (a value the caller read a moment ago: usually still right)
  while (true) {
    if (old - base + want > cap) return -1;
    const unsigned long long seen = atomicCAS(top, old, old + want);
    if (seen == old) return static_cast<long long>(old - base);
    old = seen;
  }
}

__device__ __forceinline__ long long PoolTake(unsigned long long *top, unsigned long long base,
                                              unsigned long long cap, unsigned long long want) {
  return PoolTake(top, __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), base, cap, want);
}

// What the slow path of TakeChunks reads of one pool, requested TOGETHER before any of it is
// used: every one of these was a dependent round trip to memory of ONE lane with the whole
// workgroup waiting at the frame's barrier -- eight of them per new chunk, 21 k cycles, 3.6 %
// of the kernel on the bench workload (round 3, a diagnostic build that timed this function).
struct PoolView {
  int32_t chunk, chunk_len, n_free, n_retired;
  unsigned long long top;
};
__device__ __forceinline__ PoolView LoadPoolView(CParams &p, KD_G UttMem *m, bool links) {
  KD_G ChunkRing *ring = links ? &m->link : &m->tok;
  PoolView v;
  v.chunk = links ? m->link_chunk : m->arena_chunk;
  v.chunk_len = links ? m->link_chunk_len : m->arena_chunk_len;
  v.n_free = ring->n_free;
  v.n_retired = ring->n_retired;
  v.top = __hip_atomic_load(links ? &p.ctl->link_top : &p.ctl->tok_top, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

// A chunk of `want` records of pool `links ? link : token`: one of the utterance's own
// freed chunks if the request is of standard size, else the next of the pool.  -1: the
// pool is exhausted.
__device__ __forceinline__ long long PoolChunk(CParams &p, KD_G ChunkRing *ring, const PoolView &v,
                                               bool links, unsigned long long want) {
  const unsigned long long standard = links ? p.link_chunk : p.tok_chunk;
  if (want == standard && v.n_free > 0) {
    ring->n_free = v.n_free - 1;
    return ring->free_start[v.n_free - 1];
  }
  unsigned long long *top = links ? &p.ctl->link_top : &p.ctl->tok_top;
  const unsigned long long base = links ? p.link_base : p.tok_base, cap = links ? p.link_cap : p.tok_cap;
  // ONE atomic add, no compare-and-swap loop: a thousand utterances take a chunk every few
  // microseconds between them, a round trip to the cursor takes a few microseconds under
  // load, and a compare-and-swap from a value that old mostly fails -- a take cost 5-10 round
  // trips of one lane with its workgroup waiting at the frame's barrier (30 k cycles; 8 % of
  // the kernel in a diagnostic build that timed it, round 3).  The cursor is never taken back
  // (PoolTake says why it must not be): a take that does not fit leaves it beyond `cap`, where
  // every later take fails its check -- which is what an exhausted pool means -- and the few
  // records between the last block handed out and `cap` stay unused until the host has grown
  // the pool (kd_api.hip clamps the cursor it reads to the capacity).
  if (v.top - base + want > cap) return -1;  // (already exhausted: do not push the cursor further)
  const unsigned long long old = atomicAdd(top, want);
  if (old - base + want > cap) return -1;
  return static_cast<long long>(old - base);
}

__device__ __forceinline__ long long PoolChunk(CParams &p, KD_G ChunkRing *ring, bool links,
                                               unsigned long long want) {  // (the collector's)
  PoolView v;
  v.chunk = -1; v.chunk_len = 0; v.n_retired = 0;
  v.n_free = ring->n_free;
  v.top = __hip_atomic_load(links ? &p.ctl->link_top : &p.ctl->tok_top, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
  return PoolChunk(p, ring, v, links, want);
}

// Notes a nursery chunk that is full (`last` = the last frame plus one with a block in
// it) so that the collector can free it once that frame is archived.  A chunk taken for
// one oversized frame is noted as the standard-size pieces it can be cut into.
__device__ __forceinline__ void RetireChunk(KD_G ChunkRing *ring, const PoolView &v, int32_t standard,
                                            int32_t last) {
  if (v.chunk < 0) return;
  int32_t n = v.n_retired;
  for (int32_t off = 0; off + standard <= v.chunk_len && n < kChunkRing; off += standard, ++n) {
    ring->retired_start[n] = v.chunk + off;
    ring->retired_last[n] = last;
  }  // (what does not fit the ring is not recycled: lost until InitDecoding)
  ring->n_retired = n;
}

__device__ __forceinline__ uint32_t TakeChunks(CParams &p, Shared &s, int utt, int frame_plus_one,
                                               long long need_tokens, long long need_links) {
  const bool new_tok = s.arena_next + need_tokens > s.arena_end;
  const bool new_link = need_links > 0 && s.link_next + need_links > s.link_end;
  if (!new_tok && !new_link) return 0;
  KD_G UttMem *m = ToG(p.mem) + utt;
  const PoolView tv = LoadPoolView(p, m, false), lv = LoadPoolView(p, m, true);
  if (new_tok) {
    const unsigned long long want =
        static_cast<unsigned long long>(need_tokens > p.tok_chunk ? need_tokens : p.tok_chunk);
    const long long b = PoolChunk(p, &m->tok, tv, false, want);
    if (b < 0) return KD_UTT_ARENA_FULL;
    RetireChunk(&m->tok, tv, p.tok_chunk, frame_plus_one - 1);
    m->arena_chunk = static_cast<int32_t>(b);
    m->arena_chunk_len = static_cast<int32_t>(want);
    s.arena_next = b;
    s.arena_end = b + static_cast<long long>(want);
  }
  if (new_link) {
    const unsigned long long want =
        static_cast<unsigned long long>(need_links > p.link_chunk ? need_links : p.link_chunk);
    const long long b = PoolChunk(p, &m->link, lv, true, want);
    if (b < 0) return KD_UTT_LINKS_FULL;
    // (the links of frame_plus_one - 1's tokens are being written: the old chunk's last are older)
    RetireChunk(&m->link, lv, p.link_chunk, frame_plus_one - 1);
    m->link_chunk = static_cast<int32_t>(b);
    m->link_chunk_len = static_cast<int32_t>(want);
    s.link_next = b;
    s.link_end = b + static_cast<long long>(want);
  }
  return 0;
}

// ---- table access: LDS (ds_* ops) or global memory (L2 atomics) --------------
// The LDS table is three arrays (bank-friendly).  The global table is an array of 16-byte
// slots (the words by decoder: SlotCost / SlotKey below): a frame scatters its tokens over
// the table, and with one array per word every token dirtied three 64-byte sectors per
// phase (claim + relax, resolve, clear) that the L2 wrote back in between -- 2.3 MB of
// HBM writes per global-table frame of 7.5 k tokens, 60 % of all the kernel wrote
// (tools/write_breakdown.sh, round 3).  One sector per token now.
template <bool LDS>
struct Table {
  Shared &s;
  KD_G uint32_t *gslot;  // 4 words per slot; KeyP / CostP / ArcP and the word-wise loads and
                         // stores below are the FasterDecoder / SimpleDecoder layout
  uint32_t *bitmask;
  int log2_slots;
  uint32_t mask;

  __device__ __forceinline__ uint32_t *KeyP(uint32_t h) const { if constexpr (LDS) return &s.key[h]; else return ToFlat(&gslot[4ull * h + 2]); }
  __device__ __forceinline__ unsigned long long *CostP(uint32_t h) const { if constexpr (LDS) return &s.cost[h]; else return reinterpret_cast<unsigned long long *>(ToFlat(&gslot[4ull * h])); }
  __device__ __forceinline__ uint32_t *ArcP(uint32_t h) const { if constexpr (LDS) return &s.arc[h]; else return ToFlat(&gslot[4ull * h + 3]); }

  // Global-table words are updated by L2 atomics, which do not refresh this
  // CU's L1: read and write them with agent-scope (sc1) accesses.
  __device__ __forceinline__ uint32_t LoadKey(uint32_t h) const {
    if constexpr (LDS) return s.key[h];
    else return __hip_atomic_load(KeyP(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void StoreKey(uint32_t h, uint32_t v) const {
    if constexpr (LDS) s.key[h] = v;
    else __hip_atomic_store(KeyP(h), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ unsigned long long LoadCost(uint32_t h) const {
    if constexpr (LDS) return s.cost[h];
    else return __hip_atomic_load(CostP(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // The cost word as decoder M keys it.  The lattice decoders' keys are 32 bits wide (the
  // high word of the LDS table's 64-bit word; a word of its own in the global table).
  template <int M>
  __device__ __forceinline__ unsigned long long LoadCostM(uint32_t h) const {
    if constexpr (LDS || M != 1) {
      return LoadCost(h);
    } else {
      // (lattice layout: the cost key is the slot's third word)
      const uint32_t hi = __hip_atomic_load(ToFlat(&gslot[4ull * h + 2]), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
      return static_cast<unsigned long long>(hi) << 32;
    }
  }
  __device__ __forceinline__ void StoreCost(uint32_t h, unsigned long long v) const {
    if constexpr (LDS) s.cost[h] = v;
    else __hip_atomic_store(CostP(h), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ uint32_t LoadArc(uint32_t h) const {
    if constexpr (LDS) return s.arc[h];
    else return __hip_atomic_load(ArcP(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void StoreArc(uint32_t h, uint32_t v) const {
    if constexpr (LDS) s.arc[h] = v;
    else __hip_atomic_store(ArcP(h), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // The whole slot of the global table in one 16-byte agent-scope load (cost lo, hi, key,
  // arc): the passes that need two or three of its words issue one L2 request, not three
  // (8-byte and narrower sc1 accesses run at about half the 16-byte rate per access).
  typedef uint32_t SlotV __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ SlotV LoadSlot(uint32_t h) const {
    static_assert(!LDS, "global table only");
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        ToFlat(gslot), 0, static_cast<int>(16u << log2_slots), 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(h << 4), 0, 16 /* sc1 */);
  }
  // The slot's words by decoder.  FasterDecoder / SimpleDecoder (64-bit cost keys):
  // {cost lo, cost hi, state key, winning arc}.  Lattice decoders (32-bit cost keys, no
  // back-pointer): {place in the frame's block, -, cost key, state key} -- state key and
  // cost key form ONE 64-bit word, so that a single compare-and-swap claims an empty slot
  // AND gives it its first cost (ClaimRelaxLattice).
  template <int M>
  static __device__ __forceinline__ unsigned long long SlotCost(const SlotV &v) {
    if constexpr (M == 1) return static_cast<unsigned long long>(v.z) << 32;
    else return static_cast<unsigned long long>(v.y) << 32 | v.x;
  }
  template <int M>
  static __device__ __forceinline__ uint32_t SlotKey(const SlotV &v) { return M == 1 ? v.w : v.z; }
  static __device__ __forceinline__ uint32_t SlotPlace(const SlotV &v) { return v.x; }  // (lattice)
  __device__ __forceinline__ void StorePlace(uint32_t h, uint32_t place) const {
    static_assert(!LDS, "global table only");
    gslot[4ull * h] = place;
  }
  // Lattice decoders, global table: claim + relax.  The pass runs at the rate one CU issues
  // L2 atomics, and nineteen of twenty entries that reach it are their state's first token
  // of the frame (every entry below the cutoff is a forward link; links = 1.05 x tokens): the
  // compare-and-swap of {state key, cost key} against "empty" settles those with ONE atomic
  // instead of two.  A slot that already holds the state answers with its cost: only an
  // entry that lowers it follows up with the 64-bit min (equal state keys: ordered by cost).
  // `old`: the slot's cost key before (kCostDead: the state's first token).
  __device__ __forceinline__ uint32_t ClaimRelaxLattice(uint32_t k, unsigned long long nk,
                                                        unsigned long long &old) const {
    static_assert(!LDS, "global table only");
    const unsigned long long want = static_cast<unsigned long long>(k) << 32 | nk >> 32;
    uint32_t h = Hash(k & kStateMask);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      unsigned long long *w = reinterpret_cast<unsigned long long *>(ToFlat(&gslot[4ull * h + 2]));
      unsigned long long was = atomicCAS(w, ~0ull, want);
      if (was == ~0ull) {
        old = kCostDead;
        return h;
      }
      if (static_cast<uint32_t>(was >> 32) == k) {
        if (static_cast<uint32_t>(was) > static_cast<uint32_t>(want)) was = atomicMin(w, want);
        old = (was & 0xFFFFFFFFull) << 32;
        return h;
      }
      h = (h + 1) & mask;
    }
    old = 0;
    return kKeyEmpty;
  }
  // An empty slot.  (Global: one 16-byte store; stores go through to the L2, where the
  // table's atomics and agent-scope loads see them.)
  __device__ __forceinline__ void ClearSlot(uint32_t h) const {
    if constexpr (LDS) {
      s.key[h] = kKeyEmpty;
      s.cost[h] = kCostDead;
      s.arc[h] = kArcNone;
    } else {
      static_assert(kKeyEmpty == 0xFFFFFFFFu && kArcNone == 0xFFFFFFFFu && kCostDead == ~0ull, "all ones");
      typedef uint32_t V4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<KD_G V4 *>(&gslot[4ull * h]) = V4{~0u, ~0u, ~0u, ~0u};
    }
  }
  __device__ __forceinline__ uint32_t Hash(uint32_t state) const {
    return (state * 2654435761u) >> (32 - log2_slots);
  }
  __device__ __forceinline__ uint32_t FillLimit() const {
    return 3u << (log2_slots - 2);
  }

  // Finds or claims the slot of `k` (state | has-eps flag); kKeyEmpty if the
  // probe sequence is exhausted (table full).
  __device__ __forceinline__ uint32_t Claim(uint32_t k) const {
    uint32_t h = Hash(k & kStateMask);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      if constexpr (LDS) {
        const uint32_t cur = s.key[h];
        if (cur == k) return h;
        if (cur == kKeyEmpty) {
          const uint32_t old = atomicCAS(&s.key[h], kKeyEmpty, k);
          if (old == kKeyEmpty || old == k) return h;
        }
      } else {
        // one L2 round trip per probe: the compare-and-swap also answers "is it k?"
        const uint32_t old = atomicCAS(KeyP(h), kKeyEmpty, k);
        if (old == kKeyEmpty || old == k) return h;
      }
      h = (h + 1) & mask;
    }
    return kKeyEmpty;
  }
};

// MODE selects the decoder whose arithmetic the kernel instance runs, so that
// each instance carries only its own code and registers:
//   0 FasterDecoder   1 LatticeSimple/LatticeFaster   2 SimpleDecoder
template <int MODE>
struct FrameCtx {
  static constexpr bool lattice = MODE == 1;
  static constexpr bool simple = MODE == 2;  // SimpleDecoder semantics
  CParams *pp;  // the launch parameters (kernel-argument segment); see KD_PHASE
  __device__ __forceinline__ CParams &P() const { return *pp; }
  Shared &s;
  int utt;
  long long base_in;
  int n_in;
  const KD_G float *ll_row;  // global row (used when V > kVmax)
  bool ll_in_lds;
  double weight_cutoff;  // tokens with cost < this are expanded
  float adaptive_beam;
  double cutoff;  // epsilon-closure cutoff == final emitting cutoff
  KD_G uint32_t *wl0, *wl1;  // epsilon worklists (slots)
  KD_G uint32_t *live;       // slots that hold a token this frame, in arena order
  // candidate queue: destination (state|flag, later its slot), arc, source, cost key
  KD_G uint32_t *q_h, *q_arc, *q_src;
  KD_G unsigned long long *q_nk;
  KD_G float *q_ac, *q_gr;
  int q_cap;
  bool replay;    // spill mode after an aborted LDS attempt: queue destinations were overwritten
  // This lane's lowest new cost so far in the emitting filter.  The shared running
  // minimum (s.min_new) is only updated where the wave is converged (PublishMin):
  // a same-address atomic of per-lane values inside divergent code is compiled to a
  // loop over the lanes.  A looser running cutoff only lengthens the queue.
  mutable unsigned long long run_min;
  bool min_only;  // simple: first filter pass only finds the minimum total cost
};

// A phase boundary of a tier-1 frame: the launch parameters' pointer is made opaque, so that what
// the compiler loads through it (scalar loads from the kernel-argument segment: pointers, options)
// is loaded again in the phase that uses it instead of once per frame and held -- and spilled --
// across all of them.
#define KD_PHASE(c)                                                                          \
  do {                                                                                       \
    unsigned long long pv_ = reinterpret_cast<unsigned long long>((c).pp);                   \
    asm volatile("" : "+v"(pv_));                                                            \
    (c).pp = reinterpret_cast<CParams *>(Uni(pv_));                                           \
  } while (0)

// Scores one arc out of a token with cost `csrc`: the reference's test at
// faster-decoder.cc:208-211 (emitting; `seen_min` is the lowest new cost seen so
// far) or :90-95 (epsilon).  Returns whether the arc survives; `nc` is the new
// token's cost, `aux` the lattice decoders' acoustic cost (emitting) / source cost
// (epsilon).  (SimpleDecoder's emitting arcs are scored in FilterArc.)
template <bool EPS, int M>
__device__ __forceinline__ bool ScoreArc(const FrameCtx<M> &c, const ArcRec &a, uint32_t aidx, double csrc,
                                         unsigned long long seen_min, double &nc, float &aux) {
  aux = 0.f;
  if constexpr (EPS) {
    if (c.lattice) {
      // lattice-simple-decoder.cc:170-174: float, strict
      const float cur_cost = static_cast<float>(csrc), tot_cost = cur_cost + a.weight;
      nc = tot_cost;
      aux = cur_cost;
      return tot_cost < static_cast<float>(c.cutoff);
    }
    nc = csrc + a.weight;     // faster-decoder.h:134
    return !(nc > c.cutoff);  // faster-decoder.cc:92
  } else {
    const float ll = c.ll_in_lds ? c.s.ll[a.ilabel - 1] : c.ll_row[FullIlabel(c.P(), a, aidx) - 1];
    if (c.lattice) {
      // lattice-simple-decoder.cc:383-390: float, cur + ac + graph
      const float ac_cost = -ll, cur_cost = static_cast<float>(csrc),
                  tot_cost = cur_cost + ac_cost + a.weight;
      nc = tot_cost;
      aux = ac_cost;
      return tot_cost < KCf<M>(seen_min) + c.adaptive_beam;
    }
    const float ac_cost = -1 * ll;   // :209
    nc = a.weight + csrc + ac_cost;  // :210 (== Token cost, faster-decoder.h:126)
    return nc < KC<M>(seen_min) + static_cast<double>(c.adaptive_beam);  // :211, :215-217
  }
}

// Candidate-queue entry `pos` of a tier-1 frame: the first kQL entries are in LDS.
//
// Lattice decoders: the entry also carries the DESTINATION's arc range, which the arc record
// names (ArcPacked) -- its first arc in the low word of the cost key (a lattice key is the float's
// 32-bit key in the HIGH word; all candidates of one state carry the same low word, so the
// 64-bit ds_min orders them by cost as before, and the slot ends up holding the word), its
// degrees in bits 16..27 of the source word (a tier-1 source is a frontier index or a slot:
// 13 bits and the epsilon flag).  The token a candidate creates thus knows its arcs: neither
// the epsilon rounds nor the next frame's frontier fetch a state record.  (Measured + 0.5 % on
// the bench: the fetch was L2-resident and cheap -- DESIGN section 5 -- but it is work removed,
// and a new token's arcs are known the moment it is created.)
constexpr uint32_t kT1SrcMask = kSrcEps | 0xFFFFu;
constexpr int kT1SrcDegShift = 16;
static_assert(kT1Slots <= 0x10000 && kT1Front <= 0x10000, "a tier-1 source fits 16 bits");
template <int M>
__device__ __forceinline__ void T1Put(const FrameCtx<M> &c, int pos, const ArcRec &a, uint32_t aidx,
                                      uint32_t srcid, unsigned long long nk, float aux) {
  Shared &s = c.s;
  if constexpr (M == 1) {
    nk |= a.dst_arcs;
    srcid |= a.dst_degs << kT1SrcDegShift;
  }
  if (pos < kQL) {
    T1QH(s)[pos] = a.next;
    T1QArc(s)[pos] = aidx;
    T1QSrc(s)[pos] = srcid;
    T1QNk(s)[pos] = nk;
    T1QAc(s)[pos] = aux;
    if (c.lattice) T1QGr(s)[pos] = a.weight;
  } else if (pos < c.q_cap) {
    c.q_h[pos] = a.next;
    c.q_arc[pos] = aidx;
    c.q_src[pos] = srcid;
    c.q_nk[pos] = nk;
    if (c.lattice || c.simple) c.q_ac[pos] = aux;
    if (c.lattice) c.q_gr[pos] = a.weight;
  }
}

// Scores one arc and queues it if it survives the (running) cutoff.
template <bool EPS, bool T1 = false, int M>
__device__ __forceinline__ void FilterArc(const FrameCtx<M> &c, const ArcRec &a, uint32_t aidx,
                                          double csrc, uint32_t srcid) {
  Shared &s = c.s;
  double nc;
  bool keep;
  float aux = 0.f;  // lattice: acoustic cost (emitting) / source cost (epsilon)
  unsigned long long seen_min = 0;
  if constexpr (!EPS) {
    seen_min = s.min_new;
    seen_min = c.run_min < seen_min ? c.run_min : seen_min;
  }
  if constexpr (!EPS && M == 2) {
    // simple-decoder.cc:167-176.  Two passes: the first only finds the minimum
    // total cost, the second queues what is below min + beam.  The token stores
    // prev.cost + (weight + ac) with the float sum first (simple-decoder.h:99).
    const float ll = c.ll_in_lds ? s.ll[a.ilabel - 1] : c.ll_row[FullIlabel(c.P(), a, aidx) - 1];
    const float acoustic_cost = -ll;
    const double total_cost = csrc + a.weight + acoustic_cost;
    if (c.min_only) {
      const unsigned long long tk = CK<M>(total_cost);
      if (tk < seen_min) c.run_min = tk;
      return;
    }
    keep = !(total_cost >= c.cutoff);
    nc = csrc + (a.weight + acoustic_cost);
    aux = acoustic_cost;
  } else {
    keep = ScoreArc<EPS>(c, a, aidx, csrc, seen_min, nc, aux);
  }
  if (keep) {
    const unsigned long long nk = CK<M>(nc);
    const int pos = AggInc(&s.q_n);
    if (T1) {
      T1Put(c, pos, a, aidx, srcid, nk, aux);
    } else if (pos < c.q_cap) {
      c.q_h[pos] = a.next;
      c.q_arc[pos] = aidx;
      c.q_src[pos] = srcid;
      c.q_nk[pos] = nk;
      if (c.lattice || c.simple) c.q_ac[pos] = aux;
      if (c.lattice) c.q_gr[pos] = a.weight;
    }
    if constexpr (!EPS) {
      if (!c.simple && nk < seen_min) c.run_min = nk;
    }
  }
}

// The arcs with input label `il` among the emitting arcs [abeg, abeg + deg) of a state, through
// the label index (kd_label_index.h: the range's (ilabel, offset) keys as an implicit 5-ary
// search tree, one 16-byte node per level -- three dependent loads for the 38-arc states that make
// up an HLG frontier), then `visit(arc index, arc)` for every match: a state may hold several
// arcs of one label, and the next one is looked for with the arc's fetch in flight.  Replaces,
// for a token whose cost leaves room for this one label only, the loop over all its arcs
// (faster-decoder.cc:204-211, lattice-simple-decoder.cc:375-399).  `probes` counts the
// nodes read, `fetched` the arcs.
typedef uint32_t IdxNodeVec __attribute__((ext_vector_type(4)));
typedef IdxNodeVec IdxNodeMem __attribute__((aligned(4)));  // (a node starts wherever its range does)
// The first key of label `il` in the range (kIdxNone: the state has no such arc) ...
__device__ __forceinline__ uint32_t IndexFind(CParams &p, int abeg, int deg, uint32_t from_key, int &probes) {
  const KD_G uint32_t *idx = ToG(p.lab_idx) + abeg;
  return LabelIndexLowerBound([&](int j) {
    const IdxNodeVec m = *reinterpret_cast<const KD_G IdxNodeMem *>(idx + 4 * j);
    return IdxNode{{m.x, m.y, m.z, m.w}};
  }, deg, from_key, probes);
}
// ... and, from it, the arcs of that label.
template <class Visit>
__device__ __forceinline__ void IndexedArcsFrom(CParams &p, int abeg, int deg, uint32_t il, uint32_t cand,
                                                int &probes, int &fetched, Visit visit) {
  while (cand != kIdxNone && IdxKeyLabel(cand) == il) {
    const uint32_t aidx = static_cast<uint32_t>(abeg) + IdxKeyOffset(cand);
    const ArcRec ar = LoadArcRec(p.arcs, aidx);
    // (the key says whether another arc of this label follows: looked for with the fetch in flight)
    cand = IdxKeyMore(cand) ? IndexFind(p, abeg, deg, cand + 1, probes) : kIdxNone;
    ++fetched;
    visit(aidx, ar);
  }
}
template <class Visit>
__device__ __forceinline__ void IndexedArcs(CParams &p, int abeg, int deg, uint32_t il, int &probes,
                                            int &fetched, Visit visit) {
  IndexedArcsFrom(p, abeg, deg, il, IndexFind(p, abeg, deg, IdxKey(il, 0), probes), probes, fetched, visit);
}

// Folds the lanes' running minima into s.min_new.  Call where the wave is converged.
template <int M>
__device__ __forceinline__ void PublishMin(const FrameCtx<M> &c) {
  if (c.run_min < c.s.min_new) atomicMin(&c.s.min_new, c.run_min);
}

// Scores N arcs per lane (`on[g]`: this lane holds arc g) and appends the survivors to
// the candidate queue in global memory with ONE queue allocation for the wave (the big
// tiers' form of the tier-1 frames' score_batch).  Call where the wave is converged.
// Not for SimpleDecoder's emitting arcs (two passes: FilterArc).
template <bool EPS, int N, int M>
__device__ __forceinline__ void ScoreBatchBig(const FrameCtx<M> &c, const ArcRec (&ar)[N],
                                              const uint32_t (&aidx)[N], const bool (&on)[N],
                                              const double (&cst)[N], const uint32_t (&src)[N]) {
  Shared &s = c.s;
  unsigned long long seen = 0;
  if constexpr (!EPS) {
    seen = s.min_new;
    seen = c.run_min < seen ? c.run_min : seen;
  }
  unsigned long long nk[N], bal[N];
  float aux[N];
  bool keep[N];
  int pre[N], total = 0;
  bool grouped = false;
  if constexpr (M == 1 && !EPS) grouped = c.ll_in_lds;  // (a constant in each copy of the frame)
  if (grouped) {
    // lattice decoders, emitting arcs, row in LDS: the rows' log-likelihood reads together,
    // the rows scored without a branch (see the tier-1 frames' score_batch)
    float llv[N];
#pragma unroll
    for (int g = 0; g < N; ++g) llv[g] = s.ll[on[g] ? ar[g].ilabel - 1 : 0];
    const float lim = KCf<M>(seen) + c.adaptive_beam;
#pragma unroll
    for (int g = 0; g < N; ++g) {
      const float ac_cost = -llv[g], cur_cost = static_cast<float>(cst[g]),
                  tot_cost = cur_cost + ac_cost + ar[g].weight;
      keep[g] = on[g] && tot_cost < lim;
      nk[g] = CKf<M>(tot_cost);
      aux[g] = ac_cost;
      bal[g] = __ballot(keep[g]);
      pre[g] = total;
      total += __popcll(bal[g]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < N; ++g) {
      keep[g] = false;
      nk[g] = 0;
      aux[g] = 0.f;
      if (on[g]) {
        double nc;
        keep[g] = ScoreArc<EPS>(c, ar[g], static_cast<uint32_t>(aidx[g]), cst[g], seen, nc, aux[g]);
        nk[g] = CK<M>(nc);
      }
      bal[g] = __ballot(keep[g]);
      pre[g] = total;
      total += __popcll(bal[g]);
    }
  }
  if (total == 0) return;
  int base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(&s.q_n, total);
  base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
  for (int g = 0; g < N; ++g) {
    if (!keep[g]) continue;
    const int pos = base + pre[g] +
                    __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bal[g] >> 32),
                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bal[g]), 0));
    if (pos < c.q_cap) {
      c.q_h[pos] = ar[g].next;
      c.q_arc[pos] = aidx[g];
      c.q_src[pos] = src[g];
      c.q_nk[pos] = nk[g];
      if (c.lattice || c.simple) c.q_ac[pos] = aux[g];
      if (c.lattice) c.q_gr[pos] = ar[g].weight;
    }
    if constexpr (!EPS) {
      if (nk[g] < c.run_min) c.run_min = nk[g];
    }
  }
}

// High-degree tokens queued by a filter pass: a wave takes kGroup of them at a
// time, lanes over consecutive arcs; the first 64 arcs of all kGroup tokens are
// loaded before any is scored so their latencies overlap.  `range(q, abeg, deg)`
// and `source(q, cost, srcid)` describe queue item q.
template <bool EPS, bool T1, int W, int M, class Range, class Source>
__device__ __forceinline__ void BigLoop(const FrameCtx<M> &c, int32_t *next, int nq, Range range,
                                        Source source) {
  static_assert(W == 64 || W == 32, "one token per row, or two of at most 32 arcs");
  constexpr int kPer = 64 / W;
  CParams &p = c.P();
  Shared &s = c.s;
  const int lane = threadIdx.x & 63;
  while (true) {
    int q0 = 0;
    if (lane == 0) q0 = atomicAdd(next, kBigGroup * kPer);
    q0 = __builtin_amdgcn_readfirstlane(q0);
    if (q0 >= nq) break;
    ArcRec ar[kBigGroup];
    int abeg[kBigGroup], deg[kBigGroup], item[kBigGroup];
#pragma unroll
    for (int g = 0; g < kBigGroup; ++g) {
      const int q = q0 + g * kPer + lane / W;
      item[g] = q < nq ? q : q0;  // (entry q0 exists: a slot past the end reads it)
      range(item[g], abeg[g], deg[g]);
      if (q >= nq) deg[g] = 0;
    }
#pragma unroll
    for (int g = 0; g < kBigGroup; ++g)
      ar[g] = LoadArcRecIf(p.arcs, static_cast<uint32_t>(abeg[g] + lane % W), lane % W < deg[g], 0u);
    if constexpr (!T1 && (EPS || M != 2)) {
      uint32_t aidx[kBigGroup], src[kBigGroup];
      bool on[kBigGroup];
      double cst[kBigGroup];
#pragma unroll
      for (int g = 0; g < kBigGroup; ++g) {
        on[g] = lane % W < deg[g];
        aidx[g] = static_cast<uint32_t>(abeg[g] + lane % W);
        source(item[g], cst[g], src[g]);
      }
      ScoreBatchBig<EPS>(c, ar, aidx, on, cst, src);
    } else {
#pragma unroll
      for (int g = 0; g < kBigGroup; ++g)
        if (lane % W < deg[g]) {
          double cst;
          uint32_t srcid;
          source(item[g], cst, srcid);
          FilterArc<EPS, T1>(c, ar[g], static_cast<uint32_t>(abeg[g] + lane % W), cst, srcid);
        }
    }
    if constexpr (!EPS) PublishMin(c);
    if constexpr (W == 64) {
      // tails of tokens with more than 64 arcs, four wave-rows in flight
#pragma unroll 1
      for (int g = 0; g < kBigGroup; ++g) {
        if (deg[g] <= 64) continue;
        double cst;
        uint32_t srcid;
        source(item[g], cst, srcid);
        for (int r0 = 64; r0 < deg[g]; r0 += 256) {  // (wave-uniform: every lane takes every round)
          const int a0 = r0 + lane;
          ArcRec tr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            tr[u] = LoadArcRecIf(p.arcs, static_cast<uint32_t>(abeg[g] + a0 + 64 * u),
                                 a0 + 64 * u < deg[g], static_cast<uint32_t>(abeg[g]));
          if constexpr (!T1 && (EPS || M != 2)) {
            uint32_t aidx[4], src[4];
            bool on[4];
            double cs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              on[u] = a0 + 64 * u < deg[g];
              aidx[u] = static_cast<uint32_t>(abeg[g] + a0 + 64 * u);
              cs[u] = cst;
              src[u] = srcid;
            }
            ScoreBatchBig<EPS>(c, tr, aidx, on, cs, src);
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (a0 + 64 * u < deg[g])
                FilterArc<EPS, T1>(c, tr[u], static_cast<uint32_t>(abeg[g] + a0 + 64 * u), cst, srcid);
          }
        }
      }
      if constexpr (!EPS) PublishMin(c);
    }
  }
}

// Filter phase over the input frontier (EPS=false) or over the tokens on
// worklist `wl_in` (EPS=true).  Touches the table only to read the worklist
// tokens' state and cost.
template <bool LDS, bool EPS, int M>
__device__ __forceinline__ void Filter(const FrameCtx<M> &c, const Table<LDS> &t, int n_items,
                                       const KD_G uint32_t *wl_in, Counters &cnt) {
  constexpr int KD_BUCKET = LDS ? kBucketBig : kBucketSpill;
  (void)KD_BUCKET;
  CParams &p = c.P();
  Shared &s = c.s;
  const int tid = threadIdx.x, lane = tid & 63;
  int par = 0;
  // Tokens with more than kSmallDeg arcs are queued in LDS for whole waves: those of
  // more than 32 arcs from the front of the it_* arrays (a 64-lane row each), the
  // others from the back (two per row).  Counters and cursors of the second list,
  // by chunk parity: (it_n32, it_next32) and (it_n16, it_next16).
  if (tid == 0) { s.it_count[0] = 0; s.it_next[0] = 0; s.it_n32 = 0; s.it_next32 = 0; s.idx_n = 0; }
  // The emitting filter's chain per chunk of tokens is token fields -> state record -> arcs,
  // three dependent round trips with the workgroup's barriers between chunks.  It is
  // software-pipelined: a chunk's arcs are requested while the NEXT chunk's state records
  // and the one after's token fields are already in flight (loads return in order: the wait
  // for the arcs covers them), so a chunk costs one round trip, not three.  (An index past
  // the frontier reads record 0, which exists.)
  // label index (see the tier-1 frames' second pass): the row's two lowest acoustic costs
  [[maybe_unused]] float t_m0 = 0.0f, t_m1 = 0.0f;
  [[maybe_unused]] int t_l0 = 0;
  [[maybe_unused]] bool idx_on = false;
  if constexpr (!EPS && M != 2 && KD_LABEL_INDEX != 0) {
    idx_on = c.ll_in_lds && c.adaptive_beam > 0.0f && p.emit_w_min == p.emit_w_min && Uni(s.top_valid) != 0;
    if (idx_on) LoadRowTop(s, t_m0, t_m1, t_l0);
  }
  // (the two epsilon worklists -- c.q_cap words, unused until the inserts -- as 16-byte notes)
  [[maybe_unused]] const int idx_cap = KD_IDX_NOTE_CAP > 0 ? KD_IDX_NOTE_CAP : c.q_cap / 4;
  constexpr int kIdxDegBits = 11;
  static_assert(kIdxMaxDeg <= (1 << kIdxDegBits), "a note's degree field");
  [[maybe_unused]] uint32_t pf_st = 0, pf_st2 = 0;
  [[maybe_unused]] double pf_cst = 0.0, pf_cst2 = 0.0;
  [[maybe_unused]] bool pf_ok = false;
  [[maybe_unused]] StateRec pf_r{};
  if constexpr (!EPS) {
    const long long i0 = tid < n_items ? c.base_in + tid : 0ll;
    const long long i1 = tid + kBlock < n_items ? c.base_in + tid + kBlock : 0ll;
    pf_st = static_cast<uint32_t>(ToG(p.tok_state)[i0]);
    pf_cst = ToG(p.tok_cost)[i0];
    pf_st2 = static_cast<uint32_t>(ToG(p.tok_state)[i1]);
    pf_cst2 = ToG(p.tok_cost)[i1];
    pf_ok = tid < n_items && pf_cst < c.weight_cutoff;  // faster-decoder.cc:202
    pf_r = LoadState(p.states, pf_ok ? pf_st : 0u);
  }
  LdsBarrier();
  for (int chunk = 0; chunk < n_items; chunk += kBlock, par ^= 1) {
    const int i = chunk + tid;
    [[maybe_unused]] uint32_t cur_st = 0;
    [[maybe_unused]] double cur_cst = 0.0;
    [[maybe_unused]] bool cur_ok = false;
    [[maybe_unused]] StateRec cur_r{};
    if constexpr (!EPS) {
      cur_st = pf_st; cur_cst = pf_cst; cur_ok = pf_ok; cur_r = pf_r;
      pf_st = pf_st2; pf_cst = pf_cst2;
      pf_ok = i + kBlock < n_items && pf_cst < c.weight_cutoff;
      pf_r = LoadState(p.states, pf_ok ? pf_st : 0u);
      const long long i2 = i + 2 * kBlock < n_items ? c.base_in + i + 2 * kBlock : 0ll;
      pf_st2 = static_cast<uint32_t>(ToG(p.tok_state)[i2]);
      pf_cst2 = ToG(p.tok_cost)[i2];
    }
    // (SimpleDecoder's emitting arcs are scored in two passes: FilterArc, not the batch)
    constexpr bool kBatchLane = EPS || M != 2;
    [[maybe_unused]] bool lane_on = false;
    [[maybe_unused]] int lane_abeg = 0, lane_deg = 0;
    [[maybe_unused]] double lane_cst = 0.0;
    [[maybe_unused]] uint32_t lane_src = 0;
    if (i < n_items) {
      uint32_t st, srcid;
      double cst;
      bool ok;
      if constexpr (!EPS) {
        st = cur_st;
        cst = cur_cst;
        srcid = static_cast<uint32_t>(i);
        ok = cur_ok;
      } else {
        const uint32_t hs = wl_in[i];
        if constexpr (LDS) {
          st = t.LoadKey(hs) & kStateMask;
          cst = KC<M>(t.template LoadCostM<M>(hs));
        } else {
          const auto sv = t.LoadSlot(hs);
          st = t.template SlotKey<M>(sv) & kStateMask;
          cst = KC<M>(t.template SlotCost<M>(sv));
        }
        srcid = hs | kSrcEps;
        ok = !(cst > c.cutoff);  // faster-decoder.cc:74
      }
      if (ok) {
        StateRec r;
        if constexpr (!EPS) r = cur_r;
        else r = LoadState(p.states, st);
        const int abeg = EPS ? r.emit_end : r.arc_begin;
        const int deg = EPS ? r.arc_end - r.emit_end : r.emit_end - r.arc_begin;
        if (EPS) { cnt.eps_exp += 1; cnt.eps_arcs += deg; }
        else { cnt.tokens_expanded += 1; cnt.arcs_scanned += deg; }
        bool indexed = false;
        if constexpr (!EPS && M != 2 && KD_LABEL_INDEX != 0) {
          if (idx_on && deg > kSmallDeg && deg < kIdxMaxDeg) {
            unsigned long long seen = s.min_new;
            seen = c.run_min < seen ? c.run_min : seen;
            bool others_out, best_in;
            if constexpr (M == 1) {
              const float cur = static_cast<float>(cst), lim = KCf<M>(seen) + c.adaptive_beam;
              others_out = (cur + t_m1) + p.emit_w_min >= lim;
              best_in = (cur + t_m0) + p.emit_w_min < lim;
            } else {
              const double lim = KC<M>(seen) + static_cast<double>(c.adaptive_beam);
              others_out = (static_cast<double>(p.emit_w_min) + cst) + static_cast<double>(t_m1) >= lim;
              best_in = (static_cast<double>(p.emit_w_min) + cst) + static_cast<double>(t_m0) < lim;
            }
            if (others_out) {
              indexed = true;
              // The lookup itself waits for the pass behind the chunk loop (below): done here,
              // chunk after chunk, its three dependent node loads and the arc's fetch were the
              // chunk's critical path -- 768 tokens at a time, one lookup a lane.  The token is
              // noted in the (idle) epsilon worklists: {first arc, degree | frontier index, cost}.
              int at = best_in ? AggInc(&s.idx_n) : -1;
              if (best_in && at < idx_cap) {
                const unsigned long long cb = static_cast<unsigned long long>(__double_as_longlong(cst));
                reinterpret_cast<KD_G IdxNodeVec *>(c.wl0)[at] =
                    IdxNodeVec{static_cast<uint32_t>(abeg), static_cast<uint32_t>(deg) | srcid << kIdxDegBits,
                               static_cast<uint32_t>(cb), static_cast<uint32_t>(cb >> 32)};
              } else {
                int probes = 0, fetched = 0;
                if (best_in)  // (the list is full: here and now, as until round 5)
                  IndexedArcs(p, abeg, deg, static_cast<uint32_t>(t_l0 + 1), probes, fetched,
                              [&](uint32_t aidx, const ArcRec &ar) { FilterArc<EPS>(c, ar, aidx, cst, srcid); });
                cnt.arcs_skipped += deg - fetched;
                cnt.idx_probes += probes;
              }
            }
          }
        }
        if (indexed) {
        } else if (deg > kSmallDeg) {
          const int q = deg > 32 ? AggInc(&s.it_count[par])
                                 : kT1Front - 1 - AggInc(par ? &s.it_n16 : &s.it_n32);
          s.it_abeg[q] = abeg;
          s.it_deg[q] = deg;
          s.it_cost[q] = cst;
          s.it_src[q] = srcid;
        } else if (kBatchLane) {
          lane_on = true;
          lane_abeg = abeg;
          lane_deg = deg;
          lane_cst = cst;
          lane_src = srcid;
        } else {
          // all of this token's arcs in flight before the first is scored
          ArcRec ar[kSmallDeg];
#pragma unroll
          for (int a = 0; a < kSmallDeg; ++a)
            ar[a] = LoadArcRecIf(p.arcs, static_cast<uint32_t>(abeg + a), a < deg, 0u);
#pragma unroll
          for (int a = 0; a < kSmallDeg; ++a)
            if (a < deg) FilterArc<EPS>(c, ar[a], static_cast<uint32_t>(abeg + a), cst, srcid);
        }
      }
    }
    if constexpr (kBatchLane) {
      // Tokens of at most kSmallDeg arcs (nine states in ten of an HLG): their arcs are scored
      // where the wave is converged, as one batch -- log-likelihood reads grouped, ONE queue
      // allocation for the wave instead of one wave-aggregated allocation per kept arc.
      if (__ballot(lane_on)) {
        ArcRec ar[kSmallDeg];
        uint32_t aidx[kSmallDeg], sr[kSmallDeg];
        bool on[kSmallDeg];
        double cs[kSmallDeg];
#pragma unroll
        for (int a = 0; a < kSmallDeg; ++a) {
          on[a] = lane_on && a < lane_deg;
          aidx[a] = static_cast<uint32_t>(lane_abeg + a);
          ar[a] = LoadArcRecIf(p.arcs, aidx[a], on[a], 0u);
          cs[a] = lane_cst;
          sr[a] = lane_src;
        }
        ScoreBatchBig<EPS>(c, ar, aidx, on, cs, sr);
      }
    }
    if constexpr (!EPS) PublishMin(c);
    LdsBarrier();  // the big-token queue is LDS; queue stores keep draining
    KD_LAP(s, EPS ? 7 : 3);
    const int nq = s.it_count[par], nq2 = par ? s.it_n16 : s.it_n32;
    if (tid == 0) {
      s.it_count[par ^ 1] = 0;
      s.it_next[par ^ 1] = 0;
      if (par) { s.it_n32 = 0; s.it_next32 = 0; } else { s.it_n16 = 0; s.it_next16 = 0; }
    }
    if (nq > 0)
      BigLoop<EPS, false, 64>(c, &s.it_next[par], nq, [&](int q, int &abeg, int &deg) {
        abeg = s.it_abeg[q];
        deg = s.it_deg[q];
      }, [&](int q, double &cst, uint32_t &srcid) {
        cst = s.it_cost[q];
        srcid = s.it_src[q];
      });
    if (nq2 > 0)
      BigLoop<EPS, false, 32>(c, par ? &s.it_next16 : &s.it_next32, nq2, [&](int q, int &abeg, int &deg) {
        abeg = s.it_abeg[kT1Front - 1 - q];
        deg = s.it_deg[kT1Front - 1 - q];
      }, [&](int q, double &cst, uint32_t &srcid) {
        cst = s.it_cost[kT1Front - 1 - q];
        srcid = s.it_src[kT1Front - 1 - q];
      });
    if (chunk + kBlock < n_items) LdsBarrier();  // it_* arrays are reused by the next chunk
    KD_LAP(s, EPS ? 8 : 4);
  }
  if constexpr (!EPS && M != 2 && KD_LABEL_INDEX != 0) {
    if (idx_on) {
      // The label-index pass over the tokens the chunks noted: four per lane in flight, the
      // descents level by level for all four (one dependent round trip per LEVEL for up to 3072
      // tokens, not per level per chunk of 768), then the found arcs' fetches together.
      __syncthreads();  // (the notes are global memory; idx_n is final)
      const int n_noted = s.idx_n < idx_cap ? s.idx_n : idx_cap;
      constexpr int U = KD_IDX_NOTE_ILP;
      const uint32_t il = static_cast<uint32_t>(t_l0 + 1);
      const KD_G IdxNodeVec *notes = reinterpret_cast<const KD_G IdxNodeVec *>(c.wl0);
      for (int e0 = tid; e0 < n_noted; e0 += U * kBlock) {
        int abeg[U], deg[U], j[U];
        uint32_t src[U], cand[U];
        double cst[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + u * kBlock;
          on[u] = e < n_noted;
          const IdxNodeVec r = notes[on[u] ? e : e0];
          abeg[u] = static_cast<int>(r.x);
          deg[u] = on[u] ? static_cast<int>(r.y & ((1u << kIdxDegBits) - 1u)) : 0;
          src[u] = r.y >> kIdxDegBits;
          cst[u] = __longlong_as_double(static_cast<long long>(static_cast<unsigned long long>(r.w) << 32 | r.z));
          j[u] = 0;
          cand[u] = kIdxNone;
        }
        const uint32_t target = IdxKey(il, 0);
        int probes = 0;
        while (true) {
          bool act[U], any = false;
          IdxNodeVec nd[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            act[u] = 4 * j[u] < deg[u];
            any |= act[u];
            nd[u] = *reinterpret_cast<const KD_G IdxNodeMem *>(ToG(p.lab_idx) + abeg[u] + (act[u] ? 4 * j[u] : 0));
          }
          if (!__ballot(any)) break;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!act[u]) continue;
            ++probes;
            const int have = deg[u] - 4 * j[u];
            const uint32_t k0 = nd[u].x, k1 = have > 1 ? nd[u].y : kIdxNone, k2 = have > 2 ? nd[u].z : kIdxNone,
                           k3 = have > 3 ? nd[u].w : kIdxNone;
            const int cc = (k0 < target) + (k1 < target) + (k2 < target) + (k3 < target);
            const uint32_t kc = cc == 0 ? k0 : cc == 1 ? k1 : cc == 2 ? k2 : cc == 3 ? k3 : kIdxNone;
            if (kc != kIdxNone) cand[u] = kc;
            j[u] = 5 * j[u] + 1 + cc;
          }
        }
        // the found arcs, requested together
        bool hit[U];
        ArcRec ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          hit[u] = on[u] && cand[u] != kIdxNone && IdxKeyLabel(cand[u]) == il;
          ar[u] = LoadArcRecIf(p.arcs, static_cast<uint32_t>(abeg[u]) + IdxKeyOffset(hit[u] ? cand[u] : 0u), hit[u], 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!on[u]) continue;
          int fetched = 0;
          if (hit[u]) {
            FilterArc<EPS>(c, ar[u], static_cast<uint32_t>(abeg[u]) + IdxKeyOffset(cand[u]), cst[u], src[u]);
            fetched = 1;
            if (IdxKeyMore(cand[u]))  // (a state with several arcs of the label: the rest, one by one)
              IndexedArcsFrom(p, abeg[u], deg[u], il, IndexFind(p, abeg[u], deg[u], cand[u] + 1, probes), probes,
                              fetched, [&](uint32_t aidx, const ArcRec &a2) { FilterArc<EPS>(c, a2, aidx, cst[u], src[u]); });
          }
          cnt.arcs_skipped += deg[u] - fetched;
        }
        cnt.idx_probes += probes;
      }
      PublishMin(c);
    }
  }
  __syncthreads();  // candidate-queue stores visible to the insert phase
}

// Insert phase: queue entries [q_begin, q_end), one per lane.  Relaxes the
// destination's slot; the entry's destination word is replaced by the slot (or
// kKeyEmpty if the entry lost to the final cutoff) for the resolve passes.
template <bool LDS, bool EPS, int M>
__device__ __forceinline__ void Insert(const FrameCtx<M> &c, const Table<LDS> &t, int q_begin,
                                       int q_end, unsigned long long cutkey,
                                       KD_G uint32_t *wl_out, int wl_out_idx) {
  Shared &s = c.s;
  for (int i = q_begin + threadIdx.x; i < q_end; i += kBlock) {
    // A retry in a larger table replays emitting entries whose destination word the
    // aborted attempt already replaced by a slot: take the state from the arc again.
    const uint32_t k = (!EPS && c.replay) ? LoadArcRec(c.P().arcs, c.q_arc[i]).next : c.q_h[i];
    const unsigned long long nk = c.q_nk[i];
    uint32_t h = kKeyEmpty;
    // emitting entries were queued under the running cutoff: apply the final one (:211);
    // once the frame is known not to fit, stop probing a full table
    if ((EPS || nk < cutkey) && !s.overflow) {
      unsigned long long old = 0;
      if constexpr (!LDS && M == 1) h = t.ClaimRelaxLattice(k, nk, old);
      else h = t.Claim(k);
      if (h == kKeyEmpty) {
        s.overflow = 1;
      } else {
        if constexpr (LDS || M != 1) old = atomicMin(t.CostP(h), nk);
        if (!EPS && c.simple && nk < s.min_stored) atomicMin(&s.min_stored, nk);
        bool push = false;
        if (old == kCostDead) {  // the state's first token this frame
          const int pos = AggInc(&s.n_live);
          if (static_cast<uint32_t>(pos) < t.FillLimit()) c.live[pos] = h;
          else s.overflow = 1;
          // Global table, lattice decoders (no back-pointers: the arc word is free): the
          // token's place in the frame's block is its place on the live list -- noted in
          // the slot now, next to the words this insert just touched, instead of by a
          // scattered agent-scope store per token in the compaction pass.
          if constexpr (!LDS && M == 1) t.StorePlace(h, static_cast<uint32_t>(pos));
          push = (k & kHasEps) != 0;  // every new token is expanded once (:63-65)
          if constexpr (EPS) {
            // a concurrent improver of this brand-new slot may already have queued it:
            // whoever sets the bit first queues (a slot on the worklist twice would be
            // expanded twice and, in the lattice decoders, emit its links twice)
            if (push) {
              const uint32_t bit = 1u << (h & 31);
              push = !(atomicOr(&t.bitmask[h >> 5], bit) & bit);
            }
          }
        } else if (EPS && old > nk && (k & kHasEps)) {  // re-queue on improvement, :97-113
          const uint32_t bit = 1u << (h & 31);
          push = !(atomicOr(&t.bitmask[h >> 5], bit) & bit);
        }
        if (push) wl_out[AggInc(&s.wl_count[wl_out_idx])] = h;
        if (!EPS && c.lattice) AggAdd1(&s.n_emit_links);
      }
    }
    c.q_h[i] = h;
  }
}

// k-th smallest (0-based) of float(cost) over the input frontier: 4-pass
// byte-wise radix select on order-preserving keys (std::nth_element in
// faster-decoder.cc:298-321; tmp_array_ is std::vector<float>, :278).
// Rare (only when max_active / min_active bind).
// Out of line (round 5): inlined, its two copies held 50 of the kernel's scratch reloads and
// their registers in the frame loop of every decode, and it runs only when max_active /
// min_active bind.  (The LDS block is reached through Lds(): a pointer argument would be flat.)
template <bool FRONT_LDS>
__device__ __noinline__ float SelectKth(const double *tok_cost, long long base_in, int n_in, int k) {
  Shared &s = Lds();
  const Team tm{};
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t prefix = 0, pmask = 0;
  int kk = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = tid; b < 256; b += tm.nt) s.hist[b] = 0;
    tm.Sync(false);
    for (int i = tid; i < n_in; i += tm.nt) {
      const uint32_t key =
          FloatKey(static_cast<float>(FRONT_LDS ? s.it_cost[i] : ToG(tok_cost)[base_in + i]));
      if ((key & pmask) == prefix) atomicAdd(&s.hist[(key >> shift) & 255], 1);
    }
    tm.Sync(false);
    if (tid < 64) {
      const int c0 = s.hist[4 * lane], c1 = s.hist[4 * lane + 1], c2 = s.hist[4 * lane + 2],
                c3 = s.hist[4 * lane + 3];
      const int tot = c0 + c1 + c2 + c3;
      const int x = WaveInclusiveSumInt(tot);
      const int excl = x - tot;
      if (kk >= excl && kk < x) {
        int r = kk - excl, b = 0;
        if (r >= c0) { r -= c0; b = 1; if (r >= c1) { r -= c1; b = 2; if (r >= c2) { r -= c2; b = 3; } } }
        s.sel_bin = 4 * lane + b;
        s.sel_k = r;
      }
    }
    tm.Sync(false);
    prefix |= static_cast<uint32_t>(s.sel_bin) << shift;
    pmask |= 255u << shift;
    kk = s.sel_k;
    tm.Sync(false);
  }
  return KeyFloat(prefix);
}

// Best cost and best token of an arena block (used for the first frame of a
// launch; later frames get them from the previous frame's compaction).
template <int M>
__device__ __forceinline__ void FrontierStats(const FrameCtx<M> &c) {
  CParams &p = c.P();
  Shared &s = c.s;
  const int tid = threadIdx.x;
  unsigned long long mk = kCostDead;
  int mi = -1;
  for (int i = tid; i < c.n_in; i += kBlock) {
    const unsigned long long k = CK<M>(ToG(p.tok_cost)[c.base_in + i]);
    if (k < mk) { mk = k; mi = i; }
  }
  const unsigned long long best = BlockMinU64(s, mk);
  if (tid == 0) { s.f_best = best; s.best_deg = 0; s.best_abeg = 0; s.best_cost = 0.0; }
  __syncthreads();
  if (mi >= 0 && mk == best) {
    const StateRec r = LoadState(p.states, static_cast<uint32_t>(ToG(p.tok_state)[c.base_in + mi]));
    s.best_abeg = r.arc_begin;
    s.best_deg = r.emit_end - r.arc_begin;
    s.best_cost = KC<M>(best);
  }
  __syncthreads();
}

// GetCutoff, faster-decoder.cc:244-336, by value.
template <bool FRONT_LDS, int M>
__device__ __forceinline__ void GetCutoff(FrameCtx<M> &c, const Team &tm) {
  CParams &p = c.P();
  Shared &s = c.s;
  const int tid = threadIdx.x;
  const double best_cost = Uni(KC<M>(s.f_best));
  const double beam_cutoff = best_cost + p.beam;
  if (p.max_active == INT_MAX && p.min_active == 0) {  // :249-272
    // LatticeSimpleDecoder has no cutoff on the tokens it expands (:376-401)
    c.weight_cutoff = c.lattice ? __longlong_as_double(0x7FF0000000000000ll) : beam_cutoff;
    c.adaptive_beam = p.beam;
    return;
  }
  if (c.n_in <= p.max_active && c.n_in <= p.min_active) {
    // neither limit can bind (:289, :311-323 with "min_active_cutoff = infinity")
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    c.adaptive_beam = static_cast<float>(inf - best_cost + p.beam_delta);
    c.weight_cutoff = inf;
    return;
  }
  int lt = 0, le = 0;
  for (int i = tid; i < c.n_in; i += tm.nt) {
    const double v =  // narrowed, :278
        static_cast<float>(FRONT_LDS ? s.it_cost[i] : ToG(p.tok_cost)[c.base_in + i]);
    lt += v < beam_cutoff;
    le += v <= beam_cutoff;
  }
  lt = WaveSumInt(lt);
  le = WaveSumInt(le);
  if ((tid & 63) == 0) {
    if (lt) atomicAdd(&s.f_nlt, lt);
    if (le) atomicAdd(&s.f_nle, le);
  }
  tm.Sync(false);
  const int n_lt = Uni(s.f_nlt), n_le = Uni(s.f_nle);
  // tmp_array_[k] < x  <=>  #{v < x} >= k+1 ;  tmp_array_[k] > x  <=>  #{v <= x} <= k
  if (c.n_in > p.max_active && static_cast<long long>(n_lt) >= static_cast<long long>(p.max_active) + 1) {
    const double max_active_cutoff = SelectKth<FRONT_LDS>(p.tok_cost, c.base_in, c.n_in, p.max_active);  // :297-309
    c.adaptive_beam = max_active_cutoff - best_cost + p.beam_delta;
    c.weight_cutoff = max_active_cutoff;
    return;
  }
  // the lattice decoders keep the cutoff in float
  const double beam_cutoff_r = c.lattice ? static_cast<double>(static_cast<float>(beam_cutoff)) : beam_cutoff;
  double min_active_cutoff = beam_cutoff;  // "not looser" unless proven below
  if (c.n_in > p.min_active) {             // :311-323
    if (p.min_active == 0) min_active_cutoff = best_cost;
    else if (n_le <= p.min_active) {
      min_active_cutoff = SelectKth<FRONT_LDS>(p.tok_cost, c.base_in, c.n_in, p.min_active);
    }
  } else {
    min_active_cutoff = __longlong_as_double(0x7FF0000000000000ll);
  }
  if (min_active_cutoff > beam_cutoff) {  // :325-335
    c.adaptive_beam = min_active_cutoff - best_cost + p.beam_delta;
    c.weight_cutoff = min_active_cutoff;
  } else {
    c.adaptive_beam = p.beam;
    c.weight_cutoff = beam_cutoff_r;
  }
}

// Everything of a frame before the table is touched: cutoff, the best token's
// bound, and the emitting filter.  Leaves the candidates in the queue
// [0, s.q_n) and the exact cutoff in c.cutoff.
template <int M>
__device__ __forceinline__ void FrameFront(FrameCtx<M> &c, Counters &cnt) {
  constexpr int KD_BUCKET = kBucketBig;
  (void)KD_BUCKET;
  CParams &p = c.P();
  Shared &s = c.s;
  const int tid = threadIdx.x;
  if (c.simple) {
    // SimpleDecoder expands every token of prev_toks_ with a plain beam
    c.weight_cutoff = __longlong_as_double(0x7FF0000000000000ll);
    c.adaptive_beam = p.beam;
  } else {
    GetCutoff<false>(c, Team{});
  }
  KD_LAP(s, 1);
  // Bound the next cutoff from the best token's arcs first (:176-189); it only
  // shortens the queue, the final cutoff is the same.
  if (!c.simple) {
    unsigned long long mk = kCostInf<M>;
    const int bdeg = s.best_deg, babeg = s.best_abeg;
    const double bcost = s.best_cost;
    for (int a = tid; a < bdeg; a += kBlock) {
      const ArcRec arc = LoadArcRec(p.arcs, static_cast<uint32_t>(babeg + a));
      const float ll = c.ll_in_lds ? s.ll[arc.ilabel - 1]
                                   : c.ll_row[FullIlabel(p, arc, static_cast<uint32_t>(babeg + a)) - 1];
      const float ac_cost = -1 * ll;
      const unsigned long long k =
          c.lattice ? CK<M>(static_cast<double>(static_cast<float>(bcost) + ac_cost + arc.weight))
                    : CK<M>(arc.weight + bcost + ac_cost);
      mk = k < mk ? k : mk;
    }
    mk = WaveMinU64(mk);
    if ((tid & 63) == 0 && mk != kCostInf<M>) atomicMin(&s.min_new, mk);
  }
  KD_LAP(s, 2);
  Table<true> none{s, nullptr, nullptr, 0, 0};
  // SimpleDecoder: a first pass only finds min total cost, the second queues what
  // is below min + beam (simple-decoder.cc:170-176 at the cutoff's final value)
#pragma unroll 1
  for (int pass = c.simple ? 0 : 1; pass < 3; ++pass) {
    c.min_only = pass == 0;
    if (pass == 1 && c.simple) {
      c.cutoff = KC<M>(s.min_new) + static_cast<double>(p.beam);
      cnt = Counters();
    }
    if (pass == 2) {
      // The queue overflowed under the running cutoff.  The minimum is final now,
      // so one more scan queues exactly the arcs below the final cutoff.
      if (c.simple || s.q_n <= c.q_cap) break;
      __syncthreads();
      if (tid == 0) s.q_n = 0;
      cnt = Counters();
      __syncthreads();
    }
    Filter<true, false>(c, none, c.n_in, nullptr, cnt);
  }
  if (c.simple) return;
  {  // (the label index's counters of this frame: straight into the launch totals)
    const int w0 = WaveSumInt(cnt.arcs_skipped), w1 = WaveSumInt(cnt.idx_probes);
    if ((tid & 63) == 0 && (w0 | w1)) {
      atomicAdd(&s.ustat[KD_STAT_ARCS_SKIPPED], static_cast<unsigned long long>(w0));
      atomicAdd(&s.ustat[KD_STAT_INDEX_PROBES], static_cast<unsigned long long>(w1));
    }
  }
  if (c.lattice)  // float, lattice-simple-decoder.cc:388-390
    c.cutoff = KCf<M>(s.min_new) + c.adaptive_beam;
  else
    c.cutoff = KC<M>(s.min_new) + static_cast<double>(c.adaptive_beam);  // value of :240
}

template <bool LDS>
__device__ __forceinline__ void FullClear(const Table<LDS> &t, Shared &s) {
  // only the LDS table is ever cleared wholesale (launch start, aborted frames)
  const int tid = threadIdx.x;
  for (int h = tid; h < kSlots; h += kBlock) {
    s.key[h] = kKeyEmpty;
    s.cost[h] = kCostDead;
    s.arc[h] = kArcNone;
  }
  __syncthreads();
}

// Clears the slots on the live list (and any claimed-but-unlisted slot is
// impossible: a slot is listed by whoever gives it its first cost, and every
// claimer relaxes the slot right after claiming it).
template <bool LDS, int M>
__device__ __forceinline__ void ClearLive(const FrameCtx<M> &c, const Table<LDS> &t, int n_live) {
  for (int i = threadIdx.x; i < n_live; i += kBlock) {
    t.ClearSlot(c.live[i]);
  }
}

// The table half of a frame (frame >= 0) or of InitDecoding (frame == -1):
// insert the emitting candidates queue[0, q_emit), run the epsilon closure,
// resolve back-pointers and compact.  Returns 0 on success, 1 if the table,
// the live list or the queue overflowed (nothing committed; table dirty), or a
// KD_UTT_* bit on a fatal capacity error.
template <bool LDS, int M>
__device__ __forceinline__ uint32_t FrameBack(FrameCtx<M> &c, const Table<LDS> &t, int frame,
                                              int q_emit, Counters &cnt) {
  constexpr int KD_BUCKET = LDS ? kBucketBig : kBucketSpill;
  (void)KD_BUCKET;
  CParams &p = c.P();
  Shared &s = c.s;
  const int tid = threadIdx.x;
  // simple mode queued exactly what is below the cutoff (on the total cost)
  const unsigned long long cutkey = c.simple ? kCostDead : CK<M>(c.cutoff);

  if (frame < 0) {
    if (tid == 0) {  // faster-decoder.cc:49-51: dummy arc into the start state, cost 0
      const StateRec r = LoadState(p.states, p.start);
      const bool eps = r.arc_end > r.emit_end;
      const uint32_t h = t.Hash(static_cast<uint32_t>(p.start));
      if constexpr (!LDS && M == 1) {
        unsigned long long was;
        t.ClaimRelaxLattice(static_cast<uint32_t>(p.start) | (eps ? kHasEps : 0u), CK<M>(0.0), was);
        t.StorePlace(h, 0u);  // its place in the block (see Insert)
      } else {
        t.StoreKey(h, static_cast<uint32_t>(p.start) | (eps ? kHasEps : 0u));
        t.StoreCost(h, CK<M>(0.0));
      }
      c.live[0] = h;
      s.n_live = 1;
      if (eps) { c.wl0[0] = h; s.wl_count[0] = 1; }
    }
  } else {
    Insert<LDS, false>(c, t, 0, q_emit, cutkey, c.wl0, 0);
  }
  __syncthreads();
  KD_LAP(s, 5);
  if (s.overflow) return kRetry;
  if (c.simple)  // ProcessNonemitting: best stored cost + beam (simple-decoder.cc:199-205)
    c.cutoff = (frame < 0 ? 0.0 : KC<M>(s.min_stored)) + static_cast<double>(p.beam);

  // ProcessNonemitting to a fixpoint (:59-119)
  int cur = 0, q_begin = q_emit;
  KD_G uint32_t *wl_a = c.wl0, *wl_b = c.wl1;
  while (true) {
    const int n_items = s.wl_count[cur];
    if (n_items == 0) break;
    const int words = LDS ? kSlots / 32 : (1 << (t.log2_slots - 5));
    for (int w = tid; w < words; w += kBlock) t.bitmask[w] = 0;
    if (tid == 0) s.wl_count[cur ^ 1] = 0;
    TableBarrier<LDS>();
    KD_LAP(s, 6);
    Filter<LDS, true>(c, t, n_items, wl_a, cnt);
    const int q_end = s.q_n;
    if (q_end > c.q_cap) return kRetry;
    Insert<LDS, true>(c, t, q_begin, q_end, 0, wl_b, cur ^ 1);
    __syncthreads();
    KD_LAP(s, 9);
    if (s.overflow) return kRetry;
    q_begin = q_end;
    cur ^= 1;
    KD_G uint32_t *tmp = wl_a;
    wl_a = wl_b;
    wl_b = tmp;
  }

  // Resolve R1: entries that hold their slot's final cost compete for the lowest
  // arc index.  This lane's first entries stay in registers for R2.
  const int n_q = q_begin;
  const int n_out = s.n_live;
  constexpr int kQRegs = 4;
  uint32_t lg_h[kQRegs], lg_arc[kQRegs];
#pragma unroll
  for (int u = 0; u < kQRegs; ++u) {
    lg_h[u] = kKeyEmpty;
    lg_arc[u] = 0;
  }
  if constexpr (M != 1) {  // (lattice tokens keep forward links, not back-pointers)
#pragma unroll
    for (int u = 0; u < kQRegs; ++u) {
      const int i = u * kBlock + tid;
      if (i < n_q) {
        const uint32_t h = c.q_h[i];
        if (h != kKeyEmpty && t.template LoadCostM<M>(h) == c.q_nk[i]) {
          const uint32_t aidx = c.q_arc[i];
          atomicMin(t.ArcP(h), aidx);
          lg_h[u] = h;
          lg_arc[u] = aidx;
        }
      }
    }
    for (int i = kQRegs * kBlock + tid; i < n_q; i += kBlock) {
      const uint32_t h = c.q_h[i];
      if (h != kKeyEmpty && t.template LoadCostM<M>(h) == c.q_nk[i]) atomicMin(t.ArcP(h), c.q_arc[i]);
    }
  }
  if (tid == 0) {
    // the frame's token block, and room for its forward links (at most one per queue
    // entry below the cutoff), from this utterance's chunks of the pools
    const uint32_t e = TakeChunks(p, s, c.utt, frame + 1, n_out,
                                  c.lattice ? s.n_emit_links + (n_q - q_emit) : 0);
    s.alloc_ok = static_cast<int32_t>(e);
    s.base_out = e ? -1 : s.arena_next;
    if (!e) s.arena_next += n_out;
    s.f_best_next = kCostDead;
    s.best_deg = 0;
    s.best_abeg = 0;
    s.best_cost = 0.0;
  }
  TableBarrier<LDS>();
  KD_LAP(s, 10);
  const long long base_out = s.base_out;
  if (base_out < 0) return static_cast<uint32_t>(s.alloc_ok);
  // SimpleDecoder::PruneToks (simple-decoder.cc:252-281): tokens at or beyond
  // best + beam leave the frontier.  They keep their arena record (tokens that
  // survive may point back at them), placed after the kept ones.
  unsigned long long prune_key = kCostDead;
  if (c.simple) {
    unsigned long long mk = kCostDead;
    for (int i = tid; i < n_out; i += kBlock) {
      const unsigned long long ck = t.template LoadCostM<M>(c.live[i]);
      mk = ck < mk ? ck : mk;
    }
    mk = BlockMinU64(s, mk);
    prune_key = CK<M>(KC<M>(mk) + static_cast<double>(p.beam));
  }
  // C1: live list -> (state, cost); a slot's key word now holds its index in the block.
  unsigned long long my_min = kCostDead;
  uint32_t my_min_state = 0;
  {
    // (kSpillIlp tokens per lane: their live-list entries -- global memory in both big tiers --
    // and then their table words are fetched together)
    constexpr int U = LDS ? 1 : kSpillIlp;
    for (int i0 = tid; i0 < n_out; i0 += U * kBlock) {
      uint32_t h[U], kw[U], aw[U];
      unsigned long long ckv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) h[u] = c.live[i0 + u * kBlock < n_out ? i0 + u * kBlock : 0];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (LDS) {
          ckv[u] = t.template LoadCostM<M>(h[u]);
          kw[u] = t.LoadKey(h[u]);
          aw[u] = M != 1 ? t.LoadArc(h[u]) : 0u;  // (lattice tokens have no back-pointers)
        } else {
          const auto sv = t.LoadSlot(h[u]);
          ckv[u] = t.template SlotCost<M>(sv);
          kw[u] = t.template SlotKey<M>(sv);
          aw[u] = sv.w;  // (read where M != 1)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + u * kBlock >= n_out) continue;
        const unsigned long long ck = ckv[u];
        int i = i0 + u * kBlock;
        if (c.simple)
          i = ck < prune_key ? AggInc(&s.n_act) : n_out - 1 - AggInc(&s.n_inact);
        const long long g = base_out + i;
        const uint32_t st_of_tok = kw[u] & kStateMask;
        ToG(p.tok_state)[g] = static_cast<int32_t>(st_of_tok);
        ToG(p.tok_cost)[g] = KC<M>(ck);
        if (c.lattice) ToG(p.tok_extra)[g] = 0.0f;  // lattice-simple-decoder.cc:91-95
        if (M != 1 && aw[u] == kArcNone) {  // the start token
          ToG(p.tok_prev)[g] = -1;
          ToG(p.tok_arc)[g] = -1;
        }
        // (global table, lattice: the slot's arc word holds i since the insert)
        if constexpr (LDS || M != 1) t.StoreKey(h[u], static_cast<uint32_t>(i));
        if (ck < my_min) { my_min = ck; my_min_state = st_of_tok; }
      }
    }
    const unsigned long long wmin = WaveMinU64(my_min);
    if ((tid & 63) == 0 && wmin != kCostDead) atomicMin(&s.f_best_next, wmin);
  }
  // the token records just stored are next read by the next frame's filter, behind
  // the full barrier that follows the log-likelihood row load
  TableBarrier<LDS>();
  KD_LAP(s, 11);
  const int n_front = c.simple ? s.n_act : n_out;  // tokens the next frame starts from
  // R2: the winning entry of each slot writes the back-pointer.
  {
    auto write_bp = [&](uint32_t h, uint32_t aidx, int i) {
      const long long g = base_out + t.LoadKey(h);
      const uint32_t sr = c.q_src[i];
      const long long prev = (sr & kSrcEps) ? base_out + t.LoadKey(sr & ~kSrcEps) : c.base_in + sr;
      ToG(p.tok_prev)[g] = static_cast<int32_t>(prev);
      ToG(p.tok_arc)[g] = static_cast<int32_t>(aidx);
      if (c.simple) ToG(p.tok_extra)[g] = c.q_ac[i];  // the arc's stored acoustic cost
    };
    if constexpr (M != 1) {
#pragma unroll
      for (int u = 0; u < kQRegs; ++u)
        if (lg_h[u] != kKeyEmpty && t.LoadArc(lg_h[u]) == lg_arc[u])
          write_bp(lg_h[u], lg_arc[u], u * kBlock + tid);
      for (int i = kQRegs * kBlock + tid; i < n_q; i += kBlock) {
        const uint32_t h = c.q_h[i];
        if (h == kKeyEmpty) continue;
        const uint32_t aidx = c.q_arc[i];
        if (t.template LoadCostM<M>(h) == c.q_nk[i] && t.LoadArc(h) == aidx) write_bp(h, aidx, i);
      }
    }
  }
  KD_LAP(s, 16);
  // Lattice decoders: every queue entry below the cutoff is a ForwardLink
  // (lattice-simple-decoder.cc:393-398, :179-180).  Emitting links leave the
  // previous frame's tokens; epsilon links stay inside this frame and are kept
  // only if generated from their source token's final cost (:164-165 deletes
  // and regenerates the links of a re-expanded token).
  if (c.lattice) {
    const long long lbase = s.link_next;  // (room was taken with the token block)
    const int n_emit = s.n_emit_links;
    if constexpr (LDS) {
      for (int i = tid; i < n_q; i += kBlock) {
        const uint32_t h = c.q_h[i];
        if (h == kKeyEmpty) continue;
        const uint32_t sr = c.q_src[i];
        long long at, src_tok;
        float ac;
        if (i < q_emit) {
          src_tok = c.base_in + sr;
          ac = c.q_ac[i];
          at = lbase + AggInc(&s.n_emit_written);
        } else {
          const uint32_t hs = sr & ~kSrcEps;
          // source cost at generation vs its final cost (its slot's cost word is intact)
          if (c.q_ac[i] != KCf<M>(t.template LoadCostM<M>(hs))) continue;
          src_tok = base_out + t.LoadKey(hs);
          ac = 0.0f;
          at = lbase + n_emit + AggInc(&s.n_eps_links);
        }
        const uint32_t aidx = c.q_arc[i];
        ToG(p.link_src)[at] = static_cast<int32_t>(src_tok);
        ToG(p.link_dst)[at] = static_cast<int32_t>(base_out + t.LoadKey(h));
        ToG(p.link_arc)[at] = static_cast<int32_t>(aidx);
        ToG(p.link_ac)[at] = ac;
        ToG(p.link_graph)[at] = c.q_gr[i];
        ToG(p.link_delta)[at] = KCf<M>(c.q_nk[i]) -
                                KCf<M>(t.template LoadCostM<M>(h));
      }
    } else {
      // global-table tier: kSpillIlp entries per lane, their queue words and then their
      // table words (destination slot; source slot of an epsilon entry) fetched together
      constexpr int U = kSpillIlp;
      for (int i0 = tid; i0 < n_q; i0 += U * kBlock) {
        uint32_t h[U], sr[U], aidx[U], kd[U], ks[U];
        float qa[U], qg[U];
        unsigned long long nk[U], cd[U], cs[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kBlock, idx = i < n_q ? i : 0;
          h[u] = c.q_h[idx];
          sr[u] = c.q_src[idx];
          aidx[u] = c.q_arc[idx];
          qa[u] = c.q_ac[idx];
          qg[u] = c.q_gr[idx];
          nk[u] = c.q_nk[idx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kBlock;
          ok[u] = i < n_q && h[u] != kKeyEmpty;
          const uint32_t hd = ok[u] ? h[u] : 0u, hs = ok[u] && i >= q_emit ? sr[u] & ~kSrcEps : 0u;
          const auto vd = t.LoadSlot(hd), vs = t.LoadSlot(hs);
          kd[u] = t.SlotPlace(vd);  // (the token's place in the block: Insert)
          cd[u] = t.template SlotCost<M>(vd);
          ks[u] = t.SlotPlace(vs);
          cs[u] = t.template SlotCost<M>(vs);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          const int i = i0 + u * kBlock;
          long long at, src_tok;
          float ac;
          if (i < q_emit) {
            src_tok = c.base_in + sr[u];
            ac = qa[u];
            at = lbase + AggInc(&s.n_emit_written);
          } else {
            // source cost at generation vs its final cost (its slot's cost word is intact)
            if (qa[u] != KCf<M>(cs[u])) continue;
            src_tok = base_out + ks[u];
            ac = 0.0f;
            at = lbase + n_emit + AggInc(&s.n_eps_links);
          }
          ToG(p.link_src)[at] = static_cast<int32_t>(src_tok);
          ToG(p.link_dst)[at] = static_cast<int32_t>(base_out + kd[u]);
          ToG(p.link_arc)[at] = static_cast<int32_t>(aidx[u]);
          ToG(p.link_ac)[at] = ac;
          ToG(p.link_graph)[at] = qg[u];
          ToG(p.link_delta)[at] = KCf<M>(nk[u]) - KCf<M>(cd[u]);
        }
      }
    }
  }
  KD_LAP(s, 17);
  // next frame's best token: fetch its arc range now, off the critical path
  const unsigned long long best = s.f_best_next;
  if (my_min == best && best != kCostDead) {
    const StateRec r = LoadState(p.states, my_min_state);
    s.best_abeg = r.arc_begin;
    s.best_deg = r.emit_end - r.arc_begin;
    s.best_cost = KC<M>(best);
  }
  if (tid == 0) {
    const long long fi = static_cast<long long>(c.utt) * (p.max_frames + 1) + (frame + 1);
    ToG(p.frame_base)[fi] = base_out;
    ToG(p.frame_count)[fi] = n_front;
  }
  KD_LAP(s, 18);
  // fold this frame's counters into the launch totals
  {
    const int w0 = WaveSumInt(cnt.tokens_expanded), w1 = WaveSumInt(cnt.eps_exp),
              w2 = WaveSumInt(cnt.arcs_scanned), w3 = WaveSumInt(cnt.eps_arcs);
    if ((tid & 63) == 0) {
      if (w0) atomicAdd(&s.frame_cnt[0], w0);
      if (w1) atomicAdd(&s.frame_cnt[1], w1);
      if (w2) atomicAdd(&s.frame_cnt[2], w2);
      if (w3) atomicAdd(&s.frame_cnt[3], w3);
    }
  }
  // back-pointer and link stores are only read by later kernels: let them drain
  TableBarrier<LDS>();
  KD_LAP(s, 12);
  ClearLive<LDS>(c, t, n_out);
  if (tid == 0) {
    s.stat[KD_STAT_TOKENS_EXPANDED] += s.frame_cnt[0];
    s.stat[KD_STAT_EPS_EXPANSIONS] += s.frame_cnt[1];
    s.stat[KD_STAT_ARCS_SCANNED] += s.frame_cnt[2];
    s.stat[KD_STAT_EPS_ARCS_SCANNED] += s.frame_cnt[3];
    s.stat[KD_STAT_ARCS_KEPT] += q_emit;
    s.stat[KD_STAT_TOKENS_OUT] += n_out;
    if (n_out > s.stat[KD_STAT_MAX_TOKENS]) s.stat[KD_STAT_MAX_TOKENS] = n_out;
    if (frame >= 0) {
      s.stat[KD_STAT_FRAMES] += 1;
      s.stat[KD_STAT_TOKENS_IN] += c.n_in;
    }
    if (!LDS) s.stat[KD_STAT_SPILL_FRAMES] += 1;
    s.f_best = best;
    if (c.lattice) {
      const long long lbase = s.link_next;
      const long long fb = static_cast<long long>(c.utt) * (p.max_frames + 1);
      if (frame >= 0) {
        ToG(p.link_emit_base)[fb + frame] = lbase;
        ToG(p.link_emit_count)[fb + frame] = s.n_emit_links;
      }
      ToG(p.link_eps_base)[fb + frame + 1] = lbase + s.n_emit_links;
      ToG(p.link_eps_count)[fb + frame + 1] = s.n_eps_links;
      s.link_next = lbase + s.n_emit_links + s.n_eps_links;
      s.stat[KD_STAT_LINKS] += s.n_emit_links + s.n_eps_links;
    }
  }
  c.base_in = base_out;
  c.n_in = n_front;
  return 0;
}

// ---- tier 1 ------------------------------------------------------------------
// A frame whose frontier is at most kT1Front tokens.  The frontier is in LDS
// (it_cost / it_abeg / it_deg / it_src, left there by the previous frame's
// compaction), the table is the 4096-slot lower half of the table arrays, the
// first kQL queue entries, the live list and the work items are in the upper
// half, and epsilon worklists are slot bitmasks -- so the only global reads of the
// frame are the arcs (and the state records of tokens with epsilon arcs).
//
// A small frame is a chain of dependent round trips -- a handful to memory, a few dozen
// to LDS -- that most of the workgroup's waves only watch (two thirds of a wave's cycles
// are barrier waits: profiles/r05_search_counters.json); a large one is bound by the
// scattered requests it issues.  What shortens the chain's share of a CU's time is another
// utterance on the same CU: the small geometry (KD_GEOM_SMALL) runs these frames three
// workgroups to a CU.
//
// Same results as FrameFront + FrameBack<true>.  Returns 0, 1 (does not fit:
// the caller replays the queue in the big table), 2 (the queue overflowed: the
// caller runs the whole frame in the big path) or a fatal KD_UTT_* bit; the new
// frontier is described by s.base_out / s.n_live.
template <int M>
__device__ __forceinline__ uint32_t SmallFrame(FrameCtx<M> &c, const Table<true> &t, int frame,
                                               Counters &cnt, const Team &tm,
                                               const KD_G float *next_row, bool have_next) {
  constexpr int KD_BUCKET = kBucketSmall;
  (void)KD_BUCKET;
  Shared &s = c.s;
  // The thread index is made opaque once per frame: otherwise every address the
  // frame derives from it (tid * 4 + array base, tid + kBlock, ...) is hoisted out of
  // the frame loop and held in a register of its own for the whole launch, and the
  // frame's hot loops spill.  Recomputing them where they are used costs a few
  // instructions a frame.
  int tid_opaque = threadIdx.x;
#ifndef KD_NO_OPAQUE_TID
  asm volatile("" : "+v"(tid_opaque));
#endif
  const int tid = tid_opaque, lane = tid & 63, nt = tm.nt;
  const int n_in = c.n_in;
  uint32_t *wlmask = s.bitmask;  // two masks of kT1Slots bits
  constexpr int kMaskWords = kT1Slots / 32;

  // Queue entry i: the first kQL entries are in LDS, the rest in global memory.  The
  // passes over the queue run once per part (split): a loop that picks the location per
  // entry makes hipcc wait for all outstanding vector-memory operations -- the pass's
  // own stores -- before every LDS read behind the branch.
  using InLds = std::true_type;
  using InGlobal = std::false_type;
  auto q_h = [&](auto w, int i) -> uint32_t { if constexpr (decltype(w)::value) return T1QH(s)[i]; else return c.q_h[i]; };
  auto q_nk = [&](auto w, int i) -> unsigned long long { if constexpr (decltype(w)::value) return T1QNk(s)[i]; else return c.q_nk[i]; };
  auto q_arc = [&](auto w, int i) -> uint32_t { if constexpr (decltype(w)::value) return T1QArc(s)[i]; else return c.q_arc[i]; };
  auto q_src = [&](auto w, int i) -> uint32_t { if constexpr (decltype(w)::value) return T1QSrc(s)[i]; else return c.q_src[i]; };
  auto q_ac = [&](auto w, int i) -> float { if constexpr (decltype(w)::value) return T1QAc(s)[i]; else return c.q_ac[i]; };
  auto q_gr = [&](auto w, int i) -> float { if constexpr (decltype(w)::value) return T1QGr(s)[i]; else return c.q_gr[i]; };
  auto split = [&](int b, int e, auto body) {  // body(where, lo, hi) over [b, e)
    const int m = e < kQL ? e : kQL;
    if (b < m) body(InLds{}, b, m);
    if (e > kQL) body(InGlobal{}, b > kQL ? b : kQL, e);
  };
  // queue entries past kQL are in global memory: their stores must have landed
  auto queue_sync = [&]() {
    tm.Sync(false);
    if (s.q_n > kQL) tm.Sync(true);
  };
  // Scores the arcs [abeg, abeg + deg) of one token on this lane, kSmallDeg loads in flight.
  auto lane_arcs = [&](auto eps_tag, int abeg, int deg, double cst, uint32_t srcid) {
    constexpr bool EPS = decltype(eps_tag)::value;
    for (int a0 = 0; a0 < deg; a0 += kSmallDeg) {
      ArcRec ar[kSmallDeg];
#pragma unroll
      for (int a = 0; a < kSmallDeg; ++a)
        ar[a] = LoadArcRecIf(c.P().arcs, static_cast<uint32_t>(abeg + a0 + a), a0 + a < deg, 0u);
#pragma unroll
      for (int a = 0; a < kSmallDeg; ++a)
        if (a0 + a < deg)
          FilterArc<EPS, true>(c, ar[a], static_cast<uint32_t>(abeg + a0 + a), cst, srcid);
    }
  };
  // A token with more than kSmallDeg arcs becomes work items of 64 consecutive arcs:
  // one head entry (its first 64) and, beyond that, tail entries (token, chunk).
  // `ref` indexes the it_* arrays.  Tails that find their list full are scored here.
  auto push_items = [&](auto eps_tag, int ref, int abeg, int deg, double cst, uint32_t srcid) {
    // by width class: 16-lane and 32-lane slots pack several tokens into a wave row
    if (deg <= 16) {
      const int at = AggInc(&s.it_n16);
      if (at < kT1W16) {
        T1W16(s)[at] = static_cast<uint16_t>(ref);
        return;
      }
    }
    if (deg <= 32) {  // (and the 16-lane tokens that found their list full)
      T1Heads(s)[kT1Front - 1 - AggInc(&s.it_n32)] = static_cast<uint16_t>(ref);
      return;
    }
    if (deg <= 48) {
      // 33..48 arcs (the bulk of an HLG frontier: a token inside a word, ~38 arcs): a
      // 32-lane slot for the first 32 and a 16-lane slot (kSecondPart: arcs 32..47) for
      // the rest -- 48 lanes instead of a 64-lane row that would be 40 % idle
      const int at = AggInc(&s.it_n16);
      if (at < kT1W16) {
        T1W16(s)[at] = static_cast<uint16_t>(ref | kSecondPart);
        T1Heads(s)[kT1Front - 1 - AggInc(&s.it_n32)] = static_cast<uint16_t>(ref);
        return;
      }
    }
    T1Heads(s)[AggInc(&s.it_count[0])] = static_cast<uint16_t>(ref);
  };
  // Scores N arcs per lane (`on[g]`: this lane holds arc g) and appends the survivors
  // to the candidate queue with ONE queue allocation for the wave.  Call where the
  // wave is converged.  The running cutoff is read once: a looser running cutoff
  // only lengthens the queue, the exact one is applied at insert.  (Tightening it
  // first with the wave's own minimum -- a 64-bit wave reduction -- costs more than
  // the shorter queue saves: measured.)
  auto score_batch = [&](auto eps_tag, auto n_tag, const ArcRec *ar, const int *aidx, const bool *on,
                         const double *cst, const uint32_t *src) {
    constexpr bool EPS = decltype(eps_tag)::value;
    constexpr int N = decltype(n_tag)::value;
    unsigned long long seen = 0;
    if constexpr (!EPS) {
      seen = s.min_new;
      seen = c.run_min < seen ? c.run_min : seen;
    }
    unsigned long long nk[N], bal[N];
    float aux[N];
    bool keep[N];
    int pre[N], total = 0;
    if constexpr (M == 1 && !EPS) {
      if (c.ll_in_lds) {
        // Lattice decoders, emitting arcs, row in LDS (every HLG / HL / H decode): the rows'
        // log-likelihoods are read TOGETHER and the rows scored without a branch.  Written
        // row by row under `if (on[g])`, each row waits for its own LDS read behind the
        // branch (four exposed LDS round trips an iteration, in the kernel's hottest loop:
        // a quarter of the frames -- frontiers of 512-1023 tokens -- spend 40 % of their
        // cycles here).  The arithmetic is ScoreArc's (lattice-simple-decoder.cc:383-390).
        // (A lane without an arc holds arc 0 -- LoadArcRecIf -- whose ilabel may be 0.)
        float llv[N];
#pragma unroll
        for (int g = 0; g < N; ++g) llv[g] = s.ll[on[g] ? ar[g].ilabel - 1 : 0];
        const float lim = KCf<M>(seen) + c.adaptive_beam;
#pragma unroll
        for (int g = 0; g < N; ++g) {
          const float ac_cost = -llv[g], cur_cost = static_cast<float>(cst[g]),
                      tot_cost = cur_cost + ac_cost + ar[g].weight;
          keep[g] = on[g] && tot_cost < lim;
          nk[g] = CKf<M>(tot_cost);
          aux[g] = ac_cost;
          bal[g] = __ballot(keep[g]);
          pre[g] = total;
          total += __popcll(bal[g]);
        }
      } else {
#pragma unroll
        for (int g = 0; g < N; ++g) {
          keep[g] = false;
          nk[g] = 0;
          aux[g] = 0.f;
          if (on[g]) {
            double nc;
            keep[g] = ScoreArc<EPS>(c, ar[g], static_cast<uint32_t>(aidx[g]), cst[g], seen, nc, aux[g]);
            nk[g] = CK<M>(nc);
          }
          bal[g] = __ballot(keep[g]);
          pre[g] = total;
          total += __popcll(bal[g]);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < N; ++g) {
        keep[g] = false;
        nk[g] = 0;
        aux[g] = 0.f;
        if (on[g]) {
          double nc;
          keep[g] = ScoreArc<EPS>(c, ar[g], static_cast<uint32_t>(aidx[g]), cst[g], seen, nc, aux[g]);
          nk[g] = CK<M>(nc);
        }
        bal[g] = __ballot(keep[g]);
        pre[g] = total;
        total += __popcll(bal[g]);
      }
    }
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&s.q_n, total);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int g = 0; g < N; ++g) {
      if (!keep[g]) continue;
      const int pos = base + pre[g] +
                      __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bal[g] >> 32),
                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bal[g]), 0));
      T1Put(c, pos, ar[g], static_cast<uint32_t>(aidx[g]), src[g], nk[g], aux[g]);
      if constexpr (!EPS) {
        if (nk[g] < c.run_min) c.run_min = nk[g];
      }
    }
  };
  // One pass over a list of work items of width W lanes: a wave takes kGroup rows
  // (kGroup * 64 / W items) at a time.  `item(q, ref, chunk)` describes item q.
  // (One loop per width on purpose: this phase is bound by instruction issue, and a
  // single loop over rows of any width -- fewer rounds, more instructions per row --
  // was measured 25% slower.)
  auto rows = [&](auto eps_tag, auto width_tag, int nq, int32_t *next, auto item_ref) {
    constexpr bool EPS = decltype(eps_tag)::value;
    constexpr int W = decltype(width_tag)::value, kPer = 64 / W;
    while (true) {
      int q0 = 0;
      if (lane == 0) q0 = atomicAdd(next, kGroup * kPer);
      q0 = __builtin_amdgcn_readfirstlane(q0);
      if (q0 >= nq) break;
      ArcRec ar[kGroup];
      int aidx[kGroup], ref[kGroup];
      bool on[kGroup];
      // (no branches: the list reads of the kGroup rows go out together, then their
      // descriptor reads -- written per row under `if (q < nq)`, every LDS read waits
      // for the one before it; entry 0 exists whenever the loop is entered)
      bool valid[kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        const int q = q0 + g * kPer + lane / W;
        valid[g] = q < nq;
        ref[g] = item_ref(valid[g] ? q : 0);
      }
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        int off = lane % W;
        if (W == 16 && (ref[g] & kSecondPart)) off += 32;
        ref[g] &= kSecondPart - 1;
        const int deg = s.it_deg[ref[g]], abeg = s.it_abeg[ref[g]];
        on[g] = valid[g] && off < deg;
        aidx[g] = abeg + off;
        if (!valid[g]) ref[g] = -1;
      }
#pragma unroll
      for (int g = 0; g < kGroup; ++g)
        ar[g] = LoadArcRecIf(c.P().arcs, static_cast<uint32_t>(aidx[g]), on[g], 0u);
      double cst[kGroup];
      uint32_t src[kGroup];
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        const int rr = ref[g] < 0 ? 0 : ref[g];  // (unconditional reads: see LoadArcRecIf)
        cst[g] = s.it_cost[rr];
        src[g] = EPS ? s.it_src[rr] : static_cast<uint32_t>(rr);
      }
      score_batch(eps_tag, std::integral_constant<int, kGroup>{}, ar, aidx, on, cst, src);
      if constexpr (W == 64) {
        // tokens with more than 64 arcs: the rest, four wave-rows in flight
#pragma unroll 1
        for (int g = 0; g < kGroup; ++g) {
          if (ref[g] < 0) continue;
          const int deg = s.it_deg[ref[g]];
          if (deg <= 64) continue;
          const int abeg = s.it_abeg[ref[g]];
          const double cst = s.it_cost[ref[g]];
          const uint32_t srcid = EPS ? s.it_src[ref[g]] : static_cast<uint32_t>(ref[g]);
          for (int a0 = 64 + lane; a0 < deg; a0 += 256) {
            ArcRec tr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              tr[u] = LoadArcRecIf(c.P().arcs, static_cast<uint32_t>(abeg + a0 + 64 * u),
                                   a0 + 64 * u < deg, static_cast<uint32_t>(abeg));
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (a0 + 64 * u < deg)
                FilterArc<EPS, true>(c, tr[u], static_cast<uint32_t>(abeg + a0 + 64 * u), cst, srcid);
          }
        }
      }
      if constexpr (!EPS) PublishMin(c);
    }
  };
  auto item_loop = [&](auto eps_tag) {
    const int n64 = Uni(s.it_count[0]), n32 = Uni(s.it_n32), n16r = Uni(s.it_n16);
    const int n16 = n16r < kT1W16 ? n16r : kT1W16;
    // (an empty list is not entered: finding that out inside costs every wave an LDS
    // round trip, and the epsilon rounds' lists are empty more often than not)
    if (n64 > 0)
      rows(eps_tag, std::integral_constant<int, 64>{}, n64, &s.it_next[0],
           [&](int q) -> int { return T1Heads(s)[q]; });
    if (n32 > 0)
      rows(eps_tag, std::integral_constant<int, 32>{}, n32, &s.it_next32,
           [&](int q) -> int { return T1Heads(s)[kT1Front - 1 - q]; });
    if (n16 > 0)
      rows(eps_tag, std::integral_constant<int, 16>{}, n16, &s.it_next16,
           [&](int q) -> int { return T1W16(s)[q]; });
  };
  using EpsNo = std::false_type;
  using EpsYes = std::true_type;

#ifdef KD_ROW_EARLY  // (A/B knob: the request where it was until round 3)
  float ll_next[kPfRegs];
#pragma unroll
  for (int k = 0; k < kPfRegs; ++k) {
    const int v = tid + k * kBlock;
    ll_next[k] = have_next && v < c.P().num_cols ? next_row[v] : 0.0f;
  }
#endif
  for (int w = tid; w < kMaskWords; w += nt) wlmask[w] = 0;  // worklist 0 collects the emitting insert's tokens
  c.run_min = kCostInf<M>;
  GetCutoff<true>(c, tm);
  KD_LAP(s, 1);
  KD_PHASE(c);
  // (workgroup-uniform; an adaptive beam of zero or less -- beam_delta at or below zero with
  // max_active binding -- would let a bound at the cutoff hide the frame's minimum; a graph
  // with a NaN emitting weight has no lowest weight to bound with)
  const bool idx_frame = KD_LABEL_INDEX != 0 && M != 2 && n_in >= kIdxMinFront && c.ll_in_lds &&
                         c.adaptive_beam > 0.0f && c.P().emit_w_min == c.P().emit_w_min && Uni(s.top_valid) != 0;
  // (this frame's lowest and second-lowest acoustic costs, the likeliest label; what each of this
  // lane's tokens found in its state's label index)
  constexpr int kIdxSlots = (kT1Front + kBlock - 1) / kBlock;
  float idx_m0 = 0.0f, idx_m1 = 0.0f;
  int idx_l0 = 0;
  uint32_t idx_found[kIdxSlots];
#pragma unroll
  for (int j = 0; j < kIdxSlots; ++j) idx_found[j] = kIdxNone;
  if (idx_frame) LoadRowTop(s, idx_m0, idx_m1, idx_l0);

  // ---- emitting filter over the LDS frontier (faster-decoder.cc:195-236)
  // The frontier's best token goes first (faster-decoder.cc:176-189 does the same to
  // bound the cutoff early): the wave that holds it scores its arcs as one row right
  // here and publishes the lowest new cost, so the work-item loops below start under a
  // cutoff close to the final one and queue few arcs that the insert pass would drop.
  const unsigned long long best_key = Uni(s.f_best);
  for (int i0 = 0; i0 < n_in; i0 += nt) {  // (every wave takes every round)
    const int i = i0 + tid;
    int abeg = 0, deg = 0;
    double cst = 0.0;
    bool ok = false;
    if (i < n_in) {
      cst = s.it_cost[i];
      ok = cst < c.weight_cutoff;  // :202
      if (ok) {
        abeg = s.it_abeg[i];
        deg = s.it_deg[i];
        cnt.tokens_expanded += 1;
        cnt.arcs_scanned += deg;
      }
    }
    const unsigned long long first =
        __ballot(n_in >= kLeadMinFront && ok && deg > kSmallDeg && deg <= 64 && CK<M>(cst) == best_key);
    const int leader = first ? __builtin_ctzll(first) : -1;
    if (leader >= 0) {
      // (the leader is wave-uniform: v_readlane, not a shuffle through LDS)
      const int b_abeg = __builtin_amdgcn_readlane(abeg, leader), b_deg = __builtin_amdgcn_readlane(deg, leader);
      const int b_i = __builtin_amdgcn_readlane(i, leader);
      const double b_cst = s.it_cost[b_i];
      ArcRec ar1[1];
      int aidx1[1];
      bool on1[1];
      double cst1[1];
      uint32_t src1[1];
      on1[0] = lane < b_deg;
      aidx1[0] = b_abeg + lane;
      cst1[0] = b_cst;
      src1[0] = static_cast<uint32_t>(b_i);
      ar1[0] = LoadArcRecIf(c.P().arcs, static_cast<uint32_t>(aidx1[0]), on1[0], 0u);
      score_batch(EpsNo{}, std::integral_constant<int, 1>{}, ar1, aidx1, on1, cst1, src1);
      LaneMin(&s.min_new, c.run_min, c.run_min != kCostInf<M>);
    }
    if (ok && lane != leader) {
      if (deg <= kSmallDeg) lane_arcs(EpsNo{}, abeg, deg, cst, static_cast<uint32_t>(i));
      else if (!idx_frame) push_items(EpsNo{}, i, abeg, deg, cst, static_cast<uint32_t>(i));
    }
    // Label index, first half: where this token's arcs of the row's likeliest label are.  The
    // descent does not depend on the cutoff -- only the decision below does -- so its three
    // dependent loads run here, next to the other tokens' arc fetches and the best token's row,
    // instead of behind the barrier that publishes the bound.
    if (idx_frame && i0 / nt < kIdxSlots) {
      int probes = 0;
      uint32_t found = kIdxNone;
      if (ok && lane != leader && deg > kSmallDeg && deg < kIdxMaxDeg) {
        found = IndexFind(c.P(), abeg, deg, IdxKey(static_cast<uint32_t>(idx_l0 + 1), 0), probes);
        // ... and what that arc costs, for the BOUND only: nothing is queued here (the arc is
        // fetched and scored again in the second pass, by the index or with the token's row), but
        // its cost is a real candidate's, so it may lower the running minimum -- and the second
        // pass then decides, and queues, under a cutoff that EVERY token's likeliest-label arc has
        // tightened, not the best token's row alone.  (Without it a frame whose best token lacks
        // such an arc, or has more than 64 arcs and so no row to go first -- a history state of a
        // large LM -- let a quarter of its tokens fall back to reading all their arcs: bench
        // 38.2 -> 40.0 M, hlg-large 7.4-7.7 -> 8.0 M; candidates queued per frame 894 -> 615.)
        if (found != kIdxNone && IdxKeyLabel(found) == static_cast<uint32_t>(idx_l0 + 1)) {
          const ArcRec ar = LoadArcRec(c.P().arcs, static_cast<uint32_t>(abeg) + IdxKeyOffset(found));
          double nc;
          float aux;
          // (only a cost that is a number and below +infinity bounds anything: an arc whose
          // weight is NaN is never kept by the reference's test either -- faster-decoder.cc:211)
          const bool is_cost = ScoreArc<false>(c, ar, 0u, cst, kCostInf<M>, nc, aux);  // (an index frame: the row is in LDS)
          const unsigned long long nk = CK<M>(nc);
          if (is_cost && nk < c.run_min) c.run_min = nk;
          cnt.arcs_skipped -= 1;  // (an arc read: `arcs_skipped` is scanned minus read)
        }
      }
      // (a register array: written and read through constant indices only)
#pragma unroll
      for (int k = 0; k < kIdxSlots; ++k)
        if (k == i0 / nt) idx_found[k] = found;
      cnt.idx_probes += probes;
    }
  }
  PublishMin(c);
  tm.Sync(false);
  KD_LAP(s, 3);
  KD_PHASE(c);
  if (idx_frame) {
    // ---- the tokens of more than kSmallDeg arcs, now that the best token's arcs have bounded
    // the cutoff.  94 % of the arcs a frame used to score died at the cutoff: with peaky CTC
    // posteriors a token that is not among the frame's best can only continue along the
    // likeliest label.  Exactly: an arc out of a token with cost `cur` costs
    // fl(fl(cur + ac[ilabel]) + weight) (lattice-simple-decoder.cc:383-385; faster-decoder.cc:210
    // in double); float addition is monotone, so for every label but the row's likeliest it is at
    // least fl(fl(cur + m1) + w_min) -- m1 the row's second-lowest acoustic cost, w_min the graph's
    // lowest emitting weight.  If that bound is not below the RUNNING cutoff (which never
    // undercuts the final one), none of those arcs can be kept, whatever they are: the token
    // looks up its arcs of the likeliest label alone -- if even that label's bound is below the
    // cutoff -- through the label index, and scores them as ever.  Tokens with more room take
    // the work-item rows as before.  Which arcs are queued changes (fewer doomed ones); which
    // are below the final cutoff, and the minimum that defines it, do not.
    const float t_m0 = idx_m0, t_m1 = idx_m1;
    const int t_l0 = idx_l0;
    const float w_min = c.P().emit_w_min;
    for (int i0 = 0; i0 < n_in; i0 += nt) {
      const int i = i0 + tid;
      int abeg = 0, deg = 0;
      double cst = 0.0;
      bool ok = false;
      if (i < n_in) {
        cst = s.it_cost[i];
        ok = cst < c.weight_cutoff;  // :202
        if (ok) {
          abeg = s.it_abeg[i];
          deg = s.it_deg[i];
        }
      }
      // (the same leader as in the pass above: its row is done)
      const unsigned long long first =
          __ballot(n_in >= kLeadMinFront && ok && deg > kSmallDeg && deg <= 64 && CK<M>(cst) == best_key);
      const int leader = first ? __builtin_ctzll(first) : -1;
      if (ok && deg > kSmallDeg && lane != leader) {
        unsigned long long seen = s.min_new;
        seen = c.run_min < seen ? c.run_min : seen;
        bool others_out, best_in;
        if constexpr (M == 1) {
          const float cur = static_cast<float>(cst), lim = KCf<M>(seen) + c.adaptive_beam;
          others_out = (cur + t_m1) + w_min >= lim;
          best_in = (cur + t_m0) + w_min < lim;
        } else {
          const double lim = KC<M>(seen) + static_cast<double>(c.adaptive_beam);
          others_out = (static_cast<double>(w_min) + cst) + static_cast<double>(t_m1) >= lim;
          best_in = (static_cast<double>(w_min) + cst) + static_cast<double>(t_m0) < lim;
        }
        if (others_out && deg < kIdxMaxDeg) {
          int probes = 0, fetched = 0;
          uint32_t found = kIdxNone;
#pragma unroll
          for (int k = 0; k < kIdxSlots; ++k)
            if (k == i0 / nt) found = idx_found[k];
          if (best_in)
            IndexedArcsFrom(c.P(), abeg, deg, static_cast<uint32_t>(t_l0 + 1), found, probes, fetched,
                            [&](uint32_t aidx, const ArcRec &ar) {
                              FilterArc<false, true>(c, ar, aidx, cst, static_cast<uint32_t>(i));
                            });
          cnt.arcs_skipped += deg - fetched;
          cnt.idx_probes += probes;
        } else {
          push_items(EpsNo{}, i, abeg, deg, cst, static_cast<uint32_t>(i));
        }
      }
    }
    PublishMin(c);
    tm.Sync(false);
    KD_LAP(s, 2);
  KD_PHASE(c);
  }
  // The next frame's log-likelihood row is requested HERE, next to the work items' arc
  // gathers, and stored when the emitting filter is done with this frame's row.  (Requested
  // when the frame began -- as until round 3 -- its latency was not hidden but exposed: the
  // frame's first scratch reload waits for every load in flight, s_waitcnt vmcnt(0), and
  // that was the row.  2.9 k cycles a frame sat in "GetCutoff".)
#ifndef KD_ROW_EARLY
  float ll_next[kPfRegs];
#pragma unroll
  for (int k = 0; k < kPfRegs; ++k) {
    const int v = tid + k * kBlock;
    ll_next[k] = have_next && v < c.P().num_cols ? next_row[v] : 0.0f;
  }
#endif
  item_loop(EpsNo{});
  queue_sync();
  KD_LAP(s, 4);
  KD_PHASE(c);
  // (the epsilon rounds' work-item lists start empty: reset here, published by the insert's barrier)
  if (tid < kItemWords) (&s.it_count[0])[tid] = 0;
  // the emitting filter is done with this frame's log-likelihood row: the next
  // frame's row (fetched into registers when this frame began) takes its place
  const int q_emit = Uni(s.q_n);
  if (have_next) {
#pragma unroll
    for (int k = 0; k < kPfRegs; ++k) {
      const int v = tid + k * kBlock;
      if (v < c.P().num_cols) s.ll[v] = ll_next[k];
    }
    // ... and this wave's share of the new row's two likeliest labels (LoadRowTop; read by the
    // next frame, behind the frame-end barrier) -- if the next frame can be one that uses the
    // label index: a frame that queued a handful of candidates is followed by a small one, and
    // these hundred instructions were on the critical path of EVERY frame (a 4-7-token frame:
    // 24.7 k -> 26.1 k cycles when they went in).  A frame that finds no note does without.
    const bool note = q_emit >= kTopMinQueue;
    if (note) {
      Top2 top = Top2Empty();
#pragma unroll
      for (int k = 0; k < kPfRegs; ++k) {
        const int v = tid + k * kBlock;
        if (v < c.P().num_cols) Top2Add(top, ll_next[k], v);
      }
      top = WaveTop2(top);
      if (lane == 0) {
        s.top_a0[tid >> 6] = top.a0;
        s.top_a1[tid >> 6] = top.a1;
        s.top_i0[tid >> 6] = top.i0;
      }
    }
    if (tid == 0) s.top_valid = note ? 1 : 0;
  }
  if (c.lattice)  // float, lattice-simple-decoder.cc:388-390
    c.cutoff = KCf<M>(s.min_new) + c.adaptive_beam;
  else
    c.cutoff = KC<M>(s.min_new) + static_cast<double>(c.adaptive_beam);  // value of :240
  c.cutoff = Uni(c.cutoff);
  if (tid == 0) {  // what a replay in the big table needs
    s.ctl[frame & 1].q_emit = q_emit;
    s.ctl[frame & 1].cutoff = c.cutoff;
  }
  if (q_emit > c.q_cap) return kRedo;  // the big path scans again under the final cutoff
  const unsigned long long cutkey = CK<M>(c.cutoff);

  // Insert queue entries [b, e): claim + relax; a token that is new, or (epsilon
  // rounds) improved, and has epsilon arcs is flagged in worklist mask `out`.
  auto insert = [&](auto eps_tag, int b, int e, int out) {
    constexpr bool EPS = decltype(eps_tag)::value;
    split(b, e, [&](auto w, int lo, int hi) {
      for (int i = lo + tid; i < hi; i += nt) {
        const uint32_t k = q_h(w, i);
        const unsigned long long nk = q_nk(w, i);
        [[maybe_unused]] uint32_t degs = 0;
        if constexpr (M == 1) degs = (q_src(w, i) & ~kSrcEps) >> kT1SrcDegShift;
        uint32_t h = kKeyEmpty;
        if ((EPS || nk < cutkey) && !s.overflow) {
          h = t.Claim(k);
          if (h == kKeyEmpty) {
            s.overflow = 1;
          } else {
            const unsigned long long old = atomicMin(&s.cost[h], nk);
            bool push = false;
            if (old == kCostDead) {  // the state's first token this frame
              const int pos = AggInc(&s.n_live);
              if (pos < static_cast<int>(t.FillLimit())) T1Live(s)[pos] = static_cast<uint16_t>(h);
              else s.overflow = 1;
              // (lattice decoders: the slot's cost word holds the state's first arc in its low
              // half -- it came with the key -- and its free arc word takes the degrees)
              if constexpr (M == 1) s.arc[h] = degs;
              push = (k & kHasEps) != 0;  // every new token is expanded once (:63-65)
            } else if (EPS && old > nk && (k & kHasEps)) {  // re-queue on improvement, :97-113
              push = true;
            }
            if (push) {
              atomicOr(&wlmask[out * kMaskWords + (h >> 5)], 1u << (h & 31));
              s.wl_count[out] = 1;
            }
            if (!EPS && c.lattice) AggAdd1(&s.n_emit_links);
          }
        }
        if constexpr (decltype(w)::value) T1QH(s)[i] = h; else c.q_h[i] = h;
      }
    });
  };

  insert(EpsNo{}, 0, q_emit, 0);
  tm.Sync(false);
  KD_LAP(s, 5);
  KD_PHASE(c);
  if (Uni(s.overflow)) return kRetry;

  // ---- ProcessNonemitting to a fixpoint (:59-119)
  // A round is scan -> barrier -> (work items, if any were pushed) -> insert -> barrier.  The
  // other mask is cleared at the top without a barrier of its own (its last readers were the
  // scan of the round before, two barriers ago; its next writers are this round's inserts,
  // behind the scan's barrier), and the work-item counters are reset by one lane between a
  // round's work items and its inserts (reset_items): until round 5 a round began with both and a
  // barrier, and ran the work-item pass with its barrier whether or not there were items.
  auto reset_items = [&]() {
    if (tid < kItemWords) (&s.it_count[0])[tid] = 0;
  };
  int cur = 0, q_begin = q_emit;
  while (Uni(s.wl_count[cur]) != 0) {
    for (int w = tid; w < kMaskWords; w += nt) wlmask[(cur ^ 1) * kMaskWords + w] = 0;
    // The team's threads scan the worklist mask, 8 slots each (a lane rarely finds
    // more than one token).  Tokens are taken two at a time: both state records are
    // requested before either is used, and so are the arcs of low-degree tokens.
    for (int w8 = tid; w8 < kT1Slots / 8; w8 += nt) {
      uint32_t m = (wlmask[cur * kMaskWords + (w8 >> 2)] >> ((w8 & 3) * 8)) & 0xFFu;
      const int hbase = w8 * 8;
      while (m) {
        uint32_t h[2];
        h[0] = static_cast<uint32_t>(hbase + __builtin_ctz(m));
        m &= m - 1;
        const bool two = m != 0;
        h[1] = two ? static_cast<uint32_t>(hbase + __builtin_ctz(m)) : h[0];
        if (two) m &= m - 1;
        double cst[2];
        bool ok[2], own[2];
        StateRec r[2];
        int abeg[2], deg[2];
        if constexpr (M == 1) {
          // the token's epsilon arcs are known from the arc that created it (T1Put): the
          // slot's cost word holds the state's first arc, its arc word the degrees
          uint32_t dg[2], first[2];
          bool lookup = false;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const unsigned long long cw = s.cost[h[j]];
            cst[j] = KC<M>(cw);
            first[j] = static_cast<uint32_t>(cw);
            dg[j] = s.arc[h[j]];
            ok[j] = (j == 0 || two) && !(cst[j] > c.cutoff);  // :74
            lookup |= ok[j] && (DegsEmit(dg[j]) == kEmitDegEscape || DegsEps(dg[j]) == kEpsDegEscape);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            r[j].emit_end = static_cast<int32_t>(first[j] + DegsEmit(dg[j]));
            r[j].arc_end = r[j].emit_end + static_cast<int32_t>(DegsEps(dg[j]));
          }
          if (lookup) {  // (a degree that does not fit its field: the state record, as ever)
#pragma unroll
            for (int j = 0; j < 2; ++j) r[j] = LoadState(c.P().states, s.key[h[j]] & kStateMask);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            cst[j] = KC<M>(s.cost[h[j]]);
            ok[j] = (j == 0 || two) && !(cst[j] > c.cutoff);  // :74
            r[j] = LoadState(c.P().states, s.key[h[j]] & kStateMask);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          abeg[j] = r[j].emit_end;
          deg[j] = r[j].arc_end - r[j].emit_end;
          own[j] = ok[j];
          if (!ok[j]) continue;
          cnt.eps_exp += 1;
          cnt.eps_arcs += deg[j];
          if (deg[j] > kSmallDeg) {
            const int q = AggInc(&s.it_count[1]);  // descriptor slot in it_*
            if (q < kT1Front) {
              const uint32_t srcid = h[j] | kSrcEps;
              s.it_abeg[q] = abeg[j];
              s.it_deg[q] = deg[j];
              s.it_cost[q] = cst[j];
              s.it_src[q] = srcid;
              push_items(EpsYes{}, q, abeg[j], deg[j], cst[j], srcid);
              own[j] = false;
            }
          }
        }
        // tokens that stay on the lane: the first kSmallDeg arcs of both in flight
        ArcRec ar[2][kSmallDeg];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int a = 0; a < kSmallDeg; ++a)
            ar[j][a] = LoadArcRecIf(c.P().arcs, static_cast<uint32_t>(abeg[j] + a), own[j] && a < deg[j], 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int a = 0; a < kSmallDeg; ++a)
            if (own[j] && a < deg[j])
              FilterArc<true, true>(c, ar[j][a], static_cast<uint32_t>(abeg[j] + a), cst[j], h[j] | kSrcEps);
        // (a token the full descriptor list sent back may have more arcs than that)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (own[j] && deg[j] > kSmallDeg)
            lane_arcs(EpsYes{}, abeg[j] + kSmallDeg, deg[j] - kSmallDeg, cst[j], h[j] | kSrcEps);
      }
    }
    tm.Sync(false);
    KD_LAP(s, 7);
  KD_PHASE(c);
    if (Uni(s.it_count[0] | s.it_n32 | s.it_n16) != 0) {
      item_loop(EpsYes{});
      queue_sync();
      reset_items();
    } else if (Uni(s.q_n) > kQL) {
      tm.Sync(true);  // (queue entries past the LDS part: their stores have landed)
    }
    // (every wave has tested wl_count[cur] at the top of this round; the next round's inserts,
    // two barriers from here, raise it again)
    if (tid == 0) s.wl_count[cur] = 0;
    KD_LAP(s, 8);
  KD_PHASE(c);
    const int q_end = Uni(s.q_n);
    if (q_end > c.q_cap) return kRetry;
    insert(EpsYes{}, q_begin, q_end, cur ^ 1);
    tm.Sync(false);
    KD_LAP(s, 9);
  KD_PHASE(c);
    if (Uni(s.overflow)) return kRetry;
    q_begin = q_end;
    cur ^= 1;
  }

  // ---- resolve R1: entries that hold their slot's final cost compete for the
  // lowest arc index
  const int n_q = q_begin;
  const int n_out = Uni(s.n_live);
  if constexpr (M != 1) {  // (lattice tokens keep forward links, not back-pointers)
    split(0, n_q, [&](auto w, int lo, int hi) {
      for (int i = lo + tid; i < hi; i += nt) {
        const uint32_t h = q_h(w, i);
        if (h != kKeyEmpty && s.cost[h] == q_nk(w, i)) atomicMin(&s.arc[h], q_arc(w, i));
      }
    });
  }
  // the frame's token block and room for its forward links (at most one per queue entry
  // below the cutoff) come from this utterance's chunks of the pools; a new chunk is a
  // rare event (one in some twenty frames) with two team barriers of its own
  const int link_room = c.lattice ? Uni(s.n_emit_links) + (n_q - q_emit) : 0;
  if (Uni(s.arena_next) + n_out > Uni(s.arena_end) ||
      (c.lattice && Uni(s.link_next) + link_room > Uni(s.link_end))) {
    tm.Sync(false);
    if (tid == 0) s.alloc_ok = static_cast<int32_t>(TakeChunks(c.P(), s, c.utt, frame + 1, n_out, link_room));
    tm.Sync(false);
    const uint32_t e = static_cast<uint32_t>(Uni(s.alloc_ok));
    if (e) return e;
  }
  const long long base_out = Uni(s.arena_next);
  // ---- C1: live list -> token records; a slot's key word now holds its index in
  // the block.  Each token's state record is fetched now for the next frame's filter.
  const bool next_small = n_out <= kT1Front;
  unsigned long long my_min = kCostDead;
  // The new frontier's state records (at most kNextRecs tokens per lane when the next
  // frame is a tier-1 frame): requested here, used when the frame ends -- the gather
  // overlaps the passes in between.
  constexpr int kNextRecs = (kT1Front + kBlock - 1) / kBlock;
  ArcRange next_rec[kNextRecs];
  bool next_on[kNextRecs];
  {
    // (lattice decoders: a token's arc range came with the arc that created it -- its slot
    // holds it -- and only a degree that does not fit its field is looked up here)
    uint32_t st_next[kNextRecs];
#pragma unroll
    for (int j = 0; j < kNextRecs; ++j) {
      const int i = tid + j * nt;
      next_on[j] = next_small && i < n_out;
      const int h = T1Live(s)[next_on[j] ? i : 0];
      if constexpr (M == 1) next_on[j] = next_on[j] && DegsEmit(s.arc[h]) == kEmitDegEscape;
      st_next[j] = next_on[j] ? s.key[h] & kStateMask : 0u;  // (before the key words become indices)
    }
    bool any_on = false;
#pragma unroll
    for (int j = 0; j < kNextRecs; ++j) any_on |= next_on[j];
    if (M != 1 || __ballot(any_on)) {
#pragma unroll
      for (int j = 0; j < kNextRecs; ++j) next_rec[j] = LoadArcRange(c.P().states, st_next[j]);
    } else {
#pragma unroll
      for (int j = 0; j < kNextRecs; ++j) next_rec[j] = ArcRange{0, 0};
    }
  }
  for (int i = tid; i < n_out; i += nt) {
    const int h = T1Live(s)[i];
    const unsigned long long ck = s.cost[h];
    const uint32_t st = s.key[h] & kStateMask;
    const long long g = base_out + i;
    ToG(c.P().tok_state)[g] = static_cast<int32_t>(st);
    ToG(c.P().tok_cost)[g] = KC<M>(ck);
    if (c.lattice) ToG(c.P().tok_extra)[g] = 0.0f;  // lattice-simple-decoder.cc:91-95
    // (lattice: the key without the arc index that rides in its low word -- the next frame
    // looks for its best token by this value)
    const unsigned long long ckc = M == 1 ? ck & 0xFFFFFFFF00000000ull : ck;
    my_min = ckc < my_min ? ckc : my_min;
    s.key[h] = static_cast<uint32_t>(i);
    if (next_small) {
      s.it_cost[i] = KC<M>(ck);
      s.it_src[i] = st;
      if constexpr (M == 1) {
        const uint32_t emit = DegsEmit(s.arc[h]);
        if (emit != kEmitDegEscape) {
          s.it_abeg[i] = static_cast<int32_t>(static_cast<uint32_t>(ck));
          s.it_deg[i] = static_cast<int32_t>(emit);
        } else {
          s.it_deg[i] = -1;  // (looked up: next_rec)
        }
      }
    }
  }
  if (my_min != kCostDead && my_min < s.f_best_next) atomicMin(&s.f_best_next, my_min);
  tm.Sync(false);  // R1's arc words and the slots' index words are final
  KD_LAP(s, 10);
  KD_PHASE(c);
  if (tid == 0) {
    s.base_out = base_out;
    s.arena_next = base_out + n_out;
  }
  KD_LAP(s, 11);
  KD_PHASE(c);
  // ---- R2: the winning entry of each slot writes the back-pointer
  if constexpr (M != 1) {
    split(0, n_q, [&](auto w, int lo, int hi) {
      for (int i = lo + tid; i < hi; i += nt) {
        const uint32_t h = q_h(w, i);
        if (h == kKeyEmpty) continue;
        const uint32_t aidx = q_arc(w, i);
        if (s.cost[h] == q_nk(w, i) && s.arc[h] == aidx) {
          const long long g = base_out + s.key[h];
          const uint32_t sr = q_src(w, i);
          const long long prev = (sr & kSrcEps) ? base_out + s.key[sr & ~kSrcEps] : c.base_in + sr;
          ToG(c.P().tok_prev)[g] = static_cast<int32_t>(prev);
          ToG(c.P().tok_arc)[g] = static_cast<int32_t>(aidx);
        }
      }
    });
  }
  KD_LAP(s, 16);
  KD_PHASE(c);
  // ---- forward links (see FrameBack)
  if (c.lattice) {
    const long long lbase = Uni(s.link_next);  // (room was taken with the token block)
    const int n_emit = Uni(s.n_emit_links);
    split(0, n_q, [&](auto w, int lo, int hi) {
      for (int i = lo + tid; i < hi; i += nt) {
        // the entry's queue words together, then its table words together (read one
        // by one behind the tests below, each waits for the one before it)
        const uint32_t h = q_h(w, i), sr = q_src(w, i) & (M == 1 ? kT1SrcMask : ~0u), aidx = q_arc(w, i);
        const float qa = q_ac(w, i), qg = q_gr(w, i);
        const unsigned long long nk = q_nk(w, i);
        const bool ok = h != kKeyEmpty, eps = i >= q_emit;
        const uint32_t hd = ok ? h : 0u, hs = ok && eps ? sr & ~kSrcEps : 0u;
        const uint32_t kd = s.key[hd], ks = s.key[hs];
        // (the lattice decoders' key is the high word; asking for the 64-bit word and
        // letting hipcc narrow the read trips a bug in its instruction selection here)
        float cdf, csf;
        if constexpr (M == 1) {
          cdf = KeyFloat(reinterpret_cast<const uint32_t *>(&s.cost[hd])[1]);
          csf = KeyFloat(reinterpret_cast<const uint32_t *>(&s.cost[hs])[1]);
        } else {
          cdf = KCf<M>(s.cost[hd]);
          csf = KCf<M>(s.cost[hs]);
        }
        if (!ok) continue;
        long long at, src_tok;
        float ac;
        if (!eps) {
          src_tok = c.base_in + sr;
          ac = qa;
          at = lbase + AggInc(&s.n_emit_written);
        } else {
          // source cost at generation vs its final cost (its slot's cost word is intact)
          if (qa != csf) continue;
          src_tok = base_out + ks;
          ac = 0.0f;
          at = lbase + n_emit + AggInc(&s.n_eps_links);
        }
        ToG(c.P().link_src)[at] = static_cast<int32_t>(src_tok);
        ToG(c.P().link_dst)[at] = static_cast<int32_t>(base_out + kd);
        ToG(c.P().link_arc)[at] = static_cast<int32_t>(aidx);
        ToG(c.P().link_ac)[at] = ac;
        ToG(c.P().link_graph)[at] = qg;
        ToG(c.P().link_delta)[at] = KCf<M>(nk) - cdf;
      }
    });
  }
  KD_LAP(s, 17);
  KD_PHASE(c);
  tm.Sync(false);
  KD_LAP(s, 12);
  KD_PHASE(c);
  // (final now; read by every lane before they are reset for the next frame)
  const int n_emit_fin = Uni(s.n_emit_links), n_eps_fin = Uni(s.n_eps_links);
  const unsigned long long best_next = Uni(s.f_best_next);
  // ---- clear the table; publish the next frontier's arc ranges
  for (int i = tid; i < n_out; i += nt) {
    const int h = T1Live(s)[i];
    s.key[h] = kKeyEmpty;
    s.cost[h] = kCostDead;
    s.arc[h] = kArcNone;
  }
#pragma unroll
  for (int j = 0; j < kNextRecs; ++j) {
    if (next_on[j]) {
      s.it_abeg[tid + j * nt] = next_rec[j].arc_begin;
      s.it_deg[tid + j * nt] = next_rec[j].emit_end - next_rec[j].arc_begin;
    }
  }
  // Everything below that READS the frame's counters (n_emit_fin, n_eps_fin, best_next) and
  // the reset of those counters for the next frame are done by lanes of ONE wave, the first:
  // within a wave LDS operations execute in program order, between waves only a barrier
  // orders them.  (Until round 4 the third wave did the reset while the first wrote the frame
  // index: had the first wave ever fallen behind the third by the length of the table-clearing
  // loop above, it would have read zeros -- a frame with no links, a next frame without a best
  // cost.  Never seen in the shipped build; the timing-perturbation build (make jitter) fails
  // nine cases in ten on it.)
  if (tid == 32) {
    // launch counters (fire-and-forget LDS adds; the per-lane work counters are folded once,
    // when the launch ends)
    atomicAdd(&s.ustat[KD_STAT_ARCS_KEPT], static_cast<unsigned long long>(q_emit));
    atomicAdd(&s.ustat[KD_STAT_TOKENS_OUT], static_cast<unsigned long long>(n_out));
    atomicMax(&s.ustat[KD_STAT_MAX_TOKENS], static_cast<unsigned long long>(n_out));
    atomicAdd(&s.ustat[KD_STAT_FRAMES], 1ull);
    atomicAdd(&s.ustat[KD_STAT_SMALL_FRAMES], 1ull);
    atomicAdd(&s.ustat[KD_STAT_TOKENS_IN], static_cast<unsigned long long>(n_in));
    if (c.lattice)
      atomicAdd(&s.ustat[KD_STAT_LINKS], static_cast<unsigned long long>(n_emit_fin + n_eps_fin));
  }
  if (tid == 0) {
    // frame index and the next frame's start state
    const long long fi = static_cast<long long>(c.utt) * (c.P().max_frames + 1) + (frame + 1);
    ToG(c.P().frame_base)[fi] = base_out;
    ToG(c.P().frame_count)[fi] = n_out;
    if (c.lattice) {
      const long long lbase = s.link_next;
      const long long fb = static_cast<long long>(c.utt) * (c.P().max_frames + 1);
      ToG(c.P().link_emit_base)[fb + frame] = lbase;
      ToG(c.P().link_emit_count)[fb + frame] = n_emit_fin;
      ToG(c.P().link_eps_base)[fb + frame + 1] = lbase + n_emit_fin;
      ToG(c.P().link_eps_count)[fb + frame + 1] = n_eps_fin;
      s.link_next = lbase + n_emit_fin + n_eps_fin;
    }
    s.f_best = best_next;
    s.ctl[frame & 1].n_in = n_out;
    s.ctl[frame & 1].base_out = base_out;
  }
  ResetFrame<M>(s, 0);  // (the first wave, behind its own reads above)
  return 0;
}

// Frontier of a tier-1 frame that follows a frame of another tier (or the start
// of a launch): arena -> LDS.
template <int M>
__device__ __forceinline__ void LoadFront(const FrameCtx<M> &c, const Team &tm) {
  CParams &p = c.P();
  Shared &s = c.s;
  for (int i = threadIdx.x; i < c.n_in; i += tm.nt) {
    const uint32_t st = static_cast<uint32_t>(ToG(p.tok_state)[c.base_in + i]);
    const StateRec r = LoadState(p.states, st);
    s.it_cost[i] = ToG(p.tok_cost)[c.base_in + i];
    s.it_src[i] = st;
    s.it_abeg[i] = r.arc_begin;
    s.it_deg[i] = r.emit_end - r.arc_begin;
  }
  tm.Sync(false);
}

template <int M>
__device__ __forceinline__ void InitCtx(FrameCtx<M> &c, CParams &p) {
  // the spill table, the candidate queue, the live list and the worklists belong to the
  // WORKGROUP (they are empty between frames), not to the utterance it is decoding
  const long long gslots = 1ll << p.g_log2_slots, wg = p.sched.wg_base + blockIdx.x;
  c.wl0 = ToG(p.g_wl) + (2ll * wg) * gslots;
  c.wl1 = ToG(p.g_wl) + (2ll * wg + 1) * gslots;
  c.live = ToG(p.g_used) + wg * gslots;
  c.q_cap = static_cast<int>(2 * gslots);
  c.q_h = ToG(p.g_log_h) + wg * 2 * gslots;
  c.q_arc = ToG(p.g_log_arc) + wg * 2 * gslots;
  c.q_src = ToG(p.g_log_src) + wg * 2 * gslots;
  c.q_nk = ToG(p.g_log_nk) + wg * 2 * gslots;
  c.q_ac = ToG(p.g_log_ac) + wg * 2 * gslots;
  c.q_gr = ToG(p.g_log_gr) + wg * 2 * gslots;
  c.min_only = false;
  c.run_min = kCostInf<M>;
  c.replay = false;
  c.ll_in_lds = p.num_cols <= kVmax;
}

// Spill path: a function of its own, so that its registers are allocated apart
// from the hot path's.  Everything it needs comes as scalars or from LDS.
template <int M>
__device__ __noinline__ uint32_t FrameBackSpill(CParams *pp, int utt, long long base_in, int n_in,
                                                double cutoff, int frame, int q_emit,
                                                int n_expanded, int n_arcs, bool replay) {
  CParams &p = *pp;
  Shared &s = Lds();
  FrameCtx<M> c{&p, s, utt};
  InitCtx(c, p);
  c.base_in = base_in;
  c.n_in = n_in;
  c.cutoff = cutoff;
  c.replay = replay;
  Counters cnt;
  cnt.tokens_expanded = n_expanded;
  cnt.arcs_scanned = n_arcs;
  const long long gslots = 1ll << p.g_log2_slots, wg = p.sched.wg_base + blockIdx.x;
  // The table is as large as the frame needs, not as large as it could be: a frame of
  // 8 k tokens hashes into 32 k slots (512 KiB of key / cost / arc words, which stay in
  // the XCD's L2) instead of scattering its atomics over all 2^17..2^19 slots (2-8 MiB:
  // every one of them a memory transaction), and the worklist bitmask that each epsilon
  // round clears shrinks with it.  The emitting candidates bound the emitting tokens;
  // the epsilon closure may add more: a frame that outgrows the guess runs again in the
  // whole table.
  // (the worklist bitmask of a global table lives in the LDS table's cost words: 64 slots a word)
  constexpr int kMaxLg = kLog2Slots + 6;
  const int lg_max = p.g_log2_slots < kMaxLg ? p.g_log2_slots : kMaxLg;
  int lg = 14 < lg_max ? 14 : lg_max;
  while ((1ll << lg) * 4 < static_cast<long long>(KD_GT_SIZE_X4) * q_emit && lg < lg_max) ++lg;
  uint32_t r;
  while (true) {
    Table<false> tg{s, ToG(p.g_slot) + 4 * wg * gslots, reinterpret_cast<uint32_t *>(s.cost),
                    lg, (1u << lg) - 1u};
    r = FrameBack<false>(c, tg, frame, q_emit, cnt);
    if (r == 0) break;
    // Leave the global table clean: this workgroup goes on with other utterances of the
    // work queue (and, after the host has grown what ran out, with this one).  A frame
    // that ran out of token or link records has a complete live list; one that overflowed
    // the table itself has claimed slots that are on no list, so all of it is wiped.
    __syncthreads();
    if (r == kRetry) {
      for (long long h = threadIdx.x; h < (1ll << lg); h += kBlock) tg.ClearSlot(static_cast<uint32_t>(h));
    } else {
      const int used = s.n_live < static_cast<int>(tg.FillLimit()) ? s.n_live
                                                                   : static_cast<int>(tg.FillLimit());
      ClearLive<false>(c, tg, used);
    }
    if (r != kRetry || lg == lg_max) break;
    // once more, in the whole table (the queue's emitting part is intact; its
    // destination words now hold slots of the table just wiped: replay from the arcs)
    lg = lg_max;
    if (threadIdx.x == 0) {
      s.q_n = q_emit;
      s.n_live = 0;
      s.overflow = 0;
      s.wl_count[0] = 0;
      s.wl_count[1] = 0;
      s.n_emit_links = 0;
      s.n_eps_links = 0;
      s.n_emit_written = 0;
    }
    c.replay = true;
    cnt.eps_exp = 0;
    cnt.eps_arcs = 0;
    __syncthreads();
  }
  __syncthreads();
  return r == kRetry ? static_cast<uint32_t>(KD_UTT_TABLE_FULL) : r;
}

// Tiers 2 and 3: a frame whose frontier is larger than one token per lane, a
// tier-1 frame that outgrew its table (flags & 1: its queue is replayed), and
// InitDecoding (frame == -1).  Not inlined: these frames are 4% of an HLG decode
// and must not cost the small frames registers.  The new frontier is returned in
// LDS (base_out, n_live / n_act).
template <int M>
__device__ __noinline__ uint32_t BigFrame(CParams *pp, int utt, int frame, long long base_in,
                                          int n_in, double cutoff, int q_emit, int flags,
                                          int n_expanded, int n_arcs) {
  constexpr int KD_BUCKET = kBucketBig;
  (void)KD_BUCKET;
  CParams &p = *pp;
  Shared &s = Lds();
  const int tid = threadIdx.x;
  const bool filtered = flags & 1, upper_dirty = flags & 2, best_stale = flags & 4;
  Table<true> tl{s, nullptr, s.bitmask, kLog2Slots, kSlots - 1};
  FrameCtx<M> c{&p, s, utt};
  InitCtx(c, p);
  c.base_in = base_in;
  c.n_in = n_in;
  c.cutoff = cutoff;
  Counters cnt;
  cnt.tokens_expanded = n_expanded;
  cnt.arcs_scanned = n_arcs;
  if (frame >= 0)
    c.ll_row = ToG(reinterpret_cast<const float *>(s.ud_ll)) +
               static_cast<long long>(frame - s.ud_offset) * p.row_stride;
  __syncthreads();  // token records stored by a small team are visible to every wave
  if (upper_dirty) FullClear<true>(tl, s);
  if (filtered) {
    if (tid == 0) {
      s.q_n = q_emit;
      s.n_live = 0;
      s.overflow = 0;
      s.wl_count[0] = 0;
      s.wl_count[1] = 0;
      s.n_emit_links = 0;
      s.n_eps_links = 0;
      s.n_emit_written = 0;
    }
    __syncthreads();
  } else if (frame >= 0) {
    if (best_stale) FrontierStats(c);
    if (c.ll_in_lds) FrameFront(c, cnt);  // (two copies: see the tier-1 call)
    else FrameFront(c, cnt);
    q_emit = s.q_n;
  } else {
    // faster-decoder.cc:53; the lattice decoders use best (= 0) + beam
    // (lattice-simple-decoder.cc:135-154)
    c.cutoff = c.lattice ? static_cast<double>(0.0f + p.beam) : static_cast<double>(FLT_MAX);
    if (c.simple) c.cutoff = 0.0 + static_cast<double>(p.beam);
  }
  c.replay = filtered;
  if (kT1SoftQueue > 0 && tid == 0) s.ctl[0].pad = q_emit;  // (read back by RunUtterance behind this call's barriers)
  // the queue is a complete journal of the frame's inserts: a frame that does
  // not fit the LDS table is replayed from it into the global table
  if (q_emit > c.q_cap) return KD_UTT_TABLE_FULL;
  // a frontier this large, or more than two candidates per slot (the queue also
  // holds entries the exact cutoff will drop), will not fit the LDS table
  const bool try_lds = c.n_in <= kLdsFillLimit && q_emit <= 2 * kLdsFillLimit;
  uint32_t r = kRetry;
  if (try_lds) r = FrameBack<true>(c, tl, frame, q_emit, cnt);
  if (r == kRetry) {
    __syncthreads();
    KD_LAP(s, 15);
    FullClear<true>(tl, s);  // s.cost becomes the spill bitmask
    if (tid == 0) {
      s.q_n = q_emit;
      s.n_live = 0;
      s.overflow = 0;
      s.wl_count[0] = 0;
      s.wl_count[1] = 0;
      s.n_emit_links = 0;
      s.n_eps_links = 0;
      s.n_emit_written = 0;
    }
    __syncthreads();
    // (epsilon counters of the aborted closure are dropped)
    r = FrameBackSpill<M>(pp, utt, c.base_in, c.n_in, c.cutoff, frame, q_emit,
                          cnt.tokens_expanded, cnt.arcs_scanned, try_lds || filtered);
    FullClear<true>(tl, s);
    KD_LAP(s, 14);
  }
  return r;
}

#include "kd_lattice_tail.h"
#include "kd_path_tail.h"

// Fused log-softmax (kd_batch_set_log_softmax): the caller's matrices hold
// logits, and the row a frame reads is normalised where it already sits -- in LDS -- by the
// whole workgroup: x - (max + log(sum(exp(x - max)))), float, the partial sums combined in a
// fixed order (so a row comes out the same whichever workgroup stages it).  The reference
// leaves log_softmax to the caller (DecodableCtc takes log-probs, csrc/decodable-ctc.cc:28);
// icefall applies it to the network output before decoding.
__device__ __forceinline__ void NormalizeRow(Shared &s, int V) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *red = reinterpret_cast<float *>(s.red64);
  float m = -FLT_MAX;
  for (int v = tid; v < V; v += kBlock) m = fmaxf(m, s.ll[v]);
  m = WaveMaxF(m);
  if (lane == 0) red[wave] = m;
  LdsBarrier();
  float mx = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) mx = fmaxf(mx, red[w]);
  LdsBarrier();
  float sum = 0.0f;
  for (int v = tid; v < V; v += kBlock) sum += expf(s.ll[v] - mx);
  sum = WaveSumF(sum);
  if (lane == 0) red[wave] = sum;
  LdsBarrier();
  float total = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) total += red[w];
  const float lse = mx + logf(total);
  for (int v = tid; v < V; v += kBlock) s.ll[v] -= lse;
  LdsBarrier();
}

// What one workgroup does for one utterance of the work queue: InitDecoding if asked,
// the frames up to the utterance's target, then -- lattice decoders, when asked -- the
// tail: ComputeFinalCosts, FinalizeDecoding's backward prune and GetRawLattice into the
// output pool (kd_lattice_tail.h), all while the utterance's LDS is still this workgroup's.
// What a call of RunUtterance ends with: the utterance is where it was sent (or stopped), or it
// changes hands -- the frame loop's state is in UttResume and the caller queues it for the other
// kind of workgroup.
constexpr int kSegFinished = 0, kSegToSmall = 1, kSegToLarge = 2;
// how the utterance came to this workgroup
constexpr int kTakeFresh = 0, kTakeResumed = 1, kTakeStolen = 2;  // (stolen: a large workgroup, from the small queue)

__device__ __forceinline__ uint32_t AgentLoad(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the XCD this workgroup runs on (HW_REG_XCC_ID): its queues are that XCD's
__device__ __forceinline__ int ThisXcd() {
  return static_cast<int>(Uni(__builtin_amdgcn_s_getreg(20 | (31 << 11)))) & (kMaxXcd - 1);
}
__shared__ int kd_work_item, kd_work_take, kd_progressed, kd_decision;
__shared__ long long kd_t_setup;  // (lane 0: cycles between taking an utterance up and its first frame)
__shared__ long long kd_prof[8];   // (lane 0: idle, pop, setup, busy, taken fresh / resumed / stolen, stamp)
// Workgroups of the other kind have started on this XCD (they stay until the launch ends).
// Call from ONE lane: the words read here change under the launch's feet, and a workgroup's
// waves must not come to different conclusions (Decide).
__device__ __forceinline__ bool OtherKindRuns(CParams &p) {
  return AgentLoad(&p.sched.ctl->n_reg[ThisXcd()][1 - kKind]) != 0u;
}
// The frontier's size says the next frame is the other kind's: does the utterance go?  Not
// before this workgroup has run a frame or a prune of it (two workgroups must not pass an
// utterance back and forth), not when there is nobody to take it, and a large workgroup that
// took it from the small queue because nothing waited for its own kind keeps it while that is
// still so.  Call from ONE lane.
__device__ __forceinline__ bool HandOver(CParams &p) {
  if (kd_progressed == 0) return false;
  const int xcd = ThisXcd();
  if (AgentLoad(&p.sched.ctl->n_reg[xcd][1 - kKind]) == 0u) return false;
  if (kKind == kKindLarge && kd_work_take == kTakeStolen) {
    const SchedQueue *q = &p.sched.ctl->q[xcd][kKindLarge];
    return static_cast<int32_t>(AgentLoad(&q->tail) - AgentLoad(&q->head)) > 0;
  }
  return true;
}
// One lane's answer for the whole workgroup (every wave calls; two barriers).
template <class F>
__device__ __forceinline__ bool Decide(F f) {
  if (threadIdx.x == 0) kd_decision = f() ? 1 : 0;
  LdsBarrier();
  const bool r = Uni(kd_decision) != 0;
  LdsBarrier();  // (read by every wave before the word is written again)
  return r;
}

template <int M>
__device__ __forceinline__ int RunUtterance(CParams &p, Shared &s, const int utt, const bool resumed) {
  const int tid = threadIdx.x;
  constexpr int KD_BUCKET = kBucketBig;
  (void)KD_BUCKET;

  Table<true> tl{s, nullptr, s.bitmask, kLog2Slots, kSlots - 1};
  FrameCtx<M> c{&p, s, utt};
  InitCtx(c, p);

  UttCtl ud;
  int frame, visited;
  [[maybe_unused]] int prev_q_emit = 0;  // candidates the previous frame's emitting filter queued
  unsigned long long f_best_in = 0;
  c.base_in = 0;
  c.n_in = 0;
  if (resumed) {
    // every word the loop needs, from ONE record in device memory, the loads issued together
    const KD_G UttResume *rs = ToG(c.P().sched.resume) + utt;
    ud.ll = reinterpret_cast<const float *>(Uni(reinterpret_cast<unsigned long long>(rs->ll)));
    ud.offset = Uni(rs->offset);
    ud.target = Uni(rs->target);
    ud.flags = Uni(rs->flags);
    frame = Uni(rs->frame);
    c.base_in = Uni(rs->base_in);
    c.n_in = Uni(rs->n_in);
    prev_q_emit = Uni(rs->prev_q_emit);
    f_best_in = Uni(rs->f_best);
    visited = Uni(rs->visited);
  } else {
    ud.ll = reinterpret_cast<const float *>(Uni(reinterpret_cast<unsigned long long>(ToG(c.P().utt)[utt].ll)));
    ud.offset = Uni(ToG(c.P().utt)[utt].offset);
    ud.target = Uni(ToG(c.P().utt)[utt].target);
    ud.flags = Uni(ToG(c.P().utt)[utt].flags);
    frame = (ud.flags & kUttInit) ? -1 : Uni(ToG(c.P().frames_decoded)[utt]);
    visited = (ud.flags & kUttInit) ? -1 : Uni(ToG(c.P().mem)[utt].visited);
  }
  const bool do_init = !resumed && (ud.flags & kUttInit) != 0;
  // (an utterance that stopped for want of capacity is only sent back here after the
  // host has grown what was exhausted: the status word is output, not state)
  uint32_t err = 0u;
  const int target = ud.target;
  const bool search = err == 0 && frame < target;  // (otherwise only the tail has work)
  if (tid < KD_NUM_STATS) s.stat[tid] = 0;
  if (tid < 4) s.frame_cnt[tid] = 0;
  if (tid == 0) {
    s.ud_ll = reinterpret_cast<unsigned long long>(ud.ll);
    s.ud_offset = ud.offset;
    kd_progressed = 0;
  }
#ifdef KD_PROFILE
  if (tid < kProfSlots) s.cyc[tid] = 0;
  if (tid == 0) {
    s.lap_last = static_cast<long long>(__builtin_amdgcn_s_memtime());
    s.lap_on = 1;
  }
#endif
  if (search) FullClear<true>(tl, s);

  if (tid == 0) {
    KD_G UttMem *m = ToG(c.P().mem) + utt;
    if (do_init) {  // no chunk of any pool yet, nothing archived, nothing to recycle
      m->atok_next = 0; m->atok_end = 0; m->alink_next = 0; m->alink_end = 0;
      m->arena_chunk = -1; m->link_chunk = -1; m->arena_chunk_len = 0; m->link_chunk_len = 0;
      m->gc_frame = 0;
      m->tok.n_retired = 0; m->tok.n_free = 0;
      m->link.n_retired = 0; m->link.n_free = 0;
    }
    s.arena_next = do_init ? 0 : m->arena_next;
    s.arena_end = do_init ? 0 : m->arena_end;
    s.link_next = do_init ? 0 : m->link_next;
    s.link_end = do_init ? 0 : m->link_end;
  }
  if (do_init && c.lattice && tid == 0) ToG(c.P().lat.finalized)[utt] = 0;
  bool best_stale = false;
  if (resumed) {
    // (the best token's arc range is not carried over: the big tiers look it up, FrontierStats)
    if (tid == 0) s.f_best = f_best_in;
    best_stale = true;
  } else if (search && frame >= 0) {
    const long long fi = static_cast<long long>(utt) * (c.P().max_frames + 1) + frame;
    c.base_in = Uni(static_cast<long long>(ToG(c.P().frame_base)[fi]));
    c.n_in = Uni(static_cast<int>(ToG(c.P().frame_count)[fi]));
    FrontierStats(c);
  }
  __syncthreads();

  // Tier bookkeeping (all workgroup-uniform): is the frontier in LDS (left there by
  // a tier-1 frame), do the upper table halves hold tier-1 data, and is the best
  // token's arc range (the tier-2 prepass input) out of date.
  Table<true> t1{s, nullptr, s.bitmask, kT1Log2, kT1Slots - 1};
  bool front_lds = false, upper_dirty = false;
  auto row_of = [&](int f) {
    return ToG(ud.ll) + static_cast<long long>(f - ud.offset) * c.P().row_stride;
  };
  auto row_to_lds = [&](int f) {  // whole workgroup; the caller synchronises
    if (!c.ll_in_lds || f < 0 || f >= target) return;
    const KD_G float *row = row_of(f);
    for (int v = tid; v < c.P().num_cols; v += kBlock) s.ll[v] = row[v];
    LdsBarrier();
    if (c.P().log_softmax) NormalizeRow(s, c.P().num_cols);  // (logits: normalised here, once per staging)
    StoreRowTop(s, c.P().num_cols);  // (the label index's view of the row)
    if (tid == 0) s.top_valid = 1;
  };
  // At the top of the loop the per-frame counters are reset and s.ll holds the
  // frame's log-likelihood row: a frame boundary costs one barrier.
  ResetFrame<M>(s);
  if (err == 0) row_to_lds(frame);
  __syncthreads();
  auto fold = [&](int stat, int v) {  // a per-lane count of this frame into the launch's totals
    if (__ballot(v != 0) == 0) return;  // (most waves of a small frame have counted nothing)
    const int w = WaveSumInt(v);
    if ((tid & 63) == 0 && w) atomicAdd(&s.ustat[stat], static_cast<unsigned long long>(w));
  };
  // ---- PruneActiveTokens, owed and paid (lattice decoders).  The reference prunes every
  // prune_interval frames (lattice-simple-decoder.cc:57-59) to bound memory and the work of later
  // prunes; what FinalizeDecoding leaves does not depend on it (the exact final
  // pass recomputes every frame).  A prune at frame m reads and writes nothing of the frames
  // above m -- their tokens do not exist yet when the reference runs it -- so it can be run at any
  // LATER time with the same result, as long as the prunes run in order.  With prune_lazy the
  // frame loop only notes that a prune is owed (the owed prunes are the multiples m of the
  // interval with m - 1 > visited: no state of their own), and they are run, in order, when
  // something can tell the difference: an export of the un-finalized lattice, or a frame that
  // cannot get its token or link block -- the collector behind each prune frees the utterance's
  // old chunks, so the owed prunes are this decoder's garbage collection.
  [[maybe_unused]] auto first_owed = [&]() {  // (the interval is INT_MAX when pruning is off)
    const int iv = c.P().lat.prune_interval;
    return (visited + 1) / iv >= 0x7FFFFFFF / iv ? 0x7FFFFFFF : ((visited + 1) / iv + 1) * iv;
  };
  [[maybe_unused]] auto run_prunes = [&](int upto) -> uint32_t {  // every wave; prunes m <= upto
    uint32_t pe = 0;
    while (pe == 0) {
      const int m = first_owed();
      if (m > upto) break;
      __syncthreads();
      KD_LAP(s, 13);
      if (tid == 0) {  // (the collector takes archive chunks through the same cursors' owner)
        KD_G UttMem *um = ToG(c.P().mem) + utt;
        um->arena_next = s.arena_next; um->arena_end = s.arena_end;
        um->link_next = s.link_next; um->link_end = s.link_end;
        s.stat[KD_STAT_PRUNES] += 1;
      }
      __syncthreads();
      pe = Uni(PeriodicPrune(c.pp, utt, m, visited));
      visited = m - 1;
      __syncthreads();
      // the sweep used the LDS block: the table is cleared again, the frontier comes
      // back from the arena (this frame's log-likelihood row and the frame counters
      // lie beyond what the sweep touches)
      FullClear<true>(tl, s);
      front_lds = false;
      upper_dirty = false;
      KD_LAP(s, 23);
    }
    return pe;
  };
  if (tid == 0) kd_t_setup = static_cast<long long>(__builtin_amdgcn_s_memtime()) - kd_t_setup;
  int seg = kSegFinished;
  [[maybe_unused]] bool gc_retry = false;  // this frame found no room in a pool: the owed prunes first
  while (err == 0 && frame < target) {
    KD_PHASE(c);  // (nothing loaded through the parameters lives across a frame)
    [[maybe_unused]] bool prune_due = false, prune_now = false;
    if constexpr (M == 1) {
      prune_due = frame > 0 && frame % c.P().lat.prune_interval == 0 && frame - 1 > visited;
      // (owed prunes are paid when their frame comes after all once three quarters of a pool
      // THAT IS NOT TO GROW ANY FURTHER -- no_grow, or the batch's memory budget reached: the host
      // says which, gc_tokens / gc_links -- have been handed out: the collector behind a prune
      // needs room for the archive copies it makes, and every other utterance of the batch goes
      // on allocating until its own next prune, so the garbage collection must not wait for the
      // pool to be exhausted -- one lane looks at the cursors, every prune_interval frames.  A
      // pool that may still grow is grown instead: memory for time)
      prune_now = prune_due &&
                  (c.P().lat.prune_lazy == 0 || Decide([&]() {
                     CParams &q = c.P();
                     const unsigned long long tu = __hip_atomic_load(&q.ctl->tok_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.tok_base;
                     const unsigned long long lu = __hip_atomic_load(&q.ctl->link_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.link_base;
                     return (q.lat.gc_tokens != 0 && tu * 4 >= q.tok_cap * 3) ||
                            (q.lat.gc_links != 0 && lu * 4 >= q.link_cap * 3);
                   }));
    }
    // ---- does the utterance change hands here?  (workgroup-uniform; sizes of the decode and the
    // kinds of workgroups the launch has, never timings: which kind runs a frame changes nothing
    // of what the frame computes)
    if (c.P().sched.split != 0 && frame >= 0 && !gc_retry) {
      bool go = false;
      if constexpr (kKind == kKindLarge) go = c.n_in <= c.P().sched.mig_lo || (prune_now && c.P().sched.prune_small != 0);
      else go = c.n_in > c.P().sched.mig_hi && !(prune_now && c.P().sched.prune_small != 0);
      // (the rest of the decision is off the frames' path: it is reached once per hand-over)
      if (go && Decide([&]() { return HandOver(c.P()); })) {
        seg = kKind == kKindLarge ? kSegToSmall : kSegToLarge;
        break;
      }
    }
    if (tid == 0) kd_progressed = 1;  // (read behind the frame's barriers)
    if constexpr (M == 1) {
      // PruneActiveTokens every prune_interval frames, BEFORE the frame is processed like
      // the reference's (lattice-simple-decoder.cc:57-59) -- so a call that resumes on a
      // multiple of the interval (streaming in chunks of 1, 5, 25, ... frames) performs the
      // prune its predecessor left pending.  Lazily (the default: LatticeParams::prune_lazy)
      // the prune is only OWED here and performed when something can tell the difference
      // (run_prunes below).
      if (prune_now || gc_retry) {
        const uint32_t pe = run_prunes(frame);
        if (pe) { err |= pe; break; }
        if (gc_retry) {  // (the frame that found no room starts again from the arena)
          gc_retry = false;
          ResetFrame<M>(s);
          if (tid < 4) s.frame_cnt[tid] = 0;
          row_to_lds(frame);
          __syncthreads();
          best_stale = true;
        }
      }
    }
    KD_LAP(s, 13);
#if defined(KD_PROFILE) && defined(KD_PROF_NMIN)
    if (tid == 0) s.lap_on = c.n_in >= KD_PROF_NMIN && c.n_in < KD_PROF_NMAX;
#endif
#ifdef KD_PROF_SIZE
    // -DKD_PROFILE -DKD_PROF_SIZE (make knob; tools/prof_sizes.py): the frames' cycles by the
    // SIZE of their frontier -- classes 0, 1, 2-3, 4-7, ..., 512-1023, >= 1024: sums in buckets
    // 0..11, counts in 12..23 -- instead of by phase
    const int prof_n = c.n_in;
    long long prof_t0 = 0;
    if (tid == 0) prof_t0 = static_cast<long long>(__builtin_amdgcn_s_memtime());
#endif
    Counters cnt;
    bool small = false;
    if constexpr (!c.simple) small = frame >= 0 && c.n_in <= kT1Front;
    // A frontier in the upper quarter of the tier's range that follows a frame which queued
    // several times what the LDS part of the candidate queue holds goes to the big tiers at once
    // (a deterministic rule: sizes of the decode, not timings): its queue will be in global
    // memory mostly, its states will overflow the 4096-slot table, and handing a frame over
    // AFTER its filter costs more than either tier.
    if constexpr (kT1SoftQueue > 0) small = small && !(c.n_in > kT1SoftFront && prev_q_emit > kT1SoftQueue);
    uint32_t r = kRetry;
    int q_emit = 0;
    bool filtered = false;  // the emitting filter already ran (tier 1 did not fit)
    if (small) {
      const Team tm{};
      const bool have_next = c.ll_in_lds && frame + 1 < target;
      {
        const KD_G float *ll_next = row_of(have_next ? frame + 1 : frame);
        c.ll_row = row_of(frame);
        if (!front_lds) LoadFront(c, tm);
        // (two copies of the frame on purpose: in each the compiler knows where the
        // log-likelihood row is, and the copy for rows held in LDS loses the branch --
        // and the wait for outstanding vector-memory operations -- per arc)
        if (c.ll_in_lds) r = SmallFrame(c, t1, frame, cnt, tm, ll_next, have_next);
        else r = SmallFrame(c, t1, frame, cnt, tm, ll_next, have_next);
        if (tid == 0) s.ctl[frame & 1].r = r;
        LdsBarrier();  // nobody reads this frame's row any more
      }
      r = Uni(s.ctl[frame & 1].r);
      // (logits: the row the frame swapped in for the next one; a frame that is run again
      // below stages its rows afresh)
      if (c.P().log_softmax && r == 0 && have_next) {
        NormalizeRow(s, c.P().num_cols);
        StoreRowTop(s, c.P().num_cols);  // (the frame noted the top of the logits)
        if (tid == 0) s.top_valid = 1;
        LdsBarrier();
      }
      upper_dirty = true;
      best_stale = true;
      if (r == 0) {
        c.base_in = Uni(s.ctl[frame & 1].base_out);
        c.n_in = Uni(s.ctl[frame & 1].n_in);
        if constexpr (kT1SoftQueue > 0) prev_q_emit = Uni(s.ctl[frame & 1].q_emit);
        front_lds = c.n_in <= kT1Front;
        // (the frame's per-lane work counters go to the launch totals here: six wave sums a
        // frame, instead of six more registers per lane held across the whole frame loop)
        fold(KD_STAT_TOKENS_EXPANDED, cnt.tokens_expanded);
        fold(KD_STAT_EPS_EXPANSIONS, cnt.eps_exp);
        fold(KD_STAT_ARCS_SCANNED, cnt.arcs_scanned);
        fold(KD_STAT_EPS_ARCS_SCANNED, cnt.eps_arcs);
        fold(KD_STAT_ARCS_SKIPPED, cnt.arcs_skipped);
        fold(KD_STAT_INDEX_PROBES, cnt.idx_probes);
      } else if (r == kRedo) {
        // the queue overflowed: all of the frame again in the big path, which
        // needs this frame's row back (its place was taken by the next frame's)
        r = kRetry;
        cnt = Counters();
        ResetFrame<M>(s);
        row_to_lds(frame);
        __syncthreads();
      } else if (r == kRetry) {
        // more states than the small table holds: move the LDS part of the queue
        // next to the rest and replay all of it in the big table
        q_emit = Uni(s.ctl[frame & 1].q_emit);
        c.cutoff = Uni(s.ctl[frame & 1].cutoff);
        const int nl = q_emit < kQL ? q_emit : kQL;
        // (lattice: without the destination's arc range that tier-1 entries carry in the low
        // word of the key and the upper bits of the source: T1Put)
        constexpr unsigned long long kNkMask = M == 1 ? 0xFFFFFFFF00000000ull : ~0ull;
        constexpr uint32_t kSrMask = M == 1 ? kT1SrcMask : ~0u;
        for (int i = tid; i < nl; i += kBlock) {
          c.q_arc[i] = T1QArc(s)[i];
          c.q_src[i] = T1QSrc(s)[i] & kSrMask;
          c.q_nk[i] = T1QNk(s)[i] & kNkMask;
          if (c.lattice) c.q_ac[i] = T1QAc(s)[i];  // (allocated for the lattice decoders only)
          if (c.lattice) c.q_gr[i] = T1QGr(s)[i];
        }
        if constexpr (M == 1) {
          for (int i = kQL + tid; i < q_emit; i += kBlock) {
            c.q_src[i] &= kSrMask;
            c.q_nk[i] &= kNkMask;
          }
        }
        filtered = true;
        cnt.eps_exp = 0;
        cnt.eps_arcs = 0;
        fold(KD_STAT_ARCS_SKIPPED, cnt.arcs_skipped);  // (the emitting filter is not run again)
        fold(KD_STAT_INDEX_PROBES, cnt.idx_probes);
      }
    }
    if (r == kRetry) {
      // frames that need the big tables run in a function of their own
      r = BigFrame<M>(c.pp, utt, frame, c.base_in, c.n_in, c.cutoff, q_emit,
                      (filtered ? 1 : 0) | (upper_dirty ? 2 : 0) | (best_stale ? 4 : 0),
                      cnt.tokens_expanded, cnt.arcs_scanned);
      front_lds = false;
      upper_dirty = false;
      best_stale = false;
      r = Uni(r);
      if (r == 0) {
        c.base_in = Uni(s.base_out);
        c.n_in = Uni(c.simple ? s.n_act : s.n_live);
        if constexpr (kT1SoftQueue > 0) prev_q_emit = Uni(s.ctl[0].pad);  // (BigFrame leaves it there)
      }
      // the next frame's start state; every wave's token stores have landed before a
      // small team reads them
      __syncthreads();
      ResetFrame<M>(s);
      if (tid < 4) s.frame_cnt[tid] = 0;
      row_to_lds(frame + 1);
      __syncthreads();
    }
    if (r != 0) {
      if constexpr (M == 1) {
        // No room for this frame's tokens or links in a pool that is not to grow, and prunes are
        // owed: they are run now -- the collector gives the utterance its old chunks back -- and
        // the frame, which was not committed, starts again from the arena.  (A pool that may grow:
        // the utterance stops here and the host grows it, as ever.)
        if ((((r & KD_UTT_ARENA_FULL) != 0 && c.P().lat.gc_tokens != 0) ||
             ((r & KD_UTT_LINKS_FULL) != 0 && c.P().lat.gc_links != 0)) && first_owed() <= frame) {
          gc_retry = true;  // (at the top of the loop, where the prunes are run)
          continue;
        }
      }
      if constexpr (kKind == kKindSmall) {
        // A frame that outgrew what this geometry's tables hold (its global table's worklist mask
        // lives in a third of the LDS) was not committed: it is a large workgroup's.
        if (c.P().sched.split != 0 && r == KD_UTT_TABLE_FULL && Decide([&]() { return OtherKindRuns(c.P()); })) {
          seg = kSegToLarge;
          break;
        }
      }
      err |= r;
      break;
    }
#ifdef KD_PROF_SIZE
    if (tid == 0) {
      int cls = 0;
      while ((1 << cls) <= prof_n && cls < 11) ++cls;
      s.cyc[cls] += static_cast<long long>(__builtin_amdgcn_s_memtime()) - prof_t0;
      s.cyc[12 + cls] += 1;
    }
#endif
    if (kKind == kKindSmall && frame >= 0 && tid == 0) s.stat[KD_STAT_SMALL_WG_FRAMES] += 1;
    frame++;
  }
  if constexpr (M == 1) {
    // GetRawLattice of the un-finalized lattice shows what the prunes the reference has run by
    // now left: those at frames m < frame (the one at m == frame runs at the top of the NEXT
    // frame's iteration, :57-59).  FinalizeDecoding needs none of them.
    if (err == 0 && seg == kSegFinished && frame == target && frame > 0 &&
        (ud.flags & (kUttExport | kUttExportNoFinal)) != 0 && (ud.flags & kUttFinalize) == 0 &&
        Uni(ToG(c.P().lat.finalized)[utt]) == 0) {
      const uint32_t pe = run_prunes(frame - 1);
      if (pe) err |= pe;
    }
  }

  __syncthreads();
  const int frames_done = frame < 0 ? -1 : frame;
  if (tid == 0) {
    ToG(c.P().frames_decoded)[utt] = frames_done;
    ToG(c.P().status)[utt] = err;
    KD_G UttMem *m = ToG(c.P().mem) + utt;
    m->arena_next = s.arena_next;
    m->arena_end = s.arena_end;
    m->link_next = s.link_next;
    m->link_end = s.link_end;
    m->visited = visited;
    if (seg != kSegFinished) {
      // what the frame loop carries from frame to frame (everything else of the utterance is
      // in global memory already); the utterance's UttCtl goes along so that the taker does
      // not read pinned host memory again
      KD_G UttResume *rs = ToG(c.P().sched.resume) + utt;
      rs->ll = ud.ll;
      rs->offset = ud.offset;
      rs->target = ud.target;
      rs->flags = ud.flags & ~kUttInit;
      rs->frame = frame;
      rs->base_in = c.base_in;
      rs->n_in = c.n_in;
      rs->prev_q_emit = prev_q_emit;
      rs->f_best = s.f_best;
      rs->visited = visited;
    }
  }
  if (tid == KD_STAT_HANDOVERS && seg != kSegFinished) s.stat[tid] += 1;
  if (tid < KD_NUM_STATS) {
    KD_G long long *st = ToG(c.P().stats) + static_cast<long long>(utt) * kStatStride;
    const long long was = do_init ? 0 : st[tid];
    if (tid == KD_STAT_MAX_TOKENS) st[tid] = s.stat[tid] > was ? s.stat[tid] : was;
    else st[tid] = was + s.stat[tid];
  }
#ifdef KD_PROFILE
  if (tid < kProfSlots) {
    KD_G long long *cy = ToG(c.P().stats) + static_cast<long long>(utt) * kStatStride + KD_NUM_STATS;
    cy[tid] = (do_init ? 0 : cy[tid]) + s.cyc[tid];
  }
#endif
  if (seg != kSegFinished) return seg;
  // ---- the tail (lattice decoders): the searching is done, the LDS is free
  UttOut o;
  o.frames_decoded = frames_done;
  o.status = err;
  o.sizes[0] = 0; o.sizes[1] = 0; o.sizes[2] = 0; o.sizes[3] = -1;
  o.out_off = -1;
  o.final_rel = 0.0f;
  o.phase = kPhaseDecoding;
  if constexpr (M == 1) {
    if (err == 0 && frame == target && frame >= 0 &&
        (ud.flags & (kUttFinalStats | kUttFinalize | kUttExport | kUttExportNoFinal))) {
      __syncthreads();  // (every wave is done with the search's LDS)
      const uint32_t e = Uni(LatticeTail(c.pp, utt, frame, ud.flags));
      __syncthreads();
      if (e && tid == 0) ToG(c.P().status)[utt] = e;
      const TailShared &ts = TailLds();
      o.status = e;
      o.sizes[0] = ts.r_sizes[0]; o.sizes[1] = ts.r_sizes[1];
      o.sizes[2] = ts.r_sizes[2]; o.sizes[3] = ts.r_sizes[3];
      o.out_off = ts.r_out_off;
      o.final_rel = ts.r_final_rel;
      if (ts.r_arena_order) o.phase |= kPhaseArenaOrder;
    }
    if (Uni(ToG(c.P().lat.finalized)[utt]) != 0) o.phase |= kPhaseFinalized;
  } else {
    // 1-best decoders: ReachedFinal and GetBestPath of what has been decoded so far
    if (err == 0 && frame == target && frame >= 0 && (ud.flags & kUttBestPath)) {
      __syncthreads();
      const uint32_t e = Uni(PathTail(c.pp, utt, frame, ud.flags));
      __syncthreads();
      if (e && tid == 0) ToG(c.P().status)[utt] = e;
      const TailShared &ts = TailLds();
      o.status = e;
      o.sizes[0] = ts.r_sizes[0]; o.sizes[1] = ts.r_sizes[1];
      o.sizes[2] = ts.r_sizes[2]; o.sizes[3] = ts.r_sizes[3];
      o.out_off = ts.r_out_off;
      o.final_rel = ts.r_final_rel;
    }
  }
  if (tid == 0) c.P().out[utt] = o;
  return kSegFinished;
}

// ---- the queues of utterances that wait for a workgroup of one kind (kd_device.h, SchedCtl) ----
// Rings of `ring_cap` words (utterance + 1; 0: empty), one per XCD and kind; an utterance is in
// at most one of them, so a ring never holds more than `batch` entries.  Call from one lane.
__device__ __forceinline__ uint32_t *SchedRing(CParams &p, int xcd, int kind) {
  return p.sched.ring + static_cast<long long>(xcd * 2 + kind) * p.sched.ring_cap;
}
__device__ __forceinline__ void QueuePush(CParams &p, int xcd, int kind, int utt) {
  const uint32_t pos = atomicAdd(&p.sched.ctl->q[xcd][kind].tail, 1u);
  uint32_t *slot = SchedRing(p, xcd, kind) + (pos & static_cast<uint32_t>(p.sched.ring_cap - 1));
  // (the slot's previous entry, a lap ago, has been taken long since: the loop is a formality)
  while (atomicCAS(slot, 0u, static_cast<uint32_t>(utt) + 1u) != 0u) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ int QueuePop(CParams &p, int xcd, int kind) {  // -1: nothing waits
  SchedQueue *q = &p.sched.ctl->q[xcd][kind];
  uint32_t h = AgentLoad(&q->head);
  while (true) {
    if (static_cast<int32_t>(AgentLoad(&q->tail) - h) <= 0) return -1;
    const uint32_t seen = atomicCAS(&q->head, h, h + 1u);
    if (seen == h) break;
    h = seen;
  }
  uint32_t *slot = SchedRing(p, xcd, kind) + (h & static_cast<uint32_t>(p.sched.ring_cap - 1));
  uint32_t v;
  // (position h is this lane's alone; its entry may still be on its way)
  while ((v = atomicExch(slot, 0u)) == 0u) __builtin_amdgcn_s_sleep(2);
  return static_cast<int>(v) - 1;
}

template <int M>
// (HIP's second bound is waves per SIMD: the workgroups sharing a CU spread over four)
#if KD_WG_PER_CU > 1
__attribute__((amdgpu_waves_per_eu((KD_WG_PER_CU * (KD_BLOCK / 64) + 3) / 4, (KD_WG_PER_CU * (KD_BLOCK / 64) + 3) / 4)))
#endif
__global__ __launch_bounds__(kBlock, (KD_WG_PER_CU * (KD_BLOCK / 64) + 3) / 4) void SearchKernel(const SearchParams p_by_value) {
  Shared &s = Lds();
  (void)p_by_value;  // the only argument: it is the kernel-argument segment
  CParams &p = *(CParams *)__builtin_amdgcn_kernarg_segment_ptr();
  // Persistent workgroups and a work queue: each takes the next utterance of `order` (longest
  // first) until none is left, so a CU that finishes a short utterance goes on with another one
  // -- of any size of batch -- instead of idling until the launch's slowest utterance is done.
  // With two kinds of workgroups in the launch (sched.split) an utterance may leave a workgroup
  // at a frame boundary for a queue of the other kind, and a workgroup first looks for an
  // utterance that waits for its kind; the launch is over when every utterance is where it
  // was sent.  The queues are per XCD (kd_device.h): this workgroup's is the one it RUNS on.
  const bool split = p.sched.split != 0;
  const int xcd = ThisXcd();
  if (split && threadIdx.x == 0) {
    atomicAdd(&p.sched.ctl->live[xcd], 1ull << 32);
    atomicAdd(&p.sched.ctl->n_reg[xcd][kKind], 1u);
  }
  bool fresh_left = true;  // (lane 0's: positions of `order` may still be had)
  // (lane 0's clocks: kd_device.h SchedProf)
  if (threadIdx.x < 8) kd_prof[threadIdx.x] = 0;
  while (true) {
    __syncthreads();  // the previous utterance is finished by every wave
    if (threadIdx.x == 0) {
      const long long t0 = static_cast<long long>(__builtin_amdgcn_s_memtime());
      long long t_found = t0;
      int item = -1, take = kTakeFresh;
      while (true) {
        t_found = static_cast<long long>(__builtin_amdgcn_s_memtime());
        if (split) {
          int u = QueuePop(p, xcd, kKind);
          if (u >= 0) { item = u; take = kTakeResumed; break; }
        }
        if (fresh_left) {
          const int i = static_cast<int>(atomicAdd(&p.ctl->work_ctr, 1u) - p.work_base);
          if (i < p.batch) {
            item = ToG(p.order)[i];
            if (split) atomicAdd(&p.sched.ctl->live[xcd], 1ull);  // (this XCD's from now on)
            break;
          }
          fresh_left = false;  // (one failed attempt per workgroup: the host counts on it)
          if (!split) break;
          continue;
        }
        if (kKind == kKindLarge || AgentLoad(&p.sched.ctl->n_reg[xcd][kKindLarge]) == 0u) {
          // Nothing waits for this kind: a large workgroup takes what waits for a small one
          // (better a small frame on a whole CU than an idle CU); a small one takes what waits for
          // a large one only when the XCD's large workgroups have all left (below).
          const int u = QueuePop(p, xcd, 1 - kKind);
          if (u >= 0) { item = u; take = kTakeStolen; break; }
        }
        if (AgentLoad(&p.sched.ctl->n_finished) >= static_cast<uint32_t>(p.batch)) break;
        // More workgroups than unfinished utterances on this XCD: this one leaves (SchedCtl::live).
        if (p.sched.stay == 0) {
          unsigned long long *lw = &p.sched.ctl->live[xcd];
          unsigned long long w = __hip_atomic_load(lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool left = false;
          while ((w >> 32) > (w & 0xFFFFFFFFull)) {
            const unsigned long long seen = atomicCAS(lw, w, w - (1ull << 32));
            if (seen == w) { left = true; break; }
            w = seen;
          }
          if (left) {
            atomicSub(&p.sched.ctl->n_reg[xcd][kKind], 1u);
            break;
          }
        }
        __builtin_amdgcn_s_sleep(64);
      }
      kd_work_item = item;
      kd_work_take = take;
      // what the utterance's previous workgroup stored is in this XCD's L2; this CU's L1 may
      // hold older contents of those lines
      if (take != kTakeFresh) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const long long t1 = static_cast<long long>(__builtin_amdgcn_s_memtime());
      kd_prof[0] += t_found - t0;
      kd_prof[1] += t1 - t_found;
      if (item >= 0) kd_prof[4 + take] += 1;
      kd_t_setup = t1;
      kd_prof[7] = t1;
    }
    __syncthreads();
    const int item = Uni(kd_work_item), take = Uni(kd_work_take);
    if (item < 0) break;
    const int seg = RunUtterance<M>(p, s, item, take != kTakeFresh);
    if (threadIdx.x == 0) {
      const long long t2 = static_cast<long long>(__builtin_amdgcn_s_memtime());
      // (kd_t_setup: the stamp until the utterance's first frame, the difference from there on;
      // an utterance with no frame to run leaves the stamp)
      const long long su = kd_t_setup == kd_prof[7] ? 0 : kd_t_setup;
      kd_prof[2] += su;
      kd_prof[3] += t2 - kd_prof[7] - su;
    }
    if (split) {
      __syncthreads();  // every wave's stores of this utterance have left the CU (vmcnt(0))
      if (threadIdx.x == 0) {
        if (seg == kSegFinished) {
          atomicAdd(&p.sched.ctl->live[xcd], ~0ull);  // (one utterance fewer is bound to this XCD)
          atomicAdd(&p.sched.ctl->n_finished, 1u);
        }
        else QueuePush(p, xcd, seg == kSegToSmall ? kKindSmall : kKindLarge, item);
      }
    }
  }
  if (split && threadIdx.x == 0) {
    SchedProf *pf = p.sched.prof;
    atomicAdd(&pf->idle[kKind], static_cast<unsigned long long>(kd_prof[0]));
    atomicAdd(&pf->setup[kKind], static_cast<unsigned long long>(kd_prof[2]));
    atomicAdd(&pf->busy[kKind], static_cast<unsigned long long>(kd_prof[3]));
    atomicAdd(&pf->wait_pop[kKind], static_cast<unsigned long long>(kd_prof[1]));
    atomicAdd(&pf->n_fresh[kKind], static_cast<unsigned long long>(kd_prof[4 + kTakeFresh]));
    atomicAdd(&pf->n_resumed[kKind], static_cast<unsigned long long>(kd_prof[4 + kTakeResumed]));
    atomicAdd(&pf->n_stolen[kKind], static_cast<unsigned long long>(kd_prof[4 + kTakeStolen]));
    atomicAdd(&pf->n_workgroups[kKind], 1ull);
  }
  // The last workgroup to leave tells the host where the pools' cursors stand (every
  // other workgroup's allocations are behind its n_done increment) and leaves the
  // scheduler's words zero for the next launch.
  if (threadIdx.x == 0) {
    __threadfence_system();
    const uint32_t left = atomicAdd(&p.ctl->n_done, 1u) - p.done_base;
    if (left == static_cast<uint32_t>(p.sched.total_wgs) - 1u) {
      if (split) {
        uint32_t *w = reinterpret_cast<uint32_t *>(p.sched.ctl);
        for (size_t i = 0; i < sizeof(SchedCtl) / 4; ++i)
          __hip_atomic_store(w + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      HostTops t;
      t.tok_top = atomicAdd(&p.ctl->tok_top, 0ull);
      t.link_top = atomicAdd(&p.ctl->link_top, 0ull);
      t.out_top = atomicAdd(&p.ctl->out_top, 0ull);
      t.launches = p.tops->launches + 1;
      *p.tops = t;
      __threadfence_system();
    }
  }
}

}  // namespace

#ifdef KD_GEOM_SMALL
#define KD_EXPORT(name) name##Small
#else
#define KD_EXPORT(name) name
#endif
size_t KD_EXPORT(SearchLdsBytes)() { return sizeof(Shared); }
int KD_EXPORT(SearchMaxLdsCols)() { return kVmax; }
int KD_EXPORT(SearchWorkgroupsPerCu)() { return KD_WG_PER_CU; }
#ifdef KD_GEOM_SMALL
int SearchT1FrontSmall() { return kT1Front; }
#endif

namespace {
template <int M>
hipError_t LaunchMode(const SearchParams &p, int workgroups, hipStream_t stream) {
  hipLaunchKernelGGL(SearchKernel<M>, dim3(workgroups), dim3(kBlock), 0, stream, p);
  return hipGetLastError();
}
}  // namespace

hipError_t KD_EXPORT(LaunchSearch)(const SearchParams &p, int workgroups, hipStream_t stream) {
  if (p.lattice) return LaunchMode<1>(p, workgroups, stream);
#ifndef KD_ONLY_LATTICE  // (tools: a build of the lattice decoders' kernel alone compiles in a third of the time)
  if (p.simple) return LaunchMode<2>(p, workgroups, stream);
  return LaunchMode<0>(p, workgroups, stream);
#else
  return hipErrorNotSupported;
#endif
}

}  // namespace kd
