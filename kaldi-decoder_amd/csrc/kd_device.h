// kaldi-decoder_amd/csrc/kd_device.h -- device-side data layout shared by the
// search kernel, the traceback kernels and the host C-ABI implementation.
//
// HBM layout (DESIGN.md "Data layout"):
//   graph    StateRec[S] (16 B)  + ArcPacked[A] (16 B) + olabel[A] (4 B, output side only),
//            arcs of a state contiguous, emitting arcs first (stable) so the emitting and
//            epsilon passes each read one contiguous, 16-byte-aligned range per token; an
//            arc names its destination's arc range, so a token knows its arcs from the
//            arc that created it.
//   tokens   append-only records of the whole batch as parallel arrays (state,
//            cost, prev, arc / extra): one pool shared by the utterances; an
//            utterance takes a chunk at a time (one atomic) and cuts one contiguous
//            block per frame from it.  The link pool works the same way.
//   frames   frame_base/frame_count[utt][frame_plus_one] index the blocks.
#ifndef KD_DEVICE_H_
#define KD_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kd {

struct __attribute__((aligned(16))) StateRec {
  int32_t arc_begin;  // first (emitting) arc
  int32_t emit_end;   // one past the last emitting arc == first epsilon arc
  int32_t arc_end;    // one past the last arc
  float final_cost;   // +inf == not final
};

// An arc as the search reads it: 16 bytes, one global_load_dwordx4.  Besides what the
// reference's ArcIterator yields (ilabel, weight, nextstate; the olabel, which only the export
// and the traceback read, lives in a side array) it carries what the search would otherwise
// fetch from the DESTINATION's state record -- a dependent round trip per epsilon round and one
// for the new frontier: the destination's first arc and its emitting / epsilon degrees.
struct __attribute__((aligned(16))) ArcPacked {
  uint32_t lab;       // bits 0..19 ilabel | 20..27 destination's emitting degree | 28..31 its epsilon degree
  uint32_t dst_arcs;  // the destination's first arc (StateRec::arc_begin)
  float weight;
  uint32_t next;      // bits 0..30 next state, bit 31: that state has epsilon arcs
};
constexpr int kIlabelBits = 20;
constexpr int kIdxOffBits = 12;  // label-index keys: ilabel << 12 | offset << 1 | more (kd_label_index.h)
constexpr int32_t kMaxIlabel = (1 << kIlabelBits) - 1;
// The largest 20-bit value is not a label but an ESCAPE: the arc's input label is this or
// larger and stands in a side array of full int32 labels (SearchParams::big_ilabel, parallel to
// the arcs; allocated only for a graph that holds such a label -- an H over the transition-ids
// of a large acoustic model, not a CTC token graph).  The reference takes any int32
// (faster-decoder.cc:204-209).  Such a graph's matrices have more than 2^20 columns, so their
// rows are never staged in LDS: only the paths that read a row in place look the label up.
constexpr int32_t kIlabelEscape = kMaxIlabel;
// degrees that do not fit are stored as these, and looked up in the state record as before
constexpr uint32_t kEmitDegEscape = 255u, kEpsDegEscape = 15u;
// the 12 degree bits as one field ("degs": emitting degree | epsilon degree << 8)
__host__ __device__ inline uint32_t PackDegs(int64_t emit_deg, int64_t eps_deg) {
  const uint32_t e = emit_deg < kEmitDegEscape ? static_cast<uint32_t>(emit_deg) : kEmitDegEscape;
  const uint32_t n = eps_deg < kEpsDegEscape ? static_cast<uint32_t>(eps_deg) : kEpsDegEscape;
  return e | n << 8;
}
__host__ __device__ inline uint32_t DegsEmit(uint32_t degs) { return degs & 255u; }
__host__ __device__ inline uint32_t DegsEps(uint32_t degs) { return (degs >> 8) & 15u; }
// ... and in registers
struct ArcRec {
  int32_t ilabel;
  uint32_t dst_arcs;
  uint32_t dst_degs;
  float weight;
  uint32_t next;
};

constexpr uint32_t kStateMask = 0x7FFFFFFFu;
constexpr uint32_t kHasEps = 0x80000000u;
constexpr uint32_t kKeyEmpty = 0xFFFFFFFFu;
constexpr uint32_t kArcNone = 0xFFFFFFFFu;
constexpr uint32_t kSrcEps = 0x80000000u;  // back-pointer source is a table slot
constexpr uint32_t kSrcNone = 0xFFFFFFFFu;
constexpr unsigned long long kCostDead = 0xFFFFFFFFFFFFFFFFull;
constexpr int kProfSlots = 24;   // diagnostic cycle buckets (KD_PROFILE builds)
constexpr int kStatStride = 48;  // KD_NUM_STATS counters + kProfSlots, padded

// ---- launch control ------------------------------------------------------------
// No copy engine and no copy kernel takes part in a launch.  The per-utterance inputs
// (UttCtl, the work-queue order) and outputs (UttOut, the pools' cursors) live in ONE block
// of pinned host memory that the GPU addresses directly: workgroups read their UttCtl
// from it and write their UttOut into it, and exported lattices go straight into a
// pinned output pool.  The only device-resident control words are the atomically
// updated counters of BatchCtl; they are never reset -- every launch is told the value
// each one has when it starts (`*_base` in SearchParams: the host knows them, the last
// workgroup of a launch publishes them) and works with differences.
struct UttCtl {
  const float *ll;     // row 0 of this utterance's log-likelihood matrix
  int32_t num_rows;
  int32_t offset;      // frame index of row 0 (DecodableCtc offset)
  int32_t target;      // decode until frames_decoded == target
  uint32_t flags;      // kUtt* below
};
constexpr uint32_t kUttInit = 1u;        // InitDecoding before anything else
constexpr uint32_t kUttFinalStats = 2u;  // ComputeFinalCosts / FinalRelativeCost only
constexpr uint32_t kUttFinalize = 4u;    // FinalizeDecoding when the target is reached
constexpr uint32_t kUttExport = 8u;      // GetRawLattice into the output pool
constexpr uint32_t kUttExportNoFinal = 16u;  // ... with use_final_probs == false
constexpr uint32_t kUttBestPath = 32u;       // 1-best decoders: ReachedFinal + GetBestPath into the pool
constexpr uint32_t kUttPathNoFinal = 64u;    // ... with use_final_probs == false

struct BatchCtl {
  uint32_t work_ctr;             // work queue: positions of `order` handed out, ever
  uint32_t n_done;               // workgroups that have left, ever
  unsigned long long tok_top;    // token pool: records handed out since the pool's base
  unsigned long long link_top;   // link pool
  unsigned long long out_top;    // output pool: 8-byte words
};
// what the last workgroup of a launch leaves in pinned memory for the host
struct HostTops {
  unsigned long long tok_top, link_top, out_top, launches;
};

// ---- two kinds of persistent workgroups in one call (round 6) ---------
// A call's utterances are decoded by two kernels that run side by side on disjoint sets of
// CUs (CU-masked streams): "large" workgroups -- one per CU, the whole CU's LDS: the geometry
// every frame fits -- and "small" ones, three to a CU, whose chains of dependent round trips
// overlap.  An utterance changes hands at a frame boundary when the size of the frontier its
// next frame starts from crosses a threshold (hysteresis: mig_lo / mig_hi): the workgroup that
// has it stores what the frame loop carries in registers and LDS (UttResume; everything else of
// an utterance already lives in global memory) and pushes the utterance onto a device queue of
// the other kind.  Queues are per XCD and indexed by the XCD the pushing workgroup RUNS on
// (HW_REG_XCC_ID): an utterance stays on one XCD for the whole launch, so that what one
// workgroup stored is in the L2 the next one reads through -- a hand-over needs the storing
// waves' s_waitcnt vmcnt(0) and an L1 invalidate at the taker, never an L2 write-back.
constexpr int kMaxXcd = 8;
constexpr int kKindSmall = 0, kKindLarge = 1;
struct SchedQueue {
  uint32_t head, tail;  // consumed / produced, since the launch began
  uint32_t pad[14];     // (a 64-byte line of its own)
};
struct SchedCtl {
  SchedQueue q[kMaxXcd][2];      // [xcd][kind]
  uint32_t n_reg[kMaxXcd][2];    // workgroups of each kind that are at work, per XCD
  // per XCD: workgroups at work << 32 | utterances bound to the XCD that are not finished.  A
  // workgroup that finds nothing to do leaves the launch for good once the XCD has more
  // workgroups than unfinished utterances (one compare-and-swap on this word decides it, so the
  // last utterances always keep a workgroup each): the CUs it held go to the next call's kernels
  // instead of idling through this call's tail.
  unsigned long long live[kMaxXcd];
  uint32_t n_finished;           // utterances that have reached their target (or stopped) in this launch
  uint32_t pad[15];
};  // all zero between launches: the last workgroup to leave a launch zeroes it
// Where the workgroups' time went, summed over the launches of a batch (never reset; shader-clock
// cycles of lane 0; kd_batch_sched_profile): per kind -- looking for an utterance / taking one
// up (RunUtterance's entry to its first frame) / running its frames and tail, and the numbers of
// utterances taken up fresh, resumed, stolen.
struct SchedProf {
  unsigned long long idle[2], setup[2], busy[2], wait_pop[2];
  unsigned long long n_fresh[2], n_resumed[2], n_stolen[2], n_workgroups[2];
};
// what the frame loop carries across a frame boundary
struct UttResume {
  const float *ll;                // the utterance's UttCtl, read from pinned host memory once per launch
  int32_t offset, target;
  uint32_t flags;
  int32_t frame;                  // the next frame to decode
  long long base_in;              // its frontier: an arena block ...
  int32_t n_in;
  int32_t prev_q_emit;
  unsigned long long f_best;      // ... and its best cost (key)
  int32_t visited;
  int32_t moves;                  // hand-overs of this utterance in this launch
};
struct SchedParams {
  SchedCtl *ctl;                  // device memory
  SchedProf *prof;                // device memory
  uint32_t *ring;                 // [kMaxXcd][2][ring_cap] entries utt + 1 (0: empty)
  UttResume *resume;              // [batch]
  int32_t ring_cap;               // a power of two >= batch
  int32_t wg_base;                // this kernel's first per-workgroup scratch set
  int32_t total_wgs;              // workgroups of both kernels of this launch
  int32_t mig_lo, mig_hi;         // large -> small at n_in <= mig_lo; small -> large at n_in > mig_hi
  int32_t prune_small;            // 1: periodic pruning is handed to the small kind
  int32_t stay;                   // 1: no workgroup leaves before the call's last utterance is done
  int32_t split;                  // 0: one kernel, no hand-overs (the kernel of rounds 1-5)
};

struct UttOut {
  int32_t frames_decoded;
  uint32_t status;
  int32_t sizes[4];       // exported lattice: states, arcs, finals, start state
  long long out_off;      // its first 8-byte word in the output pool (-1: not exported)
  float final_rel;        // FinalRelativeCost
  int32_t phase;          // kPhase* below
};
constexpr int32_t kPhaseDecoding = 0, kPhaseFinalized = 1;
// ... ORed with: the exported lattice is in arena order, four words an arc (the fourth names
// the graph arc): the host puts it into the canonical order (kd_api.hip, CanonicalizeExport)
constexpr int32_t kPhaseArenaOrder = 2;

// Per-utterance allocation state, device resident.  Frames take contiguous blocks from
// the chunk of each pool the utterance is filling (its "nursery"); a block that does not
// fit takes a new chunk -- one the utterance has freed itself, or else the next of the
// pool (one atomic on BatchCtl::tok_top / link_top).  Lattice decoders with periodic
// pruning also keep an "archive": what survives of frames older than prune_interval is
// moved there (compacted), and nursery chunks that hold nothing but such frames go on
// the utterance's own free list, so an utterance's footprint follows its pruned lattice
// plus a window of recent frames, not its length.
constexpr int kChunkRing = 48;
struct ChunkRing {
  int32_t n_retired, n_free;
  int32_t retired_start[kChunkRing];  // full nursery chunks, oldest first ...
  int32_t retired_last[kChunkRing];   // ... and the last frame (plus one) that has a block in each
  int32_t free_start[kChunkRing];     // chunks ready to be used again
};
struct UttMem {
  long long arena_next, arena_end;   // token nursery: the chunk being filled
  long long link_next, link_end;     // link nursery
  long long atok_next, atok_end;     // token archive
  long long alink_next, alink_end;   // link archive
  int32_t arena_chunk, link_chunk;   // start of the nursery chunks (-1: none yet) ...
  int32_t arena_chunk_len, link_chunk_len;  // ... and their sizes
  int32_t visited;    // frames <= this have been through a periodic sweep
  int32_t gc_frame;   // frames < this are in the archive
  ChunkRing tok, link;
};

struct LatticeParams {
  float lattice_beam;
  int32_t prune_interval;    // PruneActiveTokens every this many frames (lattice-simple-decoder.cc:57-59)
  float prune_scale;         // ... with delta = lattice_beam * prune_scale
  int32_t gc_tokens, gc_links;  // 1: that pool is not to grow any further (no_grow, or it has reached the
                             // batch's memory budget): owed prunes are its garbage collection
  int32_t prune_lazy;        // 1: a prune is only OWED when its frame comes, and run -- in order, with the
                             // same result -- when the un-finalized lattice is exported or a frame
                             // finds no room in the pools (kd_search.hip run_prunes)
  float *tok_extra;
  int32_t *tok_newid;        // [token pool] state id in the exported lattice, -1 if pruned
  uint32_t *scratch;         // [token pool] per-frame extra-cost keys beyond LDS
  // per utterance
  float *final_best_cost;
  float *final_rel;
  int32_t *any_final;
  int32_t *finalized;        // FinalizeDecoding has run
  // per utterance and frame [max_frames + 1]: first exported state id / arc of the frame
  int32_t *frame_state_off;
  int32_t *frame_arc_off;
  // output pool, 8-byte words: per utterance  states {frame, graph state} x ns |
  // finals {state id, weight} x nf | arcs {src, dst, ilabel, olabel, graph, ac} x na
  unsigned long long *out;   // (pinned host memory)
  unsigned long long out_cap;
  // a device buffer the caller lent (kd_batch_set_output_buffer) or null: the staged export
  // writes every word there too, at the same offsets -- the host reads its arrays from the
  // pinned pool as ever (no copy back through a blit kernel that would have to wait for a CU
  // among the persistent search workgroups: measured 18 ms a call), the multi-GPU gather
  // hands the device copy to RCCL
  unsigned long long *out2;
};

struct SearchParams {
  // graph
  const StateRec *states;
  const ArcPacked *arcs;
  const int32_t *arc_olabel;  // [arcs] read by GetRawLattice / GetBestPath only
  // label index (kd_label_index.h): parallel to the arcs; a state's emitting range holds the keys
  // (ilabel, offset of the arc in the range) as an implicit 5-ary search tree -- so a token finds
  // its arcs of ONE ilabel in a few dependent 16-byte loads instead of reading all its arcs
  const uint32_t *lab_idx;
  const int32_t *big_ilabel;  // [arcs] or null: the labels of the arcs whose 20-bit field holds kIlabelEscape
  float emit_w_min;           // the lowest weight of any emitting arc of the graph (NaN: unknown)
  int32_t start;
  // FasterDecoderOptions
  float beam;
  int32_t max_active;
  int32_t min_active;
  float beam_delta;
  // token pool (shared by the batch's utterances)
  int32_t *tok_state;
  double *tok_cost;
  int32_t *tok_prev;
  int32_t *tok_arc;
  unsigned long long tok_cap;
  int32_t tok_chunk;              // records an utterance takes from the pool at a time
  // frame index, per utterance [max_frames + 1]
  long long *frame_base;
  int32_t *frame_count;
  int32_t max_frames;
  // per-utterance state
  int32_t *frames_decoded;
  uint32_t *status;
  UttMem *mem;
  long long *stats;  // [batch][kStatStride]
  // per-WORKGROUP spill table (global memory) and scratch, g_slots entries each
  uint32_t *g_slot;  // 16-byte slots {cost key lo, hi, state key, winning arc}
  uint32_t *g_used;
  uint32_t *g_wl;    // two worklists
  // per-workgroup candidate log, 2 * g_slots entries (both table modes)
  uint32_t *g_log_h;
  uint32_t *g_log_arc;
  uint32_t *g_log_src;
  unsigned long long *g_log_nk;
  int32_t g_log2_slots;
  // launch control (utt, order, out, tops: pinned host memory; ctl: device memory)
  int32_t batch;
  const UttCtl *utt;
  const int32_t *order;      // work queue: utterance ids, longest first
  BatchCtl *ctl;
  UttOut *out;
  HostTops *tops;
  uint32_t work_base, done_base;                   // BatchCtl's counters when this launch starts
  unsigned long long tok_base, link_base, out_base;  // ... and where each pool's index 0 is
  int32_t num_cols;
  long long row_stride;
  // lattice mode (LatticeSimpleDecoder / LatticeFasterDecoder semantics)
  int32_t lattice;           // 0: FasterDecoder (double costs), 1: lattice decoders (float costs)
  int32_t simple;            // 1: SimpleDecoder semantics (plain beam, prune after closure)
  int32_t log_softmax;       // 1: the matrices hold logits; each row is normalised (x - logsumexp(x))
                             // when it is staged in LDS (kd_batch_set_log_softmax)
  float *g_log_ac;           // queue: acoustic cost (emitting) / source cost at generation (epsilon)
  float *g_log_gr;           // queue: graph cost of the arc (lattice mode)
  float *tok_extra;          // [token pool] extra_cost of each token
  int32_t *link_src;         // link pool
  int32_t *link_dst;         // -1 once pruned
  int32_t *link_arc;
  float *link_ac;
  float *link_graph;
  float *link_delta;  // (src.tot + ac + graph) - dst.tot, the constant of lattice-simple-decoder.cc:257-260
  unsigned long long link_cap;
  int32_t link_chunk;
  // per utterance [max_frames + 1]: links leaving frame f's tokens
  long long *link_emit_base;     // into frame f+1
  int32_t *link_emit_count;
  long long *link_eps_base;      // into frame f itself
  int32_t *link_eps_count;
  LatticeParams lat;
  SchedParams sched;
};

// Launchers implemented in kd_search.hip, which is compiled twice into the library: the
// large geometry (768 threads, the CU's LDS) and, with -DKD_GEOM_SMALL, the small one
// (256 threads, a third of the LDS, three workgroups to a CU).
// `workgroups` persistent workgroups take the batch's utterances from the work queue.
hipError_t LaunchSearch(const SearchParams &p, int workgroups, hipStream_t stream);
hipError_t LaunchSearchSmall(const SearchParams &p, int workgroups, hipStream_t stream);
size_t SearchLdsBytes();
size_t SearchLdsBytesSmall();
int SearchMaxLdsCols();       // widest row the kernels stage in LDS (wider ones are read in place)
int SearchMaxLdsColsSmall();
int SearchWorkgroupsPerCu();  // resident workgroups a CU holds (build geometry)
int SearchWorkgroupsPerCuSmall();
int SearchT1FrontSmall();     // largest frontier the small geometry keeps in LDS

// Order-preserving map double -> uint64 (so integer atomicMin orders costs).
__host__ __device__ inline unsigned long long CostKey(double d) {
  unsigned long long b;
  __builtin_memcpy(&b, &d, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double KeyCost(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
}
__host__ __device__ inline uint32_t FloatKey(float f) {
  uint32_t b;
  __builtin_memcpy(&b, &f, 4);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float KeyFloat(uint32_t k) {
  uint32_t b = (k >> 31) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}

}  // namespace kd

#endif  // KD_DEVICE_H_
