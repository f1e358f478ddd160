/* include/kd_hip.h -- C ABI of libkd_hip.so: the MI355X (gfx950) token-passing
 * WFST search that replaces kaldi-decoder's CPU hot path.
 *
 * Plain C, opaque handles, caller-owned buffers, int status codes; no C++,
 * torch or pybind11 types cross this boundary.  Every entry point names the
 * reference interface it stands in for ("ref:" paths are relative to
 * /root/reference/kaldi-decoder/).  A binding for the reference's own pybind11
 * layer is shown in INTEGRATION.md.
 *
 * Threading: a kd_graph is immutable after creation and may be shared by any
 * number of batches; a kd_batch is not re-entrant (ref: the decoders are
 * single-threaded objects, csrc/faster-decoder.h:75-76).  All device work of a
 * batch is enqueued on the HIP stream given at creation; calls that return
 * results synchronise that stream.
 *
 * Errors: every function returns KD_OK (0) or a negative KD_E_* code;
 * kd_last_error() returns a thread-local message (ref: KALDI_DECODER_ASSERT/ERR
 * throw std::runtime_error, csrc/log.h:46-51,86-89 -- the host wrapper turns a
 * non-zero status back into that exception).
 */
#ifndef KD_HIP_H_
#define KD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KD_OK 0
#define KD_E_INVALID -1    /* bad argument / failed reference-style assertion */
#define KD_E_HIP -2        /* a HIP runtime call failed (no GPU, OOM, ...)     */
#define KD_E_STATE -3      /* call order violated (AdvanceDecoding before Init) */
#define KD_E_CAPACITY -4   /* a pool / table / frame index is full and cannot grow */

/* bits of kd_batch_status()[i] */
#define KD_UTT_ARENA_FULL 1u
#define KD_UTT_TABLE_FULL 2u
#define KD_UTT_FRAMES_FULL 4u
#define KD_UTT_LINKS_FULL 8u
#define KD_UTT_OUT_FULL 16u
#define KD_UTT_PATH_QUEUE_FULL 32u /* a best path longer than the workgroup's staging queue (2 x table_slots arcs) */

typedef struct kd_graph kd_graph;
typedef struct kd_batch kd_batch;

const char *kd_last_error(void);

/* Number of visible HIP devices (0 when there is no GPU or no driver). */
int kd_device_count(void);

/* ---- decoding graph -------------------------------------------------------
 * Replaces the borrowed `const fst::Fst<fst::StdArc>&` the reference decoders
 * hold (ref: csrc/faster-decoder.h:179, csrc/lattice-simple-decoder.h:298).
 * Input is the FST as CSR in its own arc order: arcs of state s are
 * [row_ptr[s], row_ptr[s+1]); ilabel 0 is epsilon; final_cost[s] = +inf means
 * Weight::Zero() (not final).  The arrays are copied; the device copy is laid
 * out as 16-byte state records and 16-byte arcs with every state's emitting
 * arcs first (stable) -- an arc record also names its destination's arc range,
 * the olabels live in a side array -- and lives until kd_graph_destroy.
 * KD_E_INVALID: ilabels of 2^20 or more (20 bits of the arc record); a graph with an epsilon
 * cycle of total weight zero, or with a negative arc on an epsilon cycle (the reference ends
 * on the first only because an equal cost is no improvement -- and prunes its lattice to
 * another fixpoint than any order-free sweep -- and does not end on the second; cycles of
 * positive weight decode exactly like the reference: csrc/kd_graph_check.h).
 */
int kd_graph_create(int device, int32_t num_states, int32_t start,
                    const int32_t *row_ptr, const int32_t *ilabel,
                    const int32_t *olabel, const float *weight,
                    const int32_t *nextstate, const float *final_cost,
                    kd_graph **out);
void kd_graph_destroy(kd_graph *g);
int32_t kd_graph_num_states(const kd_graph *g);
int64_t kd_graph_num_arcs(const kd_graph *g);
size_t kd_graph_device_bytes(const kd_graph *g);

/* ---- options ----------------------------------------------------------------
 * The five search fields are FasterDecoderOptions (ref: csrc/faster-decoder.h:24-49,
 * same defaults via kd_faster_options_default).  hash_ratio is validated as the
 * reference does (>= 1.0) and otherwise unused: the HashList is replaced by a
 * fixed-capacity open-addressed table.  The remaining fields size device
 * memory; 0 picks a default.
 */
typedef struct kd_faster_options {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float beam_delta;
  float hash_ratio;
  /* capacity knobs (not in the reference).  They are STARTING sizes: whatever a decode
   * exhausts is doubled and the search resumes where it stopped (the reference, too, is
   * limited only by the machine's memory).  0 picks the default. */
  int64_t arena_tokens;  /* token pool shared by the batch's utterances (default
                            max(2^20, 2^17 per utterance); at most 2^31 - 1)     */
  int32_t table_slots;   /* spill table per resident utterance, power of two
                            (default 2^17, at most 2^19)                         */
  int32_t max_frames;    /* frames an utterance can be advanced over (default 2048) */
  int32_t no_grow;       /* 1: never grow; a full pool / table / frame index is
                            KD_E_CAPACITY, naming the knob to raise              */
} kd_faster_options;

void kd_faster_options_default(kd_faster_options *o);

/* ---- a batch of independent FasterDecoder instances --------------------------
 * One kd_batch is `max_batch` FasterDecoder objects over one graph (ref:
 * FasterDecoder::FasterDecoder, csrc/faster-decoder.cc:21-32 -- same option
 * assertions).  Utterances never interact; the batch only exists so one kernel
 * launch advances all of them.  `stream` is a hipStream_t (NULL = default).
 */
int kd_batch_create(const kd_graph *g, const kd_faster_options *opts,
                    int32_t max_batch, void *stream, kd_batch **out);
void kd_batch_destroy(kd_batch *b);

/* FasterDecoder::SetOptions (ref: csrc/faster-decoder.h:78): search fields only. */
int kd_batch_set_options(kd_batch *b, const kd_faster_options *opts);

/* FasterDecoder::InitDecoding for utterances [0, batch) (ref:
 * csrc/faster-decoder.cc:42-56): start token, epsilon closure with cutoff
 * FLT_MAX, num_frames_decoded = 0. */
int kd_batch_init_decoding(kd_batch *b, int32_t batch);

/* Memory kind of the log-likelihood pointers handed to kd_batch_advance. */
#define KD_MEM_HOST 0
#define KD_MEM_DEVICE 1

/* FasterDecoder::AdvanceDecoding over a DecodableCtc per utterance (ref:
 * csrc/faster-decoder.cc:126-152, csrc/decodable-ctc.cc:22-38).
 *   loglikes[i]   row-major [num_rows[i], num_cols] float log-probs, row stride
 *                 `row_stride` floats; LogLikelihood(frame, index) =
 *                 loglikes[i][(frame - offset[i]) * row_stride + index - 1]
 *   offset[i]     DecodableCtc's `offset` (frame index of row 0); may be NULL (0)
 *   max_num_frames as in the reference (-1: all frames ready)
 * NumFramesReady = offset + num_rows must be >= frames already decoded.
 * Host pointers are staged to the device first; device pointers are read in
 * place and must stay valid until the call returns.
 */
int kd_batch_advance(kd_batch *b, const float *const *loglikes,
                     const int32_t *num_rows, const int32_t *offset,
                     int32_t num_cols, int64_t row_stride, int mem_kind,
                     int32_t max_num_frames);

/* As kd_batch_advance but only enqueues the search kernel on the batch's
 * stream (device pointers only, no synchronisation, no status read-back);
 * pair with kd_batch_sync.  This is the call timed by bench.py. */
int kd_batch_advance_async(kd_batch *b, const float *const *loglikes_dev,
                           const int32_t *num_rows, const int32_t *offset,
                           int32_t num_cols, int64_t row_stride,
                           int32_t max_num_frames);
int kd_batch_sync(kd_batch *b);

/* FasterDecoder::Decode / LatticeSimpleDecoder::Decode (ref: csrc/faster-decoder.cc:121-124,
 * csrc/lattice-simple-decoder.cc:53-73) for utterances [0, batch) as ONE kernel launch:
 * InitDecoding, AdvanceDecoding over every row and -- lattice batches, as `want` asks --
 * FinalizeDecoding (the reference's Decode() always finalizes, :68) and GetRawLattice
 * (use_final_probs = true) into the batch's output pool -- 1-best batches: ReachedFinal
 * and GetBestPath(use_final_probs = true) -- each by the workgroup that decoded the
 * utterance.  Device pointers only; enqueues and returns, pair with
 * kd_batch_sync.  Afterwards kd_batch_raw_lattice_prepare / _copy (or the best-path
 * calls) only move results; no further kernel runs.  The batch may hold any number of
 * utterances: one launch keeps one workgroup per CU busy from a work queue. */
#define KD_DECODE_FINALIZE 1   /* lattice batches: FinalizeDecoding                        */
#define KD_DECODE_LATTICE 2    /* lattice batches: GetRawLattice(use_final_probs = true)  */
#define KD_DECODE_BEST_PATH 4  /* 1-best batches: ReachedFinal + GetBestPath(true)         */
int kd_batch_decode_async(kd_batch *b, int32_t batch, const float *const *loglikes_dev,
                          const int32_t *num_rows, const int32_t *offset, int32_t num_cols,
                          int64_t row_stride, int32_t want);
/* The same, waiting for the result; host matrices (KD_MEM_HOST) are staged first. */
int kd_batch_decode(kd_batch *b, int32_t batch, const float *const *loglikes,
                    const int32_t *num_rows, const int32_t *offset, int32_t num_cols,
                    int64_t row_stride, int mem_kind, int32_t want);
/* The same for HOST matrices without waiting for the kernel (ref: DecodableCtc(feats),
 * csrc/decodable-ctc.h:15-16, copies its matrix; here the rows go to the device): returns
 * when the rows have left the caller's memory and the launch is enqueued behind them on the
 * batch's stream; pair with kd_batch_sync.  A caller that keeps two batches going this way
 * has one batch's copies overlap the other's search. */
int kd_batch_decode_host_async(kd_batch *b, int32_t batch, const float *const *loglikes,
                               const int32_t *num_rows, const int32_t *offset, int32_t num_cols,
                               int64_t row_stride, int32_t want);

/* FasterDecoder::NumFramesDecoded per utterance (ref: csrc/faster-decoder.h:104). */
int kd_batch_num_frames_decoded(kd_batch *b, int32_t *out);

/* Per-utterance KD_UTT_* bits; non-zero means that utterance stopped early. */
int kd_batch_status(kd_batch *b, uint32_t *out);

/* FasterDecoder::ReachedFinal per utterance (ref: csrc/faster-decoder.cc:347-354). */
int kd_batch_reached_final(kd_batch *b, int32_t *out);

/* FasterDecoder::GetBestPath (ref: csrc/faster-decoder.cc:356-424), two steps:
 * kd_batch_best_path_prepare runs the traceback on the device and returns the
 * number of arcs of each utterance's path (-1: no token survived, GetBestPath
 * would return false); kd_batch_best_path_copy writes utterance i's arcs at
 * [path_offset[i], path_offset[i] + len[i]) of the caller's arrays, in path
 * order: the linear LatticeArc FST before RemoveEpsLocal (that function lives
 * in kaldifst, not in the reference tree).  final_weight[2*i..] is the
 * LatticeWeight set on the last state: (Final(state), 0) when a final state was
 * reached and use_final_probs, else One() = (0, 0).
 */
int kd_batch_best_path_prepare(kd_batch *b, int32_t use_final_probs,
                               int32_t *path_len);
int kd_batch_best_path_copy(kd_batch *b, const int64_t *path_offset,
                            int32_t *ilabel, int32_t *olabel, int32_t *state,
                            float *graph_cost, float *ac_cost,
                            float *final_weight);

/* Frontier of one utterance after `frame_plus_one` frames (0 = after
 * InitDecoding): the tokens the reference keeps in toks_ with cost below the
 * frame's cutoff, sorted by state.  Pass cap = 0 to query the count. */
int kd_batch_frame_tokens(kd_batch *b, int32_t utt, int32_t frame_plus_one,
                          int32_t cap, int32_t *count, int32_t *state,
                          double *cost);

/* Search counters summed over frames, per utterance: KD_NUM_STATS int64 each. */
enum {
  KD_STAT_FRAMES = 0,
  KD_STAT_TOKENS_IN,        /* tokens entering ProcessEmitting                 */
  KD_STAT_TOKENS_EXPANDED,  /* of those, cost < weight_cutoff                  */
  KD_STAT_ARCS_SCANNED,     /* emitting arcs of expanded tokens (the arcs the reference's loop visits,
                               faster-decoder.cc:204-211; equal to the oracle's count)        */
  KD_STAT_ARCS_KEPT,        /* of those, below the running cutoff              */
  KD_STAT_TOKENS_OUT,       /* tokens written (after closure)                  */
  KD_STAT_EPS_ARCS_SCANNED, /* epsilon arcs read in ProcessNonemitting         */
  KD_STAT_EPS_EXPANSIONS,   /* token expansions in ProcessNonemitting          */
  KD_STAT_MAX_TOKENS,       /* max tokens written in one frame                 */
  KD_STAT_SPILL_FRAMES,     /* frames that fell back to the global table       */
  KD_STAT_LINKS,            /* forward links written (lattice decoders)        */
  KD_STAT_SMALL_FRAMES,     /* frames run entirely out of LDS (frontier <= 1024) */
  KD_STAT_ARCS_SKIPPED,     /* of ARCS_SCANNED, arcs never fetched: a lower bound of their cost was
                               already at or beyond the running cutoff (label index, kd_search.hip);
                               arcs READ = ARCS_SCANNED - ARCS_SKIPPED                          */
  KD_STAT_INDEX_PROBES,     /* 4-byte label-index entries read to find the arcs that were fetched */
  KD_STAT_HANDOVERS,        /* times the utterance changed hands between the two kinds of workgroups */
  KD_STAT_SMALL_WG_FRAMES,  /* frames run by a workgroup of the small kind (three to a CU) */
  KD_STAT_PRUNES,           /* PruneActiveTokens sweeps run (lattice decoders; owed ones are not counted until run) */
  KD_NUM_STATS
};
int kd_batch_stats(kd_batch *b, int64_t *out /* [batch][KD_NUM_STATS] */);

/* ---- SimpleDecoder ------------------------------------------------------------
 * A kd_batch created with kd_batch_create_simple has SimpleDecoder's semantics
 * (ref: csrc/simple-decoder.{h,cc}): plain beam on the emitting step and in the
 * epsilon closure, tokens at or beyond best + beam dropped after the closure
 * (PruneToks, ref: csrc/simple-decoder.cc:252-281), token cost = prev + (weight +
 * acoustic) with the float sum first (ref: csrc/simple-decoder.h:99), best-path
 * arcs carrying the arc's own (graph, acoustic) pair.  All kd_batch_* calls
 * apply; kd_batch_final_relative_cost gives FinalRelativeCost (ref: :78-100).
 */
int kd_batch_create_simple(const kd_graph *g, float beam, int64_t arena_tokens,
                           int32_t table_slots, int32_t max_frames, int32_t max_batch,
                           void *stream, kd_batch **out);

/* ---- lattice decoders --------------------------------------------------------
 * A kd_batch created with kd_batch_create_lattice runs the lattice-generating
 * search: LatticeSimpleDecoder's semantics (ref: csrc/lattice-simple-decoder.cc)
 * and, with max_active / min_active set, the LatticeFasterDecoder the reference
 * only declares (ref: csrc/lattice-faster-decoder.h:23-134).  Costs are float,
 * every arc below the frame's cutoff is recorded as a forward link.  Every
 * prune_interval frames the search runs PruneActiveTokens (ref:
 * csrc/lattice-simple-decoder.cc:57-59,198-223; tolerance lattice_beam *
 * prune_scale) and moves what survives of the older frames to a compacted
 * archive, so an utterance's memory follows its pruned lattice plus a window of
 * recent frames; kd_batch_finalize_decoding then prunes exactly (delta 0), so the
 * finalized lattice does not depend on prune_interval.  init / advance /
 * num_frames_decoded / status / stats are the kd_batch_* calls above.
 * kd_batch_reached_final on a lattice batch is FinalRelativeCost() != infinity
 * (ref: csrc/lattice-simple-decoder.h:142-147); kd_batch_best_path_* are
 * KD_E_STATE (lattice tokens carry no back-pointers: take the shortest path
 * through the raw lattice, ref: csrc/lattice-simple-decoder.cc:574-580).
 */
typedef struct kd_lattice_options {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float lattice_beam;
  int32_t prune_interval;
  float beam_delta;
  float hash_ratio;
  float prune_scale;
  /* capacity knobs as in kd_faster_options: starting sizes, grown on demand */
  int64_t arena_tokens;
  int64_t arena_links;   /* forward-link pool (default two per token record)    */
  int32_t table_slots;
  int32_t max_frames;
  int32_t no_grow;
} kd_lattice_options;

/* LatticeFasterDecoderConfig defaults (ref: csrc/lattice-faster-decoder.h:48-66). */
void kd_lattice_options_default(kd_lattice_options *o);

/* Same assertions as LatticeFasterDecoderConfig::Check (ref: :128-133). */
int kd_batch_create_lattice(const kd_graph *g, const kd_lattice_options *opts,
                            int32_t max_batch, void *stream, kd_batch **out);

/* FinalizeDecoding (ref: csrc/lattice-simple-decoder.cc:407-420): final-cost
 * pruning of the last frame, then an exact (delta = 0) backward prune of every
 * frame.  Afterwards the batch cannot be advanced. */
int kd_batch_finalize_decoding(kd_batch *b);

/* FinalRelativeCost per utterance (ref: :562-572). */
int kd_batch_final_relative_cost(kd_batch *b, float *out);

/* GetRawLattice (ref: :584-657) in two steps.  prepare returns, per utterance,
 * sizes[4*i..] = {states, arcs, finals, start state}; all zero / start -1 is the
 * reference's `false` (some frame has no token).  copy writes utterance i's
 * arcs at arc_off[i], its states' (frame, graph state) labels at state_off[i]
 * and its final weights at final_off[i].  States are numbered frame-major and,
 * inside a frame, by graph state; arcs are ordered by (source state, graph arc),
 * finals by state: the same input gives the same arrays from run to run, as the
 * reference's GetRawLattice does (its own order is its containers' iteration
 * order, :623-644).  use_final_probs = 0 after kd_batch_finalize_decoding is an
 * error, as in the reference (:594-597). */
int kd_batch_raw_lattice_prepare(kd_batch *b, int32_t use_final_probs, int32_t *sizes);
int kd_batch_raw_lattice_copy(kd_batch *b, const int64_t *state_off, const int64_t *arc_off,
                              const int64_t *final_off, int32_t *src, int32_t *dst,
                              int32_t *ilabel, int32_t *olabel, float *graph_cost,
                              float *ac_cost, int32_t *state_frame, int32_t *state_graph,
                              int32_t *final_state, float *final_graph);

/* ---- logits in, log-softmax fused -------------------------------------------------------
 * The reference's DecodableCtc takes log-probabilities (ref: csrc/decodable-ctc.cc:28); its
 * callers apply log_softmax to the network output first.  With this switch on, the matrices
 * handed to advance / decode hold LOGITS and every row is normalised (x - logsumexp(x),
 * float) where the kernel stages it in LDS -- no extra pass over [B, T, V] in HBM.  Rows of
 * more than 2048 columns are not staged: KD_E_INVALID.  Costs then agree with a decode of
 * torch.log_softmax(logits) to float rounding (the tests state 1e-4 relative), not bit for
 * bit; with the switch off nothing changes. */
int kd_batch_set_log_softmax(kd_batch *b, int32_t on);

/* ---- exported results in DEVICE memory ---------------------------------------------
 * The kernel writes exported lattices into a pinned host pool of the batch.  A caller that
 * wants them on the GPU as well -- the multi-GPU gather hands RCCL a device buffer: no host
 * packing, no upload -- lends a device buffer of `cap_words` 8-byte words: the export that
 * is fused into a decode (KD_DECODE_FINALIZE | KD_DECODE_LATTICE) then writes every word
 * there too, at the same offsets; the host-array calls above keep reading the pinned pool.
 * NULL ends the loan.  The lent buffer does not grow: a call whose lattices do not fit is
 * KD_E_CAPACITY.  (No counterpart in the reference: its lattices are host objects,
 * csrc/lattice-simple-decoder.cc:584-657.) */
int kd_batch_set_output_buffer(kd_batch *b, void *device_words, int64_t cap_words);
/* After a decode with KD_DECODE_LATTICE (or kd_batch_raw_lattice_prepare): how many words of
 * the output pool / buffer are in use, and per utterance {first word, states, arcs, finals,
 * start state}.  An utterance's region: states {frame, graph state} x ns | finals {state,
 * weight} x nf | arcs {src, dst | ilabel, olabel | graph, ac} x na, ordered as
 * kd_batch_raw_lattice_copy delivers them.  With a lent buffer the regions are complete in it
 * whichever export ran: the one fused into a decode writes both copies; an export that the
 * host has to put in order first (an un-finalized lattice, kd_batch_raw_lattice_prepare apart
 * from the decode, a lattice that outgrew the staging lists) is ordered in the pinned pool
 * and copied into the buffer by this call -- once per export, on the batch's stream, which the
 * call waits for: work of the caller's that still READS the lent buffer (the previous call's
 * gather payload, say) must have completed before this call is made. */
int kd_batch_raw_lattice_table(kd_batch *b, int64_t *used_words, int64_t *table /* [batch][5] */);


/* Diagnostic builds only (make -C kaldi-decoder_amd prof, -DKD_PROFILE): shader
 * clock cycles lane 0 spent per phase, 24 int64 per utterance (bucket names in
 * tools/prof_phases.py).  The shipping library executes no stamps and returns
 * KD_E_STATE. */
int kd_batch_phase_cycles(kd_batch *b, int64_t *out /* [batch][24] */);

/* Device time of the most recent search-kernel launch, from HIP events on the
 * batch's stream (valid after a sync). */
int kd_batch_last_kernel_ms(kd_batch *b, float *ms);

size_t kd_batch_device_bytes(const kd_batch *b);

/* Current capacities and use: out[0..7] = token pool records, link pool links, spill
 * table slots, max_frames, token records in use, links in use, number of times anything
 * was grown since creation, persistent workgroups per launch. */
int kd_batch_capacity(const kd_batch *b, int64_t *out);

/* PruneActiveTokens, owed or run (lattice decoders; round 6).  The reference prunes every
 * prune_interval frames (lattice-simple-decoder.cc:57-59), to bound memory; what FinalizeDecoding
 * leaves does not depend on it.  With lazy != 0 (the default) a prune is only OWED when its frame
 * comes and is run later -- the owed ones in order, each exactly as it would have run then: a prune
 * at frame m touches nothing above m -- when something can tell the difference: GetRawLattice of
 * the un-finalized lattice (kd_batch_raw_lattice_prepare before kd_batch_finalize_decoding), or
 * memory: a pool that is not to grow any further -- `no_grow`, or it has reached the batch's budget
 * (2^29 token / 2^30 link records: 13 + 26 GB) -- is garbage-collected instead: once three
 * quarters of it have been handed out prunes are paid when their frames come, and a frame that
 * finds no room runs what it owes and starts again (the collector behind a prune returns the
 * utterance's old chunks).  A pool that may still grow is grown: memory for time.  lazy = 0: every
 * prune runs when its frame comes, as in the reference.  Results are the same either way; memory
 * follows the utterance's length instead of its pruned lattice until a pool reaches its budget. */
int kd_batch_set_lazy_prune(kd_batch *b, int32_t lazy);

/* How a call's utterances are dealt out on the chip (round 6).  A call of at least
 * `min_batch` utterances is run by TWO kernels side by side on disjoint sets of CUs: "large"
 * workgroups (one per CU, the CU's whole LDS) and "small" ones (three per CU); an utterance is
 * handed from one kind to the other at a frame boundary when the frontier its next frame starts
 * from crosses a threshold -- to a small workgroup at `mig_lo` tokens or fewer, to a large one
 * above `mig_hi` (0 <= mig_lo <= mig_hi) -- and, with `prune_small`, PruneActiveTokens runs on the
 * small kind.  Results do not depend on any of this (the reference runs every frame of an
 * utterance in one thread: lattice-simple-decoder.cc:53-66, faster-decoder.cc:126-152).
 * kd_batch_split_info: out[0..7] = split available (0: one kernel, as before round 6; the other
 * values are then 0), CUs of the large kind, large workgroups, small workgroups, mig_lo, mig_hi,
 * prune_small | stay << 1, min_batch.  kd_batch_set_split: a negative argument leaves that setting
 * as it is; enable = 0 makes every later call of this batch a one-kernel call.  `stay`: a
 * workgroup that finds nothing to do normally leaves the call once its XCD has more workgroups
 * than unfinished utterances (its CUs go to the next call's kernels); with stay = 1 every
 * workgroup stays until the call's last utterance is done -- for tests: a call of a few
 * utterances then keeps workgroups of both kinds to change hands between. */
int kd_batch_split_info(const kd_batch *b, int32_t *out);
/* Diagnostic: where the two kinds of workgroups spent their time, summed over this batch's split
 * launches so far (shader-clock cycles of one lane per workgroup).  out[0..15], each a pair
 * {small kind, large kind}: cycles looking for an utterance, taking one up (until its first
 * frame starts), running frames / pruning / tail, in the queue operation that found it;
 * utterances taken up fresh, resumed from a queue, taken from the other kind's queue; workgroups
 * that ran. */
int kd_batch_sched_profile(kd_batch *b, int64_t *out);
int kd_batch_set_split(kd_batch *b, int32_t enable, int32_t min_batch, int32_t mig_lo, int32_t mig_hi,
                       int32_t prune_small, int32_t stay);

#ifdef __cplusplus
}
#endif
#endif /* KD_HIP_H_ */
