/*
 * include/fltx.h -- C ABI of the MI355X-native batched beam-search decoder.
 *
 * This is the drop-in boundary for the hot path of flashlight/text's
 * LexiconFreeDecoder / LexiconDecoder.  The reference
 * has no FFI for this path (it is a header/C++ library); the functions below
 * are what a cgo/JNI/ctypes/C++ binding of that path binds.  Each entry point
 * cites the reference interface it replaces (paths relative to
 * /root/reference/flashlight/lib/text/).
 *
 * Conventions: plain pointers and sizes only; every function returns an int
 * status (FLTX_OK == 0) and records a message retrievable with
 * fltx_last_error() (thread-local).  The reference reports errors as C++
 * exceptions (decoder/lm/LM.h:40, decoder/Trie.cpp:32,54); the C++ facade in
 * text_amd/csrc/flashlight/ converts non-zero statuses back into the same
 * exception types.
 *
 * There is NO CPU fallback behind this ABI: every decode call runs the HIP
 * kernels in text_amd/csrc/fltx_kernels.h on a gfx950 device and fails with
 * FLTX_ERR_HIP when no device is usable.
 */
#ifndef FLTX_H_
#define FLTX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLTX_API __attribute__((visibility("default")))

enum {
  FLTX_OK = 0,
  FLTX_ERR_INVALID = 1,     /* bad argument (std::invalid_argument) */
  FLTX_ERR_HIP = 2,         /* HIP runtime failure / no device (std::runtime_error) */
  FLTX_ERR_OOM = 3,         /* device allocation failed */
  FLTX_ERR_UNSUPPORTED = 4, /* configuration the device path does not cover */
  FLTX_ERR_RANGE = 5,       /* index out of range (std::out_of_range) */
  FLTX_ERR_STATE = 6,       /* call sequence error (e.g. results before decode) */
  FLTX_ERR_CALLBACK = 7     /* a host-LM callback reported failure (the binding rethrows what the user's LM threw) */
};

/* CriterionType, decoder/Decoder.h:16 (S2S is out of scope). */
enum { FLTX_CRITERION_ASG = 0, FLTX_CRITERION_CTC = 1 };
/* SmearingMode, decoder/Trie.h:21-25. */
enum { FLTX_SMEAR_NONE = 0, FLTX_SMEAR_MAX = 1, FLTX_SMEAR_LOGADD = 2 };
enum { FLTX_DECODER_LEXFREE = 0, FLTX_DECODER_LEXICON = 1 };

/* LexiconDecoderOptions (decoder/LexiconDecoder.h:21-31); the lexicon-free
 * decoder (decoder/LexiconFreeDecoder.h:20-28) ignores word_score/unk_score. */
typedef struct fltx_options {
  int32_t beam_size;
  int32_t beam_size_token;
  double beam_threshold;
  double lm_weight;
  double word_score;
  double unk_score;
  double sil_score;
  int32_t log_add;
  int32_t criterion;
} fltx_options;

typedef struct fltx_ctx fltx_ctx;         /* one HIP device + stream */
typedef struct fltx_lm fltx_lm;           /* LM tables resident in HBM */
typedef struct fltx_trie fltx_trie;       /* flattened lexicon trie in HBM */
typedef struct fltx_decoder fltx_decoder; /* options + per-batch workspace */

FLTX_API const char* fltx_last_error(void);
FLTX_API const char* fltx_version(void);

/* ---- context ------------------------------------------------------------ */
/* device < 0: use the current HIP device.  stream == NULL: library-owned
 * stream.  A caller-owned hipStream_t may be passed as void*. */
FLTX_API int fltx_ctx_create(int device, void* stream, fltx_ctx** out);
FLTX_API int fltx_ctx_destroy(fltx_ctx* ctx);
FLTX_API int fltx_ctx_synchronize(fltx_ctx* ctx);
/* the hipStream_t kernels are launched on (for event timing by the caller) */
FLTX_API void* fltx_ctx_stream(fltx_ctx* ctx);
/* a number that names this context and is never handed out again (an address can be: callers that cache per-context
 * objects -- the facade's Trie keeps one flattened copy per context -- key them by this, not by the pointer) */
FLTX_API uint64_t fltx_ctx_uid(fltx_ctx* ctx);

/* ---- language models ---------------------------------------------------- */
/* ZeroLM (decoder/lm/ZeroLM.h:22-32, ZeroLM.cpp:14-26). */
/* (ctx may be NULL for both LM constructors: tables are built on the host and
 * uploaded when a decoder is created.) */
FLTX_API int fltx_lm_zero_create(fltx_ctx* ctx, fltx_lm** out);
/* Back-off n-gram LM with ARPA semantics, replacing the KenLM adapter
 * (decoder/lm/KenLM.h:52-63, KenLM.cpp:32-83).  The model is passed as flat
 * host arrays, one row per n-gram, all orders concatenated in increasing
 * order: ngram_order[i] in 1..order, ngram_words[i*order .. i*order+order-1]
 * = LM word ids oldest first (unused tail = -1), prob/backoff = log10 values.
 * usr_to_lm[u] maps the decoder's word/token index u to an LM word id
 * (KenLM.cpp:44-49); bos/eos/unk are LM ids of <s>, </s>, <unk>. */
FLTX_API int fltx_lm_ngram_create(fltx_ctx* ctx, int32_t order, int64_t n_ngrams,
                                  const int32_t* ngram_order,
                                  const int32_t* ngram_words,
                                  const float* prob, const float* backoff,
                                  const int32_t* usr_to_lm, int32_t n_usr,
                                  int32_t bos, int32_t eos, int32_t unk,
                                  fltx_lm** out);
/* KenLM(path, usrTknDict) for ARPA text models (decoder/lm/KenLM.cpp:32-50):
 * parse the file on the host, map every '\n'-separated entry of usr_words
 * (index = user dictionary index) to an LM word id, unknown strings to <unk>.
 * No device is needed until a decoder is created with the LM. */
FLTX_API int fltx_lm_arpa_load(const char* path, const char* usr_words, fltx_lm** out);
/* A user-defined LM: any subclass of `LM` (decoder/lm/LM.h:61-85), e.g. through the reference's Python trampoline
 * PyLM (bindings/python/flashlight/lib/text/_decoder.cpp:39-56).  It has no tables to flatten, so LM::start / score /
 * finish stay calls into host code -- but the beam search does not: candidates, merge, prune and history run in the
 * HIP kernels as for every other LM.  Once per frame a kernel lists the (LM state, index) pairs the frame's candidate
 * generation will ask about, for the whole batch; the library asks the callbacks below about each DISTINCT pair
 * once (what LMState::child's memo gives the reference, lm/LM.h:24-34: LM::score must be a function of its arguments)
 * and uploads the answers; the frame's kernel reads them.  The shape is the reference's own for LMs that batch
 * their queries (decoder/lm/ConvLM.cpp:144-240 behind Utils.h:346-354 updateLMCache).
 *
 * LM states cross this boundary as int32 ids, per utterance, handed out by the callee: 0 is what LM::start returned;
 * `score` must return the SAME id whenever the user's LM returns the same LMState object and a new id otherwise --
 * the reference merges hypotheses on the state's address (lm/LM.h:37-49), the kernels merge on this id.
 * Every callback returns 0, or non-zero to abort the decode call with FLTX_ERR_CALLBACK.  Callbacks are made on the
 * thread that called the decoder, never concurrently (decoder/Utils.h:60-62).  Not usable with fltx_group_*. */
typedef struct fltx_host_lm {
  void* user;
  /* decodeBegin: LM::start(false) (LexiconFreeDecoder.cpp:24, LexiconDecoder.cpp:24) for utterances 0 .. n_utt-1;
   * ids of an earlier decode on this decoder are void */
  int32_t (*start)(void* user, int32_t n_utt);
  /* n questions: idx[i] >= 0: LM::score(state[i] of utterance utt[i], idx[i]); idx[i] == -1: LM::finish(state[i]).
   * out_state[i] = id of the returned state, out_score[i] = the returned score */
  int32_t (*score)(void* user, int32_t n, const int32_t* utt, const int32_t* state, const int32_t* idx,
                   int32_t* out_state, float* out_score);
  /* may be NULL.  LM::updateCache(states of utterance utt's beam) after every frame (Utils.h:346-354) */
  int32_t (*update_cache)(void* user, int32_t utt, int32_t n, const int32_t* states);
  /* may be NULL.  After Decoder::prune: only these states of utterance utt are still held by a hypothesis; ids of
   * the others will not be passed again and their LMState objects may be released (what dropping the pruned
   * hypotheses' shared_ptrs does in the reference, Utils.h:312-342) */
  int32_t (*retain)(void* user, int32_t utt, int32_t n, const int32_t* states);
} fltx_host_lm;
FLTX_API int fltx_lm_host_create(const fltx_host_lm* callbacks, fltx_lm** out);
FLTX_API int fltx_lm_destroy(fltx_lm* lm);
/* LM::start + LM::score chain + optional LM::finish on the device tables
 * (decoder/lm/LM.h:61-78); per_word may be NULL.  Used by known-answer tests
 * (test/decoder/DecoderTest.cpp:107-120). */
FLTX_API int fltx_lm_score_sequence(fltx_lm* lm, const int32_t* usr_words,
                                    int32_t n, int32_t with_finish,
                                    float* per_word, float* total);

/* Explicit-state LM::start / LM::score / LM::finish on the host copy of the
 * flat tables (decoder/lm/LM.h:61-78), for the C++ facade's LM objects (trie
 * label scores, known-answer checks).  A state is fltx_lm_state_size() int32
 * context node ids (0 for ZeroLM).  usr_idx == -1 scores </s> (finish). */
FLTX_API int fltx_lm_state_size(fltx_lm* lm, int32_t* n);
FLTX_API int fltx_lm_start(fltx_lm* lm, int32_t start_with_nothing, int32_t* ctx_out);
FLTX_API int fltx_lm_step(fltx_lm* lm, const int32_t* ctx_in, int32_t usr_idx, int32_t* ctx_out,
                          float* score);

/* ---- lexicon trie -------------------------------------------------------- */
/* Upload an already built and smeared trie (decoder/Trie.h:64-92) as flat
 * arrays; node 0 is the root.  child[node*n_tokens + token] = child node or
 * -1; max_score[node] (float, after smearing); labels of node i are
 * labels[label_off[i] .. label_off[i+1]) (at most kTrieMaxLabel = 6 each,
 * Trie.h:19). */
FLTX_API int fltx_trie_create(fltx_ctx* ctx, int64_t n_nodes, int32_t n_tokens,
                              const int32_t* child, const float* max_score,
                              const int32_t* label_off, const int32_t* labels,
                              fltx_trie** out);
FLTX_API int fltx_trie_destroy(fltx_trie* trie);

/* Host-side trie builder with the reference's Trie semantics
 * (decoder/Trie.h:64-92, Trie.cpp:26-101): insert / search / smear on the CPU
 * (setup, runs once), then fltx_htrie_upload flattens it into HBM.  insert
 * returns FLTX_ERR_RANGE for an index outside [0, max_children)
 * (Trie.cpp:31-34 throws std::out_of_range) and silently drops labels beyond
 * kTrieMaxLabel = 6 per node (Trie.cpp:40-46). */
typedef struct fltx_htrie fltx_htrie;
FLTX_API int fltx_htrie_create(int32_t max_children, int32_t root_idx, fltx_htrie** out);
FLTX_API int fltx_htrie_destroy(fltx_htrie* t);
FLTX_API int fltx_htrie_insert(fltx_htrie* t, const int32_t* indices, int32_t n,
                               int32_t label, float score);
/* *found = 0/1; max_score, n_labels, labels[<=6], scores[<=6] may be NULL */
FLTX_API int fltx_htrie_search(fltx_htrie* t, const int32_t* indices, int32_t n,
                               int32_t* found, float* max_score, int32_t* n_labels,
                               int32_t* labels, float* scores);
FLTX_API int fltx_htrie_smear(fltx_htrie* t, int32_t mode);
FLTX_API int fltx_htrie_num_nodes(fltx_htrie* t, int64_t* n);
FLTX_API int fltx_htrie_upload(fltx_htrie* t, fltx_ctx* ctx, fltx_trie** out);
/* One node of the host trie by id (0 = root): TrieNode::idx / maxScore / labels / scores and
 * the (token, node id) pairs of TrieNode::children (decoder/Trie.h:30-55), for bindings that
 * expose the node tree (bindings/python/.../_decoder.cpp:173-186).  Any output may be NULL;
 * labels / scores hold up to 6 entries, the child arrays child_capacity. */
FLTX_API int fltx_htrie_node(fltx_htrie* t, int64_t node, int32_t* token, float* max_score,
                             int32_t* n_labels, int32_t* labels, float* scores, int32_t* n_children,
                             int32_t* child_tokens, int64_t* child_nodes, int32_t child_capacity);

/* ---- decoder ------------------------------------------------------------- */
/* LexiconFreeDecoder(opt, lm, sil, blank, transitions)
 * (decoder/LexiconFreeDecoder.h:102-112) when kind == FLTX_DECODER_LEXFREE
 * (trie NULL, unk/is_lm_token ignored);
 * LexiconDecoder(opt, trie, lm, sil, blank, unk, transitions, isLmToken)
 * (decoder/LexiconDecoder.h:117-133) when kind == FLTX_DECODER_LEXICON.
 * transitions: n_tokens*n_tokens floats or NULL/0 (copied). */
FLTX_API int fltx_decoder_create(fltx_ctx* ctx, int32_t kind,
                                 const fltx_options* opt, const fltx_trie* trie,
                                 const fltx_lm* lm, int32_t sil, int32_t blank,
                                 int32_t unk, const float* transitions,
                                 int32_t n_transitions, int32_t is_lm_token,
                                 fltx_decoder** out);
FLTX_API int fltx_decoder_destroy(fltx_decoder* dec);

/* Batched Decoder::decode (decoder/Decoder.h:51-57) for B independent
 * utterances: decodeBegin + decodeStep(all frames) + decodeEnd + back-trace,
 * all on the device.  Utterance b reads T[b]*N floats (frame-major, token
 * fastest, LexiconDecoder.cpp:69) starting at emissions + offsets[b].
 * emissions_on_device != 0: `emissions` is a device pointer (HBM resident);
 * otherwise it is a host pointer and is copied to the device first (the
 * pointer is only borrowed for the duration of the call).  offsets and T are
 * host arrays.  The call is asynchronous on the context stream; results are
 * read with fltx_result_*, which synchronise. */
FLTX_API int fltx_decode_batch(fltx_decoder* dec, const float* emissions,
                               int32_t emissions_on_device,
                               const int64_t* offsets, const int32_t* T,
                               int32_t B, int32_t N);

/* Streaming interface for B parallel streams (Decoder::decodeBegin /
 * decodeStep / decodeEnd / prune, decoder/Decoder.h:42-61).  max_frames bounds
 * the total frames buffered per stream between prunes.
 * Lifetime of a device buffer (emissions_on_device != 0; the same holds for fltx_decode_batch): every read of it
 * is queued on the context's stream before the call returns -- also the second pass of a lexicon stream's chunk
 * whose candidate list overflowed, which is therefore not deferred to a later call for such a chunk.  The caller
 * may overwrite the buffer in stream order on that stream, or from anywhere after fltx_ctx_synchronize / after
 * any call that returns results of this chunk.  A host buffer is only borrowed for the duration of the call. */
FLTX_API int fltx_stream_begin(fltx_decoder* dec, int32_t B, int32_t N,
                               int32_t max_frames);
FLTX_API int fltx_stream_step(fltx_decoder* dec, const float* emissions,
                              int32_t emissions_on_device,
                              const int64_t* offsets, const int32_t* T);
FLTX_API int fltx_stream_end(fltx_decoder* dec);
/* Decoder::prune(lookBack) for every stream (LexiconFreeDecoder.cpp:205-227). */
FLTX_API int fltx_stream_prune(fltx_decoder* dec, int32_t look_back);
/* nDecodedFramesInBuffer (LexiconFreeDecoder.cpp:201-203). */
FLTX_API int fltx_stream_frames_in_buffer(fltx_decoder* dec, int32_t b,
                                          int32_t* n);

/* ---- results (getAllFinalHypothesis / getBestHypothesis) ------------------ */
/* Number of hypotheses of utterance b and the length (finalFrame + 1) of each
 * tokens/words vector (decoder/Utils.h:236-247). */
FLTX_API int fltx_result_count(fltx_decoder* dec, int32_t b, int32_t* n_hyp,
                               int32_t* length);
/* Copy out the first max_hyp hypotheses of utterance b, best first:
 * scores[3*i + {0,1,2}] = score, emittingModelScore, lmScore
 * (decoder/Utils.h:30-39); tokens/words [i*length + f]; either may be NULL. */
FLTX_API int fltx_result_fetch(fltx_decoder* dec, int32_t b, int32_t max_hyp,
                               double* scores, int32_t* tokens, int32_t* words,
                               int32_t* n_copied);
/* The whole batch's n-best in one PCIe transfer per array (offline decodes):
 * counts, scores and token / word rows are copied into pinned host buffers
 * owned by the decoder and pointers to them are returned; they stay valid
 * until the next decode on this decoder.  Hypothesis k of utterance b:
 * scores[(b * beam_size + k) * 3 + {0,1,2}] = score, emitting-model score, LM
 * score; tokens + offsets[b] + k * length[b] holds its length[b] tokens (words
 * likewise; *words is NULL for the lexicon-free decoder, whose word sequence
 * is all -1, LexiconFreeDecoder.h:80-82).  Replaces a loop of
 * getAllFinalHypothesis() calls (Decoder.h:71-73) over the utterances. */
FLTX_API int fltx_result_fetch_batch(fltx_decoder* dec, const int32_t** n_hyp, const int32_t** length,
                                     const double** scores, const int32_t** tokens, const int32_t** words,
                                     const int64_t** offsets);
/* The same, compacted on the device first so that only what exists crosses PCIe: the rows of the n_hyp[b]
 * hypotheses an utterance really has (a lexicon beam of 100 returns a dozen), tokens as bytes (0xFF = -1;
 * token sets of up to 254 symbols -- FLTX_ERR_UNSUPPORTED beyond: use fltx_result_fetch_batch), words as
 * int32 rows.  Hypothesis k of utterance b: tokens_u8 + offsets[b] + k * length[b] (words likewise, NULL for
 * the lexicon-free decoder); scores as fltx_result_fetch_batch.  C4's batch of 256: 308 MB -> 23 MB.
 * Replaces the same loop of getAllFinalHypothesis() calls (Decoder.h:71-73). */
FLTX_API int fltx_result_fetch_batch_compact(fltx_decoder* dec, const int32_t** n_hyp, const int32_t** length,
                                             const double** scores, const uint8_t** tokens_u8,
                                             const int32_t** words, const int64_t** offsets);
/* getBestHypothesis(lookBack) of stream b (LexiconFreeDecoder.cpp:188-194,
 * decoder/Utils.h:268-310): *length = 0 for an empty result. */
FLTX_API int fltx_result_best(fltx_decoder* dec, int32_t b, int32_t look_back,
                              double* scores, int32_t* tokens, int32_t* words,
                              int32_t capacity, int32_t* length);
/* Device-resident results of the last batch (no host copy): pointers into HBM
 * valid until the next decode call.  n_hyp: int32[B]; scores: double[B*K*3];
 * tokens/words: int32 at tok_off[b] + i*(T[b]+2) + f. */
FLTX_API int fltx_result_device(fltx_decoder* dec, const int32_t** n_hyp,
                                const double** scores, const int32_t** tokens,
                                const int32_t** words, const int64_t** tok_off);

/* ---- one batch over several devices -------------------------------------- */
/* Utterances are independent: a group holds one context
 * and one decoder per entry of `devices` (an index may repeat: two contexts on
 * one device), the trie of `htrie` and the tables of `lm` replicated on each.
 * fltx_group_decode_batch cuts the batch into contiguous shards of about equal
 * frame count and runs fltx_decode_batch on every shard from its own host
 * thread; there is no inter-device traffic.  Replaces the loop over
 * Decoder::decode (decoder/Decoder.h:51-57) a multi-GPU caller would write.
 * emissions[i] is the buffer device i reads its shard from (the same host
 * pointer for all, or per-device HBM pointers with on_device[i] != 0); offsets
 * (NULL: utterances packed back to back) index into it with the caller's
 * utterance numbering.  Results are addressed by that numbering too. */
typedef struct fltx_group fltx_group;
FLTX_API int fltx_group_create(const int32_t* devices, int32_t n_devices, int32_t kind,
                               const fltx_options* opt, fltx_htrie* htrie, const fltx_lm* lm,
                               int32_t sil, int32_t blank, int32_t unk, const float* transitions,
                               int32_t n_transitions, int32_t is_lm_token, fltx_group** out);
FLTX_API int fltx_group_destroy(fltx_group* group);
FLTX_API int fltx_group_size(fltx_group* group, int32_t* n_devices);
/* decoder of part i and the utterances [first, first + count) it holds of the last batch */
FLTX_API int fltx_group_decoder(fltx_group* group, int32_t i, fltx_decoder** dec, int32_t* first,
                                int32_t* count);
FLTX_API int fltx_group_decode_batch(fltx_group* group, const float* const* emissions,
                                     const int32_t* on_device, const int64_t* offsets,
                                     const int32_t* T, int32_t B, int32_t N);
FLTX_API int fltx_group_result_count(fltx_group* group, int32_t b, int32_t* n_hyp, int32_t* length);
FLTX_API int fltx_group_result_fetch(fltx_group* group, int32_t b, int32_t max_hyp, double* scores,
                                     int32_t* tokens, int32_t* words, int32_t* n_copied);
FLTX_API int fltx_group_synchronize(fltx_group* group);

/* fltx_decoder_get(dec, "why_not_lane"): 0 when the last call started on a lane engine ("engine" 4 / 5 / 6), else the
 * eligibility terms it failed (the lane engines run the reference's LexiconFreeDecoder.cpp:30-125 /
 * LexiconDecoder.cpp:32-229 under these assumptions; everything else runs on the lean / generic engines) */
enum {
  FLTX_WHY_TOKENS = 1,        /* more than 64 tokens (lexicon-free decoder: a token BEAM of more than 64, or more than 16 384 tokens) */
  FLTX_WHY_BEAM = 2,          /* beam beyond the lane groups (lexicon-free: 512, lexicon: 256; 128 when spellings carry several words) */
  FLTX_WHY_STREAM = 4,        /* a stream the lane engines do not serve (lexicon streams, logAdd streams) */
  FLTX_WHY_LM = 8,            /* LM kind: a token-level LM on the lexicon decoder; a host LM (fltx_lm_host_create); an n-gram LM on the
                               * lexicon-free decoder whose contexts do not fit a dense table (more than 64 tokens, 2^24 contexts or 32 GB:
                               * round 6 -- otherwise the lane-state engine takes it at beams up to 512) */
  FLTX_WHY_LOGADD = 16,       /* lexicon-free decoder with logAdd over more than 64 tokens (round 5: the lexicon lane engines, 5 / 6,
                               * take logAdd wherever they take max-merge) */
  FLTX_WHY_ASG = 32,          /* lexicon decoder with the ASG criterion */
  FLTX_WHY_UNK = 64,          /* lexicon decoder with <unk> enabled (unk_score > -inf) */
  FLTX_WHY_TRIE_SHAPE = 128,  /* trie without a breadth-first layout (not a tree, a word that ends without the separator); several
                               * words per spelling (Trie.h:19) under ZeroLM -- those words tie in one LM state (round 5: with an
                               * n-gram LM such lexicons, the reference's own test lexicon among them, run on fltx_ylane.h) */
  FLTX_WHY_WORD_END = 256,    /* words do not all end in the separator (= sil), or sil == blank */
  FLTX_WHY_OPTIONS = 512,     /* negative beam threshold, sil / blank outside the token set */
  FLTX_WHY_LENGTH = 1024,     /* beam x frames beyond the state-id width of the history records */
  FLTX_WHY_SWITCHED_OFF = 2048, /* a tunable switched the engine off / a fallback is in force */
  FLTX_WHY_GEOMETRY = 4096    /* no compiled (threads, positions) geometry covers the token list */
};

/* ---- introspection for bench.py ------------------------------------------ */
/* Frames decoded and kernel launches issued by the last decode call, plus the
 * algorithmic HBM bytes of SURVEY.md section 8(d) for it. */
FLTX_API int fltx_decoder_stats(fltx_decoder* dec, int64_t* frames,
                                int64_t* algorithmic_bytes, int32_t* threads_per_utt,
                                int32_t* lds_bytes);
/* The same algorithmic bytes split by kernel: the decode kernel's share (emission rows in,
 * one back-pointer record per surviving slot out, trie gathers, plus 16 * order bytes per
 * n-gram LM query the kernel counted: *lm_bytes, included in *decode_bytes) and the
 * back-trace epilogue's (16 bytes per step of each hypothesis actually returned). */
FLTX_API int fltx_decoder_bytes(fltx_decoder* dec, int64_t* decode_bytes, int64_t* epilogue_bytes,
                                int64_t* lm_bytes);
/* Durations (ms) of the decode kernel and of the back-trace kernel of the last
 * fltx_decode_batch, from HIP events recorded on the context stream. */
FLTX_API int fltx_decoder_timing(fltx_decoder* dec, float* decode_ms, float* backtrace_ms);
/* Phase profile of the last launch (after fltx_decoder_set(dec,"profile",1)):
 * out[8] shader clocks summed over the batch. */
FLTX_API int fltx_decoder_profile(fltx_decoder* dec, uint64_t* out);
/* Tunables: "threads" (threads per utterance: 64..1024), "force_global_ws",
 * "dense" (0 = use the generic hash merge for lexicon-free frames too), "lean" /
 * "lane" / "slane" (0 = do not use that specialised lexicon-free + ZeroLM frame
 * step), "slane_threads", "xlane" / "ylane" (0 = do not use that lane engine of the
 * lexicon decoder; "ylane" = 2 prefers fltx_ylane.h where both apply), "keep_scores", "profile", "profile_wave"; lexicon decoder: "cut" (0 = build
 * every candidate's record), "slim" (0 = recompute form of the cut-off
 * generation), "items" (0 = no child-mask item list); test hooks: "cut_m",
 * "lds_budget" (pretend the CU has fewer bytes of LDS), "hot_level".
 * Round 3: "yshare" (-1 = the lexicon lane engines take the geometry of which several workgroups share a CU when the
 * batch exceeds the CUs; 1 / 0 = always / never), "stream_total_frames" (set before fltx_stream_begin: frames the
 * stream will decode in all -- its LM-state id tables grow with the stream, not with max_frames; default: at least
 * 2048), "sstream" (0 = a lexicon-free stream's chunks stay on the lane-per-slot step), "stream_optimistic" (0 = lexicon
 * streams use the worst-case HBM workspace from the start instead of decoding an overflowing chunk again),
 * "stream_defer" (0 = fltx_stream_step of a lexicon stream waits for its chunk; default: the chunk is launched and whether
 * a stream has to decode it again is looked at by the next call that needs the beam -- the next chunk's upload runs
 * under the kernel; a deferred fltx_stream_prune reports its errors there too), "lm_cache" (0 = the generic step asks
 * the n-gram tables for every word-end candidate instead of keeping the last (LM state, word) answers), "bt_lds_kb".  fltx_decoder_get also answers "engine", "redone", "stream_redone", "yshare", "sstream". */
/* Round 5: "defer_check" (1 = fltx_decode_batch returns as soon as its kernels are queued.  By default the call waits
 * for the decode kernel when the batch ran on a fast path that may flag an utterance, decodes the flagged ones again
 * and only then queues the back-trace -- every read of a device buffer is then queued before the call returns.  With
 * defer_check the look at the statuses, and that second pass, wait for the first call that reads results or "redone":
 * the caller keeps a device emissions buffer unchanged until then.  Batches of several decoder objects / streams
 * then run side by side on the CUs), "compact_always" (tests: streams rebuild their LM-state ids before every chunk). */
FLTX_API int fltx_decoder_set(fltx_decoder* dec, const char* key, int64_t value);
/* Geometry chosen for the last batch: "engine" (0 generic hash merge, 1 generic
 * dense merge, 2 lean register-resident step, 3 lane-per-slot step, 4 lane = LM
 * state step, 5 lane = (LM state, trie node) step of the lexicon decoder, 6 the
 * same with the LM terms: n-gram word LM, smeared trie, beams up to 128),
 * "lane" / "slane" / "xlane" (tokens per wave of engine 3 / 4 / 5, else 0),
 * "ylane" (lane groups of engine 6, else 0),
 * "redone" (utterances of the last offline call that the fast path handed to a
 * general one), "threads", "lds" (1 = workspace in LDS),
 * "hot_level" (HBM workspace: 1 = counters in LDS, 2 = candidate records too),
 * "cut" (candidates kept by the cut-off generation, 0 = off), "recompute",
 * "cap", "cap2", "items".
 * Round 5: "staged_emissions" (address of the library's own device copy of the last offline batch's host emissions, 0
 * when the caller passed a device buffer; valid until the decoder's next call -- a binding can decode the same batch
 * again from it with other settings, e.g. with "keep_scores" for Decoder::getBestHypothesis(lookBack) after decode()),
 * "compactions" / "id_cap" (streams: times the LM-state ids were rebuilt, ids per stream), "hlm_asked" / "hlm_distinct"
 * (host LM: questions listed / put to the callbacks), "fallback_reasons" (bit r: fltx_ylane.h's reason r, its header). */
FLTX_API int fltx_decoder_get(fltx_decoder* dec, const char* key, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* FLTX_H_ */
