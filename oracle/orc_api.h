/*
 * oracle/orc_api.h -- C interface shared by the two CPU checkers:
 *
 *   * oracle/oracle.cpp      (prefix orc_): this repo's CPU RESTATEMENT of the
 *                             reference algorithm.
 *   * oracle/ref_driver.cpp  (prefix ref_): a thin driver around the UNMODIFIED
 *                             reference sources compiled from /root/reference
 *                             into oracle/_ref/ (dev container only).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under text_amd/ may include, link or load
 * anything from oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed baseline.
 *
 * Both libraries export the same functions; ORC_FN(name) expands to orc_name or
 * ref_name depending on ORC_PREFIX_REF.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_PREFIX_REF
#define ORC_FN(name) ref_##name
#else
#define ORC_FN(name) orc_##name
#endif

/* Mirrors LexiconDecoderOptions (decoder/LexiconDecoder.h:21-31); the
 * lexicon-free decoder ignores word_score / unk_score
 * (decoder/LexiconFreeDecoder.h:20-28). criterion: 0 = ASG, 1 = CTC
 * (decoder/Decoder.h:16). */
typedef struct orc_options {
  int32_t beam_size;
  int32_t beam_size_token;
  double beam_threshold;
  double lm_weight;
  double word_score;
  double unk_score;
  double sil_score;
  int32_t log_add;
  int32_t criterion;
} orc_options;

/* ---- language models ---------------------------------------------------- */
/* ZeroLM (decoder/lm/ZeroLM.cpp:14-26). */
void* ORC_FN(lm_zero_create)(void);
/* ARPA back-off n-gram LM standing in for KenLM (decoder/lm/KenLM.cpp:32-83;
 * KenLM itself is an absent third-party dependency, see oracle/arpa_lm.h).
 * `usr_words` = '\n'-joined user dictionary entries, index i = usr idx i
 * (KenLM.cpp:44-49 builds the usr->LM id map from exactly this). */
void* ORC_FN(lm_arpa_create)(const char* arpa_path, const char* usr_words);
/* A user-defined LM whose state is its last input only: ONE LMState object per index, shared by every history that
 * ends in it (tests/host_lms.py LastWordLM is its Python twin).  Hypotheses with different histories then hold the
 * same LMState and merge -- on the state's address in the reference (decoder/lm/LM.h:37-49).  Scores are a fixed
 * integer hash of (previous index, index, seed) over 2^13, exact in float. */
void* ORC_FN(lm_lastword_create)(int32_t n_idx, int32_t seed);
void ORC_FN(lm_destroy)(void* lm);
/* score a word sequence from start(false); per_word[i] = score of word i;
 * returns total incl. finish() when with_finish (DecoderTest.cpp:107-120). */
float ORC_FN(lm_score_sequence)(void* lm, const int32_t* words, int32_t n,
                                int32_t with_finish, float* per_word);

/* ---- trie (decoder/Trie.h:64-92) ---------------------------------------- */
void* ORC_FN(trie_create)(int32_t max_children, int32_t root_idx);
/* returns 0, or -1 for an out-of-range index (Trie.cpp:31-34 throws) */
int32_t ORC_FN(trie_insert)(void* trie, const int32_t* indices, int32_t n,
                            int32_t label, float score);
/* mode: 0 NONE, 1 MAX, 2 LOGADD (Trie.h:21-25) */
void ORC_FN(trie_smear)(void* trie, int32_t mode);
/* returns 1 and *max_score if the path exists, else 0 */
int32_t ORC_FN(trie_search)(void* trie, const int32_t* indices, int32_t n,
                            float* max_score, int32_t* n_labels);
int64_t ORC_FN(trie_num_nodes)(void* trie);
void ORC_FN(trie_destroy)(void* trie);

/* ---- decoders (decoder/Decoder.h:36-74) ---------------------------------- */
void* ORC_FN(decoder_create_lexfree)(const orc_options* opt, void* lm,
                                     int32_t sil, int32_t blank,
                                     const float* transitions,
                                     int32_t n_transitions);
void* ORC_FN(decoder_create_lexicon)(const orc_options* opt, void* trie,
                                     void* lm, int32_t sil, int32_t blank,
                                     int32_t unk, const float* transitions,
                                     int32_t n_transitions,
                                     int32_t is_lm_token);
void ORC_FN(decoder_destroy)(void* dec);
void ORC_FN(decoder_begin)(void* dec);
void ORC_FN(decoder_step)(void* dec, const float* emissions, int32_t T,
                          int32_t N);
void ORC_FN(decoder_end)(void* dec);
void ORC_FN(decoder_prune)(void* dec, int32_t look_back);
int32_t ORC_FN(decoder_n_frames_in_buffer)(void* dec);
/* Number of hypotheses getAllFinalHypothesis() would return and the common
 * length (finalFrame + 1) of their tokens/words vectors. */
int32_t ORC_FN(decoder_n_final)(void* dec, int32_t* length);
/* Copy out up to max_hyp hypotheses: scores[3*i+{0,1,2}] = score,
 * emittingModelScore, lmScore; tokens/words [i*length + f]. Returns count. */
int32_t ORC_FN(decoder_get_all)(void* dec, int32_t max_hyp, double* scores,
                                int32_t* tokens, int32_t* words);
/* getBestHypothesis(lookBack): returns length (0 = empty result). */
int32_t ORC_FN(decoder_get_best)(void* dec, int32_t look_back, double* scores,
                                 int32_t* tokens, int32_t* words,
                                 int32_t capacity);

/* ORACLE ONLY (no ref_ twin: the reference has no such counters).  Ties the decoder has passed since it was created
 * or since the last call with reset != 0 -- the places where the reference's answer depends on addresses (oracle.cpp
 * TieCounts): out[0] merge, [1] cut, [2] order, [3] token, [4] best.  A parity mismatch on an input for which all five
 * are zero is a bug, never "a tie". */
void orc_decoder_ties(void* dec, int64_t* out, int32_t reset);

#ifdef __cplusplus
}
#endif
