/*
 * py_decoder.cpp -- pybind11 module with the Python surface of the reference's
 * decoder bindings (bindings/python/flashlight/lib/text/_decoder.cpp:167-441,
 * _dictionary.cpp:33-60, decoder/_kenlm.cpp:19-25) over this repo's facade
 * classes, so `from flashlight.lib.text.decoder import ...` code runs on the
 * MI355X path (see text_amd/compat/).  Same class names, constructor keywords,
 * method names, raw-address emissions (`ndarray.ctypes.data`) and pickle
 * support; seq2seq classes are out of scope.  Additive: decode_batch().
 */
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <functional>
#include <limits>

#include "flashlight/lib/text/decoder/LexiconDecoder.h"
#include "flashlight/lib/text/decoder/LexiconFreeDecoder.h"
#include "flashlight/lib/text/decoder/Trie.h"
#include "flashlight/lib/text/decoder/lm/KenLM.h"
#include "flashlight/lib/text/decoder/lm/ZeroLM.h"
#include "flashlight/lib/text/dictionary/Utils.h"

namespace py = pybind11;
using namespace fl::lib::text;
using namespace py::literals;

namespace {

/* Python subclasses of LM (the reference's PyLM trampoline, _decoder.cpp:39-56).  A decoder runs them through the
 * per-frame host exchange of decoder/lm/HostLM.h: the search on the device, start / score / finish in Python. */
class PyLM : public LM {
 public:
  using LM::LM;
  using LMOutput = std::pair<LMStatePtr, float>;
  LMStatePtr start(bool startWithNothing) override {
    PYBIND11_OVERRIDE_PURE(LMStatePtr, LM, start, startWithNothing);
  }
  LMOutput score(const LMStatePtr& state, const int usrTokenIdx) override {
    PYBIND11_OVERRIDE_PURE(LMOutput, LM, score, state, usrTokenIdx);
  }
  LMOutput finish(const LMStatePtr& state) override { PYBIND11_OVERRIDE_PURE(LMOutput, LM, finish, state); }
};

const float* asPtr(uintptr_t p) { return reinterpret_cast<const float*>(p); }

/* additive: the n-best of a whole batch as NumPy arrays over the library's pinned host buffers
 * (zero copy; valid until the next decode on the decoder, which the arrays keep alive).  Indexing
 * builds the DecodeResult objects of ONE utterance: Python code that wants the best hypothesis of
 * every utterance, or arrays, does not pay for 12 800 objects per batch. */
struct BatchResults {
  py::object owner; /* the decoder */
  detail::BatchView v;
  std::function<std::vector<DecodeResult>(int)> make;
  py::array nHyp, length, offsets, scores, tokens;
  py::object words;
};

template <class Dec>
BatchResults makeBatchResults(py::object self, Dec& d, const detail::BatchView& v) {
  BatchResults r;
  r.owner = self;
  r.v = v;
  Dec* dp = &d;
  r.make = [dp, v](int b) { return dp->materialise(v, b); };
  const py::ssize_t B = v.B;
  const py::ssize_t total = v.offsets[v.B];
  r.nHyp = py::array_t<int32_t>({B}, {sizeof(int32_t)}, v.nHyp, self);
  r.length = py::array_t<int32_t>({B}, {sizeof(int32_t)}, v.length, self);
  r.offsets = py::array_t<int64_t>({B + 1}, {sizeof(int64_t)}, v.offsets, self);
  r.scores = py::array_t<double>({B, (py::ssize_t)v.K, (py::ssize_t)3},
                                 {(py::ssize_t)(sizeof(double) * 3 * v.K), (py::ssize_t)(sizeof(double) * 3),
                                  (py::ssize_t)sizeof(double)},
                                 v.scores, self);
  r.tokens = py::array_t<int32_t>({total}, {sizeof(int32_t)}, v.tokens, self);
  r.words = v.words ? py::object(py::array_t<int32_t>({total}, {sizeof(int32_t)}, v.words, self)) : py::none();
  return r;
}

template <class Dec>
void bindDecoderMethods(py::class_<Dec>& c) {
  /* (the calls that wait for the device release the GIL -- the reference holds it throughout, _decoder.cpp has no
   * gil_scoped_release; here a thread per decoder object keeps several utterances on the device at a time.  A Python
   * LM's start / score / finish take it back inside their trampolines) */
  c.def("decode_begin", &Dec::decodeBegin)
      .def("decode_step",
           [](Dec& d, uintptr_t e, int T, int N) {
             py::gil_scoped_release nogil;
             d.decodeStep(asPtr(e), T, N);
           },
           "emissions"_a, "T"_a, "N"_a)
      .def("decode_end",
           [](Dec& d) {
             py::gil_scoped_release nogil;
             d.decodeEnd();
           })
      .def("decode",
           [](Dec& d, uintptr_t e, int T, int N) {
             std::vector<DecodeResult> r;
             {
               py::gil_scoped_release nogil;
               r = d.decode(asPtr(e), T, N);
             }
             return r;
           },
           "emissions"_a, "T"_a, "N"_a)
      .def("prune",
           [](Dec& d, int lookBack) {
             py::gil_scoped_release nogil;
             d.prune(lookBack);
           },
           "look_back"_a = 0)
      .def("get_best_hypothesis", &Dec::getBestHypothesis, "look_back"_a = 0)
      .def("get_all_final_hypothesis", &Dec::getAllFinalHypothesis)
      .def("n_hypothesis", &Dec::nHypothesis)
      .def("n_decoded_frames_in_buffer", &Dec::nDecodedFramesInBuffer)
      /* additive: B utterances packed back to back at `emissions`; devices = [0, 1, ...] shards the
       * batch over several GPUs (one host thread + stream per device, results in input order) */
      .def("decode_batch",
           [](Dec& d, uintptr_t e, const std::vector<int>& T, int N, bool onDevice, const std::vector<int>& devices) {
             py::gil_scoped_release nogil;
             if (!devices.empty()) {
               return d.decodeBatch(asPtr(e), T, N, devices);
             }
             return d.decodeBatch(asPtr(e), T, N, std::vector<int64_t>{}, onDevice);
           },
           "emissions"_a, "T"_a, "N"_a, "on_device"_a = false, "devices"_a = std::vector<int>{})
      .def("decode_batch_arrays",
           [](py::object self, uintptr_t e, const std::vector<int>& T, int N, bool onDevice) {
             Dec& d = self.cast<Dec&>();
             detail::BatchView v;
             {
               py::gil_scoped_release nogil;
               v = d.decodeBatchView(asPtr(e), T, N, {}, onDevice);
             }
             return makeBatchResults(self, d, v);
           },
           "emissions"_a, "T"_a, "N"_a, "on_device"_a = false)
      .def("set_max_stream_frames", &Dec::setMaxStreamFrames, "frames"_a);
}

} // namespace

PYBIND11_MODULE(flashlight_lib_text_decoder, m) {
  m.doc() = "flashlight/text decoder bindings backed by the MI355X kernels (libfltx)";

  py::enum_<SmearingMode>(m, "SmearingMode")
      .value("NONE", SmearingMode::NONE)
      .value("MAX", SmearingMode::MAX)
      .value("LOGADD", SmearingMode::LOGADD);

  py::class_<TrieNode, TrieNodePtr>(m, "TrieNode")
      .def(py::init<int>(), "idx"_a)
      .def_readwrite("children", &TrieNode::children)
      .def_readwrite("idx", &TrieNode::idx)
      .def_readwrite("labels", &TrieNode::labels)
      .def_readwrite("scores", &TrieNode::scores)
      .def_readwrite("max_score", &TrieNode::maxScore);

  py::class_<Trie, TriePtr>(m, "Trie")
      .def(py::init<int, int>(), "max_children"_a, "root_idx"_a)
      .def("get_root", &Trie::getRoot, py::return_value_policy::reference_internal)
      .def("insert", &Trie::insert, "indices"_a, "label"_a, "score"_a)
      .def("search", &Trie::search, "indices"_a)
      .def("smear", &Trie::smear, "smear_mode"_a);

  py::class_<LM, LMPtr, PyLM>(m, "LM")
      .def(py::init<>())
      .def("start", &LM::start, "start_with_nothing"_a)
      .def("score", &LM::score, "state"_a, "usr_token_idx"_a)
      .def("finish", &LM::finish, "state"_a);

  py::class_<LMState, LMStatePtr>(m, "LMState")
      .def(py::init<>())
      .def_readwrite("children", &LMState::children)
      .def("compare", &LMState::compare, "state"_a)
      .def("child", &LMState::child<LMState>, "usr_index"_a);

  py::class_<ZeroLM, ZeroLMPtr, LM>(m, "ZeroLM").def(py::init<>());
  py::class_<KenLM, KenLMPtr, LM>(m, "KenLM")
      .def(py::init<const std::string&, const Dictionary&>(), "path"_a, "usr_token_dict"_a);

  py::enum_<CriterionType>(m, "CriterionType")
      .value("ASG", CriterionType::ASG)
      .value("CTC", CriterionType::CTC)
      .value("S2S", CriterionType::S2S);

  py::class_<LexiconDecoderOptions>(m, "LexiconDecoderOptions")
      .def(py::init<const int, const int, const double, const double, const double, const double,
                    const double, const bool, const CriterionType>(),
           "beam_size"_a, "beam_size_token"_a, "beam_threshold"_a, "lm_weight"_a, "word_score"_a,
           "unk_score"_a, "sil_score"_a, "log_add"_a, "criterion_type"_a)
      .def_readwrite("beam_size", &LexiconDecoderOptions::beamSize)
      .def_readwrite("beam_size_token", &LexiconDecoderOptions::beamSizeToken)
      .def_readwrite("beam_threshold", &LexiconDecoderOptions::beamThreshold)
      .def_readwrite("lm_weight", &LexiconDecoderOptions::lmWeight)
      .def_readwrite("word_score", &LexiconDecoderOptions::wordScore)
      .def_readwrite("unk_score", &LexiconDecoderOptions::unkScore)
      .def_readwrite("sil_score", &LexiconDecoderOptions::silScore)
      .def_readwrite("log_add", &LexiconDecoderOptions::logAdd)
      .def_readwrite("criterion_type", &LexiconDecoderOptions::criterionType)
      .def(py::pickle(
          [](const LexiconDecoderOptions& p) {
            return py::make_tuple(p.beamSize, p.beamSizeToken, p.beamThreshold, p.lmWeight, p.wordScore,
                                  p.unkScore, p.silScore, p.logAdd, p.criterionType);
          },
          [](py::tuple t) {
            if (t.size() != 9) {
              throw std::runtime_error(
                  "Cannot run __setstate__ on LexiconDecoderOptions - insufficient arguments provided.");
            }
            return LexiconDecoderOptions{t[0].cast<int>(),    t[1].cast<int>(),    t[2].cast<double>(),
                                         t[3].cast<double>(), t[4].cast<double>(), t[5].cast<double>(),
                                         t[6].cast<double>(), t[7].cast<bool>(),   t[8].cast<CriterionType>()};
          }));

  py::class_<LexiconFreeDecoderOptions>(m, "LexiconFreeDecoderOptions")
      .def(py::init<const int, const int, const double, const double, const double, const bool,
                    const CriterionType>(),
           "beam_size"_a, "beam_size_token"_a, "beam_threshold"_a, "lm_weight"_a, "sil_score"_a, "log_add"_a,
           "criterion_type"_a)
      .def_readwrite("beam_size", &LexiconFreeDecoderOptions::beamSize)
      .def_readwrite("beam_size_token", &LexiconFreeDecoderOptions::beamSizeToken)
      .def_readwrite("beam_threshold", &LexiconFreeDecoderOptions::beamThreshold)
      .def_readwrite("lm_weight", &LexiconFreeDecoderOptions::lmWeight)
      .def_readwrite("sil_score", &LexiconFreeDecoderOptions::silScore)
      .def_readwrite("log_add", &LexiconFreeDecoderOptions::logAdd)
      .def_readwrite("criterion_type", &LexiconFreeDecoderOptions::criterionType)
      .def(py::pickle(
          [](const LexiconFreeDecoderOptions& p) {
            return py::make_tuple(p.beamSize, p.beamSizeToken, p.beamThreshold, p.lmWeight, p.silScore,
                                  p.logAdd, p.criterionType);
          },
          [](py::tuple t) {
            if (t.size() != 7) {
              throw std::runtime_error(
                  "Cannot run __setstate__ on LexiconFreeDecoderOptions - insufficient arguments provided.");
            }
            return LexiconFreeDecoderOptions{t[0].cast<int>(),    t[1].cast<int>(),  t[2].cast<double>(),
                                             t[3].cast<double>(), t[4].cast<double>(), t[5].cast<bool>(),
                                             t[6].cast<CriterionType>()};
          }));

  py::class_<DecodeResult>(m, "DecodeResult")
      .def(py::init<int>(), "length"_a)
      .def_readwrite("score", &DecodeResult::score)
      .def_readwrite("emittingModelScore", &DecodeResult::emittingModelScore)
      .def_readwrite("lmScore", &DecodeResult::lmScore)
      .def_readwrite("words", &DecodeResult::words)
      .def_readwrite("tokens", &DecodeResult::tokens);

  py::class_<BatchResults>(m, "BatchResults")
      .def_readonly("n_hyp", &BatchResults::nHyp)
      .def_readonly("length", &BatchResults::length)
      .def_readonly("offsets", &BatchResults::offsets)
      .def_readonly("scores", &BatchResults::scores)
      .def_readonly("tokens", &BatchResults::tokens)
      .def_readonly("words", &BatchResults::words)
      .def("__len__", [](const BatchResults& r) { return r.v.B; })
      .def("__getitem__",
           [](const BatchResults& r, int b) {
             if (b < 0) {
               b += r.v.B;
             }
             if (b < 0 || b >= r.v.B) {
               throw py::index_error();
             }
             return r.make(b);
           },
           "b"_a)
      .def("tokens_of",
           [](const BatchResults& r, int b, int i) {
             if (b < 0 || b >= r.v.B || i < 0 || i >= r.v.nHyp[b]) {
               throw py::index_error();
             }
             const py::ssize_t L = r.v.length[b];
             return py::array_t<int32_t>({L}, {sizeof(int32_t)}, r.v.tokens + r.v.offsets[b] + (int64_t)i * L,
                                         r.owner);
           },
           "b"_a, "i"_a = 0);

  py::class_<LexiconDecoder> lex(m, "LexiconDecoder");
  lex.def(py::init<LexiconDecoderOptions, const TriePtr, const LMPtr, const int, const int, const int,
                   const std::vector<float>&, const bool>(),
          "options"_a, "trie"_a, "lm"_a, "sil_token_idx"_a, "blank_token_idx"_a, "unk_token_idx"_a,
          "transitions"_a, "is_token_lm"_a,
          /* a Python subclass of LM lives in its Python object: the decoder keeps that alive, not just the C++ base
           * the shared_ptr holds (an LM passed as a temporary would otherwise lose its overrides) */
          py::keep_alive<1, 4>());
  bindDecoderMethods(lex);

  py::class_<LexiconFreeDecoder> lf(m, "LexiconFreeDecoder");
  lf.def(py::init<LexiconFreeDecoderOptions, const LMPtr, const int, const int, const std::vector<float>&>(),
         "options"_a, "lm"_a, "sil_token_idx"_a, "blank_token_idx"_a, "transitions"_a, py::keep_alive<1, 3>())
      .def("get_options", &LexiconFreeDecoder::getOptions)
      .def("get_sil_idx", &LexiconFreeDecoder::getSilIdx)
      .def("get_blank_idx", &LexiconFreeDecoder::getBlankIdx)
      .def("get_transitions", &LexiconFreeDecoder::getTransitions)
      .def(py::pickle(
          /* as the reference (_decoder.cpp:409-441): only a stateless decoder over ZeroLM pickles */
          [](const LexiconFreeDecoder& d) {
            if (!std::dynamic_pointer_cast<ZeroLM>(d.getLMPtr())) {
              throw std::runtime_error("LexiconFreeDecoder.__getstate__: only decoders using ZeroLM can be pickled");
            }
            return py::make_tuple(d.getOptions(), d.getSilIdx(), d.getBlankIdx(), d.getTransitions());
          },
          [](py::tuple t) {
            if (t.size() != 4) {
              throw std::runtime_error("Cannot run __setstate__ on LexiconFreeDecoder - insufficient arguments provided.");
            }
            return std::make_unique<LexiconFreeDecoder>(t[0].cast<LexiconFreeDecoderOptions>(),
                                                        std::make_shared<ZeroLM>(), t[1].cast<int>(),
                                                        t[2].cast<int>(), t[3].cast<std::vector<float>>());
          }));
  bindDecoderMethods(lf);

  /* ---- dictionary (bindings/python/flashlight/lib/text/_dictionary.cpp:33-60) ---- */
  py::class_<Dictionary>(m, "Dictionary")
      .def(py::init<>())
      .def(py::init([](const std::string& filename) { return loadDictionary(filename); }), "filename"_a)
      .def(py::init<const std::vector<std::string>&>(), "tkns"_a)
      .def("entry_size", &Dictionary::entrySize)
      .def("index_size", &Dictionary::indexSize)
      .def("add_entry", [](Dictionary& d, const std::string& e, int idx) { d.addEntry(e, idx); }, "entry"_a, "idx"_a)
      .def("add_entry", [](Dictionary& d, const std::string& e) { d.addEntry(e); }, "entry"_a)
      .def("get_entry", &Dictionary::getEntry, "idx"_a)
      .def("set_default_index", &Dictionary::setDefaultIndex, "idx"_a)
      .def("get_index", &Dictionary::getIndex, "entry"_a)
      .def("contains", &Dictionary::contains, "entry"_a)
      .def("is_contiguous", &Dictionary::isContiguous)
      .def("map_entries_to_indices", &Dictionary::mapEntriesToIndices, "entries"_a)
      .def("map_indices_to_entries", &Dictionary::mapIndicesToEntries, "indices"_a);
  m.def("create_word_dict", &createWordDict, "lexicon"_a);
  m.def("load_words", &loadWords, "filename"_a, "max_words"_a = -1);
  m.def("pack_replabels", &packReplabels, "tokens"_a, "dict"_a, "max_reps"_a);
  m.def("unpack_replabels", &unpackReplabels, "tokens"_a, "dict"_a, "max_reps"_a);
  m.def("tkn_to_idx", &tkn2Idx, "spelling"_a, "token_dict"_a, "maxReps"_a);
}
