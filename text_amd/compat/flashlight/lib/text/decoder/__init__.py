"""Mirror of bindings/python/flashlight/lib/text/decoder/__init__.py:12-32."""
from text_amd.flashlight_lib_text_decoder import (  # noqa: F401
    CriterionType, DecodeResult, KenLM, LexiconDecoder, LexiconDecoderOptions, LexiconFreeDecoder,
    LexiconFreeDecoderOptions, LM, LMState, SmearingMode, Trie, TrieNode, ZeroLM)
