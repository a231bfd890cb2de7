"""Mirror of bindings/python/flashlight/lib/text/dictionary.py."""
from text_amd.flashlight_lib_text_decoder import (  # noqa: F401
    Dictionary, create_word_dict, load_words, pack_replabels, tkn_to_idx, unpack_replabels)
