"""`flashlight.lib.text` namespace backed by the MI355X decoder.

Put `<repo>/text_amd/compat` on sys.path (before any installed flashlight-text)
and existing code such as

    from flashlight.lib.text.decoder import LexiconDecoder, LexiconDecoderOptions, Trie, ZeroLM
    from flashlight.lib.text.dictionary import Dictionary, load_words, create_word_dict

runs on the HIP path (mirrors bindings/python/flashlight/lib/text/ of the reference)."""
