"""Mirror of bindings/python/flashlight/lib/text/decoder/kenlm.py: KenLM(path, usr_token_dict)."""
from text_amd.flashlight_lib_text_decoder import KenLM  # noqa: F401
