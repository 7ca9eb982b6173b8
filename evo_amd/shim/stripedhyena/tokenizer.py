from evo_amd.tokenizer import CharLevelTokenizer  # noqa: F401
