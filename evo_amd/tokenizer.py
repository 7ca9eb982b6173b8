"""Byte-level tokenizer with the interface of the reference's CharLevelTokenizer
[REF evo/tokenizer.py:15-58]: ids are the UTF-8 bytes of the text, vocabulary 512, EOD/EOS id 0,
PAD id 1; decoding clamps ids into [32, vocab_size] before `chr`."""
from typing import List, Union

import numpy as np
import torch


class CharLevelTokenizer:
    def __init__(self, vocab_size: int = 512):
        self.name = "CharLevelTokenizer"
        self._vocab_size = vocab_size
        self.eod_id = 0
        self.eos_id = 0
        self.pad_id = 1

    # -- properties the generation loop reads [REF evo/generation.py:58-62,99-101]
    @property
    def vocab_size(self) -> int:
        return self._vocab_size

    @property
    def eod(self) -> int:
        return self.eod_id

    @property
    def eos(self) -> int:
        return self.eod_id

    # -- text -> ids
    def tokenize(self, text: str) -> List[np.uint8]:
        """list of np.uint8, one per UTF-8 byte (the element type the reference returns)."""
        return [np.uint8(b) for b in text.encode()]

    def tokenize_batch(self, text_batch: Union[List[str], str]):
        if isinstance(text_batch, list):
            return [self.tokenize(t) for t in text_batch]
        return self.tokenize(text_batch)

    # -- ids -> text
    def clamp(self, n: int) -> int:
        lo, hi = 32, self.vocab_size
        return hi if n > hi else (lo if n < lo else n)

    def decode_token(self, token: int) -> str:
        return str(chr(self.clamp(token)))

    def detokenize(self, token_ids) -> str:
        return "".join(self.decode_token(t) for t in token_ids)

    def detokenize_batch(self, token_ids):
        if isinstance(token_ids, torch.Tensor):
            token_ids = token_ids.tolist()
            return [self.detokenize(row) for row in token_ids]
        if isinstance(token_ids, list):
            return [self.detokenize(row) for row in token_ids]
        return self.detokenize(token_ids)
