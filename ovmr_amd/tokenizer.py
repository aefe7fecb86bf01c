"""Byte-level BPE tokenizer for CLIP prompts.

Same observable behaviour as the reference's `SimpleTokenizer.encode` / `clip.tokenize`
(clip/simple_tokenizer.py:62-132, clip/clip.py:187-223): lower-cased, whitespace-collapsed text is split with
the CLIP pattern, every piece is mapped byte -> printable code point, greedily merged by merge rank, and
looked up in the 49 408-entry vocabulary (SOT 49406, EOT 49407).  The merge table itself
(`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB) is NOT shipped here: pass its path (it is in every CLIP checkout,
e.g. `<reference>/clip/bpe_simple_vocab_16e6.txt.gz`) or set OVMR_BPE_PATH.

Written from the published algorithm (GPT-2 style byte-level BPE); `ftfy` is optional (identity for ASCII names).
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, List, Sequence, Tuple

import regex
import torch

from .synth import EOT_ID, SOT_ID

N_MERGES = 49152 - 256 - 2          # merges actually used by CLIP (clip/simple_tokenizer.py:67)
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)


def _byte_alphabet() -> List[str]:
    """256 printable stand-ins for the byte values: printable latin-1 bytes map to themselves, the other 68
    bytes to code points 256, 257, ... in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spare = [""] * 256, 256
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(spare)
            spare += 1
    return table


class BPETokenizer:
    def __init__(self, bpe_path: str = None):
        bpe_path = bpe_path or os.environ.get("OVMR_BPE_PATH")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP merge table not found: pass bpe_path= or set OVMR_BPE_PATH to "
                                    "bpe_simple_vocab_16e6.txt.gz")
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:1 + N_MERGES]]
        self.alphabet = _byte_alphabet()
        # vocabulary order: bytes in the order "kept bytes ascending, then the remapped ones", each also as word-final
        ordered = sorted(range(256), key=lambda b: (ord(self.alphabet[b]) >= 256, ord(self.alphabet[b])))
        symbols = [self.alphabet[b] for b in ordered]
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.token_id: Dict[str, int] = {t: i for i, t in enumerate(vocab)}
        self.rank: Dict[Tuple[str, str], int] = {m: i for i, m in enumerate(merges)}
        self._memo: Dict[str, List[int]] = {}
        assert self.token_id["<|startoftext|>"] == SOT_ID and self.token_id["<|endoftext|>"] == EOT_ID

    def _merge_word(self, piece: str) -> List[int]:
        got = self._memo.get(piece)
        if got is not None:
            return got
        parts = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(parts) > 1:
            best, where = None, -1
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, i
            if best is None:
                break
            a, b = parts[where], parts[where + 1]
            out, i = [], 0
            while i < len(parts):                      # merge EVERY occurrence of the best pair, left to right
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            parts = out
        ids = [self.token_id[p] for p in parts]
        self._memo[piece] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for piece in _SPLIT.findall(text):
            mapped = "".join(self.alphabet[b] for b in piece.encode("utf-8"))
            ids.extend(self._merge_word(mapped))
        return ids

    def tokenize(self, texts: Sequence[str], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """clip.tokenize: [SOT] + ids + [EOT], zero padded to context_length."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [SOT_ID] + self.encode(t) + [EOT_ID]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = EOT_ID
            out[i, :len(ids)] = torch.tensor(ids)
        return out
