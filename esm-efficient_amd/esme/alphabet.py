"""Protein alphabets and the two tokenisers that feed the packed forward path.

Host-side (CPU) code: it produces the `(tokens, (cu_lens, max_len))` inputs the
HIP path consumes.  Mirrors the public names of the reference
(`esme/alphabet.py:9-56` alphabets, `:117-145` tokenize, `:148-183`
tokenize_unpad, `:268-286` padding_mask) so callers can switch packages.

Token ids are a *data contract* (they index the embedding table of released
checkpoints), so the vocab order is fixed; everything else here is this
project's own code.
"""
from __future__ import annotations

import re
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

_SPECIAL_HEAD = ('<cls>', '<pad>', '<eos>', '<unk>')
_RESIDUES = tuple('LAGVSERTIDPKQNFYMHWCXBUZO')      # ids 4..28
_TOKEN_RE = re.compile(r'<[^>]+>|.')


class _Vocab:
    """Base: builds the lookup tables from `alphabet` at class-creation time."""
    alphabet: List[str] = []

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        vocab = list(cls.alphabet)
        cls.idx_to_token = {i: t for i, t in enumerate(vocab)}
        cls.token_to_idx = {t: i for i, t in enumerate(vocab)}
        cls.amino_acids = vocab[4:24]
        cls.amino_acids_idx = list(range(4, 24))
        cls.cls_idx = cls.token_to_idx['<cls>']
        cls.eos_idx = cls.token_to_idx['<eos>']
        cls.padding_idx = cls.token_to_idx['<pad>']
        cls.unk_idx = cls.token_to_idx['<unk>']
        cls.mask_idx = cls.token_to_idx['<mask>']

    @classmethod
    def encode(cls, pieces: Sequence[str]) -> List[int]:
        """`<cls>` + ids (unknown pieces -> `<unk>`) + `<eos>`."""
        get, unk = cls.token_to_idx.get, cls.unk_idx
        return [cls.cls_idx, *(get(p, unk) for p in pieces), cls.eos_idx]


class Alphabet(_Vocab):
    """33-token vocabulary of ESM-1b / ESM-1v / ESM-2 (reference alphabet.py:9-31)."""
    alphabet = [*_SPECIAL_HEAD, *_RESIDUES, '.', '-', '<null_1>', '<mask>']


class Alphabet3(_Vocab):
    """33 used ids of the 64-row ESM-C embedding (reference alphabet.py:34-56)."""
    alphabet = [*_SPECIAL_HEAD, *_RESIDUES, '.', '-', '|', '<mask>']


def split_alphabet(seq: Union[str, Sequence[str]]):
    """'MP<mask>A' -> ['M','P','<mask>','A']; a list of strings maps element-wise."""
    if isinstance(seq, str):
        return _TOKEN_RE.findall(seq)
    return [_TOKEN_RE.findall(s) for s in seq]


def token_to_str(tokens: torch.Tensor, alphabet=Alphabet3) -> List[str]:
    table = alphabet.idx_to_token
    return [''.join(table[i] for i in row) for row in tokens.tolist()]


def _encode_all(sequences, alphabet):
    if isinstance(sequences, str):
        sequences = [sequences]
    return [alphabet.encode(p) for p in split_alphabet(list(sequences))]


def tokenize(sequences: Union[List[str], str], alphabet=Alphabet3) -> torch.Tensor:
    """Padded (B, max_len) int64 ids, `<pad>` on the right of short rows."""
    rows = _encode_all(sequences, alphabet)
    width = max(len(r) for r in rows)
    out = np.full((len(rows), width), alphabet.padding_idx, dtype=np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return torch.from_numpy(out)


def tokenize_unpad(sequences: Union[List[str], str], alphabet=Alphabet3
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
    """Packed ids + the scatter indices into the padded layout + cu_lens + max_len.

    Returns `(tokens int64 (T,), indices int64 (T,), cu_lens int32 (B+1,), max_len)`.
    `indices[t]` is the flat position row*max_len+col the packed token t would
    occupy in the `tokenize` output (what `pad_input` scatters with).
    """
    rows = _encode_all(sequences, alphabet)
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    max_len = int(lens.max())
    cu = np.zeros(len(rows) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    tokens = np.concatenate([np.asarray(r, dtype=np.int64) for r in rows])
    seq_of = np.repeat(np.arange(len(rows), dtype=np.int64), lens)
    col = np.arange(tokens.shape[0], dtype=np.int64) - cu[:-1].astype(np.int64)[seq_of]
    indices = seq_of * max_len + col
    return (torch.from_numpy(tokens), torch.from_numpy(indices),
            torch.from_numpy(cu), max_len)


def padding_mask(cu_lens: torch.Tensor, max_len: int) -> torch.Tensor:
    """(B, max_len) bool, True where a real token sits (reference alphabet.py:268-286)."""
    lengths = (cu_lens[1:] - cu_lens[:-1]).unsqueeze(1)
    cols = torch.arange(max_len, device=cu_lens.device).unsqueeze(0)
    return cols < lengths


def pad_tokens(tokens: List[torch.Tensor], alphabet=Alphabet3) -> torch.Tensor:
    """Right-pad a list of token tensors to a common length and stack them: 1-D items -> (B, L),
    (1, L_i) items -> (B, L) (reference alphabet.py:186-213)."""
    rows = [t.reshape(-1) for t in tokens]
    width = max(r.numel() for r in rows)
    out = torch.full((len(rows), width), alphabet.padding_idx, dtype=rows[0].dtype)
    for i, r in enumerate(rows):
        out[i, :r.numel()] = r
    return out
