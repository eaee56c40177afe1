"""Masked-marginal variant-effect scoring on the packed HIP forward path.

Same public names and return conventions as the reference (`esme/variant.py`:
MaskMarginDataset :10-70, predict_mask_margin :110-165, predict_pseudoperplexity
:168-215): one row per residue of the protein with that residue replaced by
`<mask>`, scored as log p(aa | context) - log p(wild type | context) at the masked
position for the 20 amino acids.

Re-planned for MI355X instead of translated:

* the rows of one batch all have the same length (a full protein, or a `max_len`
  window of it), so a batch IS a packed input with uniform `cu_lens` -- no pad /
  unpad pass is run;
* only ONE row per sequence (the masked position) is needed from the LM head, so the
  final-LayerNorm output is row-gathered (B rows) before the head GEMMs instead of
  projecting all B*L rows to the vocabulary and indexing afterwards.  The head is
  row-wise, so the selected rows are bit-identical to the reference's
  `predict_log_prob(..., pad_output=True)[arange, local_pos]`.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from esme import _hip
from esme.alphabet import Alphabet3, tokenize


class MaskMarginDataset(Dataset):
    """Item `i`: the tokenised protein with residue `i` (token `i+1`, after `<cls>`)
    masked.  With `max_len` the tokens are cut to a `max_len`-wide window centred on the
    masked position (clamped to the ends); `local_pos` is the masked index inside the
    window, `pos` the 1-based residue number.  Reference: esme/variant.py:39-70."""

    def __init__(self, seq: str, max_len: Optional[int] = None, alphabet=Alphabet3):
        super().__init__()
        self.seq, self.max_len, self.alphabet = seq, max_len, alphabet
        self.token = tokenize([seq])[0]          # default alphabet, like the reference (:43)

    def __len__(self):
        return len(self.seq)

    @property
    def window_len(self) -> int:
        n = self.token.numel()
        return n if self.max_len is None else min(n, self.max_len)

    def _window(self, pos: int):
        """(start, local_pos) of the window that holds token index `pos`."""
        n, w = self.token.numel(), self.max_len
        if w is None or n <= w:
            return 0, pos
        start = min(n - w, max(0, pos - w // 2))
        return start, pos - start

    def __getitem__(self, idx):
        if idx < 0:
            idx += len(self)
        wt = self.seq[idx]
        pos = idx + 1
        start, local = self._window(pos)
        token = self.token.clone()
        token[pos] = self.alphabet.mask_idx
        token = token[start:start + self.window_len]
        return {'token': token, 'local_pos': local, 'pos': pos, 'wt': wt,
                'wt_token': self.alphabet.token_to_idx[wt]}

    def batch(self, first: int, last: int) -> Dict[str, object]:
        """Items [first, last) collated without going through `__getitem__`:
        `token` (B, L) int64, `local_pos` / `pos` / `wt_token` int64 (B,), `wt` list."""
        last = min(last, len(self))
        pos = np.arange(first + 1, last + 1, dtype=np.int64)
        n, L = self.token.numel(), self.window_len
        if L < n:
            start = np.minimum(n - L, np.maximum(0, pos - self.max_len // 2))
        else:
            start = np.zeros_like(pos)
        base = self.token.numpy()
        rows = base[start[:, None] + np.arange(L, dtype=np.int64)[None, :]]
        local = pos - start
        rows[np.arange(rows.shape[0]), local] = self.alphabet.mask_idx
        wt = list(self.seq[first:last])
        return {'token': torch.from_numpy(rows), 'local_pos': torch.from_numpy(local),
                'pos': torch.from_numpy(pos), 'wt': wt,
                'wt_token': torch.tensor([self.alphabet.token_to_idx[a] for a in wt], dtype=torch.int64)}

    def batches(self, batch_size: int) -> Iterable[Dict[str, object]]:
        for first in range(0, len(self), batch_size):
            yield self.batch(first, first + batch_size)


def masked_row_log_prob(model, token: torch.Tensor, local_pos: torch.Tensor, graph: bool = False) -> torch.Tensor:
    """log-softmax over the vocabulary at `token[b, local_pos[b]]` for every row b of a
    (B, L) batch: (B, V) bf16 on the model's device.  `graph=True` replays the transformer stack from a
    hipGraph captured for this (B, L) (every full batch of one protein has the same shape)."""
    device = model.embed_tokens.weight.device
    token = token.to(device)
    B, L = token.shape
    if bool((token == model.alphabet.padding_idx).any()):
        # ragged rows (user-supplied DataLoader): the general padded path
        lp = model.predict_log_prob(token, pad_output=True)
        return lp[torch.arange(B, device=device), local_pos.to(device)]
    cu_lens = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device=device)
    if getattr(model, 'precision', 'fast') in ('exact', 'half'):      # split-operand mode: fp32 log-probs of every row (the head needs the (hi, lo) pair), then pick
        lp = model.predict_log_prob(token.reshape(-1), (cu_lens, L))
        return lp[(torch.arange(B, dtype=torch.int64) * L + local_pos.to(torch.int64).cpu()).to(device)]
    if graph:
        rep = model.graphed(token.reshape(-1), (cu_lens, L), 'forward_representation', clone=False)
    else:
        rep = model.forward_representation(token.reshape(-1), (cu_lens, L))
    rows = torch.arange(B, dtype=torch.int64) * L + local_pos.to(torch.int64).cpu()
    picked = _hip.gather_rows(rep, rows.to(device))
    return _hip.softmax_rows(model.lm_head(picked), log=True)


def _batches_of(seq, batch_size, max_len, alphabet):
    if isinstance(seq, str):
        return MaskMarginDataset(seq, max_len=max_len, alphabet=alphabet).batches(batch_size)
    if isinstance(seq, MaskMarginDataset):
        return seq.batches(batch_size)
    if isinstance(seq, DataLoader):
        return iter(seq)
    raise ValueError('seq must be str or DataLoader')


def predict_mask_margin(model, seq, batch_size: int = 32, max_len: Optional[int] = None,
                        alphabet=Alphabet3, progress: bool = False):
    """DataFrame indexed by `variant` (`f'{wt}{pos}{aa}'`, 20 rows per residue, in
    residue order then `alphabet.amino_acids` order) with one column `score` =
    log p(aa) - log p(wt) at the masked position (bf16 arithmetic, like the reference's
    esme/variant.py:150-163)."""
    import pandas as pd
    aa_names: List[str] = list(alphabet.amino_acids)
    aa_idx = torch.tensor([alphabet.token_to_idx[a] for a in aa_names], dtype=torch.int64)
    batches = _batches_of(seq, batch_size, max_len, alphabet)
    if progress:
        from tqdm import tqdm
        batches = tqdm(batches)

    # batches of one protein share (batch_size, L): worth a hipGraph once there are a few of them
    n_items = len(seq) if isinstance(seq, (str, MaskMarginDataset)) else 0
    use_graph = n_items >= 4 * batch_size
    names, scores = [], []
    with torch.no_grad():
        for batch in batches:
            full = use_graph and batch['token'].shape[0] == batch_size
            lp = masked_row_log_prob(model, batch['token'], batch['local_pos'], graph=full).cpu()   # (B, V) bf16
            wt_lp = lp[torch.arange(lp.shape[0]), torch.as_tensor(batch['wt_token'])]
            margin = lp - wt_lp.unsqueeze(1)                                                # bf16 - bf16
            scores.append(margin[:, aa_idx].float().numpy())
            for p, wt in zip(torch.as_tensor(batch['pos']).tolist(), batch['wt']):
                names.extend(f'{wt}{p}{aa}' for aa in aa_names)
    score = np.concatenate(scores).reshape(-1).astype(np.float64) if scores else np.zeros(0)
    return pd.DataFrame({'variant': names, 'score': score}).set_index('variant')


def predict_pseudoperplexity(model, seq, batch_size: int = 32, max_len: Optional[int] = None,
                             alphabet=Alphabet3) -> float:
    """exp(mean over residues of -log p(wild type | rest)), every residue masked in turn
    (reference esme/variant.py:168-215, there via torchmetrics.Perplexity)."""
    total, count = 0.0, 0
    with torch.no_grad():
        for batch in _batches_of(seq, batch_size, max_len, alphabet):
            lp = masked_row_log_prob(model, batch['token'], batch['local_pos']).cpu().float()
            wt_lp = lp[torch.arange(lp.shape[0]), torch.as_tensor(batch['wt_token'])]
            total -= float(wt_lp.double().sum())
            count += lp.shape[0]
    return float(np.exp(total / max(count, 1)))
