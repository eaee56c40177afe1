"""Token-budget batching of a FASTA file into packed forward inputs.

The caller side of the hot path: `FastaTokenDataset[i]` is exactly the
`(tokens, (cu_lens, max_len))` pair `ESM2.forward` consumes.  Names, arguments and the
greedy packing rule follow `esme/data.py:12-60,63-162` of the reference; the Lightning
data modules and the masking datasets for training (data.py:165-520) are outside the
inference hot path.
"""
from __future__ import annotations

from typing import List, Sequence

from torch.utils.data import DataLoader, Dataset

from esme.alphabet import Alphabet3, pad_tokens, tokenize, tokenize_unpad
from esme.fasta import Fasta


class TokenSizeBatchSampler:
    """Greedy batches of sequence indices whose token count (length + 2 for cls/eos) stays
    within `token_per_batch`.  Walks the (optionally shuffled) order once; a batch is closed
    when the next sequence would overflow it.  Like the reference (data.py:33-54) a sequence
    longer than the whole budget still gets a batch of its own, and if the very first one
    overflows, the first batch emitted is empty."""

    def __init__(self, token_sizes: Sequence[int], token_per_batch: int, drop_last: bool = False,
                 shuffle: bool = True, random_state=None):
        self.token_sizes = token_sizes
        self.token_per_batch = token_per_batch
        self.drop_last, self.shuffle, self.random_state = drop_last, shuffle, random_state
        self._batches = list(self.batches())

    def batches(self):
        order: List[int] = list(range(len(self.token_sizes)))
        if self.shuffle:
            import sklearn.utils            # the reference's shuffle, so seeds give the same order
            order = sklearn.utils.shuffle(order, random_state=self.random_state)
        batch, used = [], 0
        for idx in order:
            need = self.token_sizes[idx] + 2
            if used + need > self.token_per_batch:
                yield batch
                batch, used = [idx], need
            else:
                batch.append(idx)
                used += need
        if batch and not self.drop_last:
            yield batch

    def __iter__(self):
        return iter(self._batches)

    def __getitem__(self, idx):
        return self._batches[idx]

    def __len__(self):
        return len(self._batches)


class BaseFastaDataset(Dataset):
    def __init__(self, fasta, fai=None, k_sample=None, max_len=None, alphabet=Alphabet3):
        self.max_len = max_len or float('inf')
        self.alphabet = alphabet
        self.fasta = Fasta(fasta, fai=fai, max_len=max_len, k_sample=k_sample)

    def read_seq(self, idx):
        return self.fasta[idx]


class FastaDataset(BaseFastaDataset):
    """One tokenised sequence per item; `collate_fn` right-pads a batch (data.py:82-112)."""

    def __len__(self):
        return len(self.fasta)

    def __getitem__(self, idx):
        return tokenize(self.read_seq(idx), alphabet=self.alphabet)

    def collate_fn(self, batch):
        return pad_tokens(batch, alphabet=self.alphabet)

    def to_dataloader(self, batch_size, shuffle=False, num_workers=0, **kwargs):
        return DataLoader(self, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                          collate_fn=self.collate_fn, **kwargs)


class FastaTokenDataset(BaseFastaDataset):
    """Item i = the i-th token-budget batch, already packed: `(tokens, (cu_lens, max_len))`
    (data.py:115-162)."""

    def __init__(self, fasta, fai=None, token_per_batch=50_000, k_sample=None, max_len=None,
                 drop_last=False, shuffle=True, random_state=None, alphabet=Alphabet3):
        super().__init__(fasta, fai=fai, k_sample=k_sample, max_len=max_len, alphabet=alphabet)
        self.token_per_batch = token_per_batch
        lengths = [row['length'] for row in self.fasta.fai]
        self.sampler = list(TokenSizeBatchSampler(lengths, token_per_batch, drop_last=drop_last,
                                                  shuffle=shuffle, random_state=random_state))

    def __len__(self):
        return len(self.sampler)

    def __getitem__(self, idx):
        token, _, cu_lens, max_len = tokenize_unpad([self.read_seq(i) for i in self.sampler[idx]],
                                                    alphabet=self.alphabet)
        return token, (cu_lens, max_len)

    def to_dataloader(self, num_workers=0, **kwargs):
        """One packed batch per item (batch_size=None).  On one GPU keep `num_workers=0`: an item takes ~20 ms of one core against 60 - 70 ms of GPU time
        for it, while worker processes FORKED from a process that already holds a GPU context stall its queues for 1 - 3 s when they start
        (profiles/r05_e2e_fork_stall.txt); pass `multiprocessing_context='forkserver'` if workers are needed."""
        return DataLoader(self, num_workers=num_workers, batch_size=None, **kwargs)
