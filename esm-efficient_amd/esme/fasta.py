"""Random-access FASTA reader over a samtools `.fai` index.

Mirrors `esme/fasta.py:5-100` of the reference (`read_fai`, `Fasta`: `fasta[i]`,
`fasta['id']`, `len(fasta)`, `max_len` filter, `k_sample`), without the polars
dependency: the index is five tab-separated columns and is read with the csv module
into the same list-of-dicts the reference ends up with (`Fasta.fai`).
"""
from __future__ import annotations

import csv
import random
from pathlib import Path
from typing import Dict, List, Optional

FAI_COLUMNS = ('id', 'length', 'offset', 'line_bases', 'line_width')


def read_fai(fai_path) -> List[Dict[str, object]]:
    """Rows of a `.fai` file as dicts with keys id, length, offset, line_bases, line_width."""
    rows = []
    with open(fai_path, newline='') as f:
        for rec in csv.reader(f, delimiter='\t'):
            if not rec:
                continue
            rows.append({'id': rec[0], **{k: int(v) for k, v in zip(FAI_COLUMNS[1:], rec[1:5])}})
    return rows


class Fasta:
    def __init__(self, fasta, fai=None, max_len: Optional[int] = None, k_sample: Optional[int] = None,
                 random_state=None):
        self.fasta = str(fasta)
        if not Path(self.fasta).exists():
            raise FileNotFoundError(f'File not found: {fasta}')
        if fai is None:
            fai = self.fasta + '.fai'
        try:
            rows = read_fai(fai)
        except FileNotFoundError as e:
            raise FileNotFoundError(
                f'Please index the fasta file with samtools faidx: `samtools faidx {fasta}`') from e
        if max_len is not None:
            rows = [r for r in rows if r['length'] <= max_len]
        if k_sample is not None:
            rows = random.Random(random_state).sample(rows, k_sample)
        self.fai = rows
        self.proteins = {row['id']: i for i, row in enumerate(rows)}

    def __getitem__(self, idx):
        if isinstance(idx, int):
            return self.read_seq(idx)
        if isinstance(idx, str):
            return self.read_seq(self.proteins[idx])
        raise ValueError(f'Invalid index: {idx}')

    def read_seq(self, idx: int) -> str:
        """Seek to the record's byte offset and join its lines (the index gives the exact
        span: `length` residues in lines of `line_bases` residues / `line_width` bytes)."""
        row = self.fai[idx]
        length, bases, width = row['length'], row['line_bases'], row['line_width']
        nbytes = length + (length // bases) * (width - bases) if bases else length
        with open(self.fasta, 'rb') as f:
            f.seek(row['offset'])
            raw = f.read(nbytes)
        seq = raw.decode('ascii').replace('\n', '').replace('\r', '')[:length]
        assert len(seq) == length
        return seq

    def __len__(self):
        return len(self.fai)


def index_fasta(fasta, fai=None) -> str:
    """Write the samtools-style `.fai` index of a FASTA file (id, length, byte offset of the first residue,
    residues per line, bytes per line) and return its path.  The reference expects `samtools faidx` to have
    been run; this is the same five columns for files whose records use one fixed line width (the last line
    of a record may be shorter), which is what samtools requires too."""
    fasta = str(fasta)
    fai = fai or fasta + '.fai'
    rows = []
    with open(fasta, 'rb') as f:
        name, length, offset, bases, width, short_seen = None, 0, 0, 0, 0, False
        pos = 0
        for line in f:
            if line.startswith(b'>'):
                if name is not None:
                    rows.append((name, length, offset, bases, width))
                name = line[1:].split()[0].decode('ascii') if len(line) > 1 else ''
                length, bases, width, short_seen = 0, 0, 0, False
                offset = pos + len(line)
            elif name is not None:
                n = len(line.rstrip(b'\r\n'))
                if n:
                    if short_seen or (bases and n > bases):
                        raise ValueError(f'{fasta}: record {name!r} has lines of different lengths (not indexable)')
                    if not bases:
                        bases, width = n, len(line)
                    elif n < bases:
                        short_seen = True
                    length += n
            pos += len(line)
        if name is not None:
            rows.append((name, length, offset, bases, width))
    with open(fai, 'w') as out:
        for r in rows:
            out.write('\t'.join(str(v) for v in r) + '\n')
    return fai
