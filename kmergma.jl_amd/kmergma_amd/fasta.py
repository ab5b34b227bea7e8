"""Minimal FASTA reader/record for the host side of the scan.

Plays the role FASTX.FASTA.Reader / FASTA.Record play for the reference's engines
(src/GenomeMiner.jl:31-35, src/Consts.jl:37-39): the identifier is the header up to the first
whitespace, the description is the whole header line, the sequence is the raw residue bytes
(either case; validation against the reference's A/C/G/T/N code happens in the scan library,
mirroring the KeyError of src/Consts.jl:22-28).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List


@dataclass
class Record:
    """A FASTA record: `description` is the full header (without '>'), `sequence` is bytes."""

    description: str
    sequence: bytes

    @property
    def identifier(self) -> str:
        parts = self.description.split(None, 1)
        return parts[0] if parts else ""

    def __len__(self) -> int:
        return len(self.sequence)

    def __eq__(self, other) -> bool:  # FASTA.Record equality = header + sequence (case kept)
        return (isinstance(other, Record) and self.description == other.description
                and self.sequence.upper() == other.sequence.upper())


def read_fasta(path: str) -> List[Record]:
    """Parse a FASTA file: multi-line records, blank lines and CR/LF tolerated."""
    records: List[Record] = []
    header = None
    chunks: List[bytes] = []
    with open(path, "rb") as fh:
        for raw in fh:
            line = raw.rstrip(b"\r\n")
            if not line:
                continue
            if line[:1] == b">":
                if header is not None:
                    records.append(Record(header, b"".join(chunks)))
                header = line[1:].decode("utf-8", "replace").strip()
                chunks = []
            elif line[:1] == b";":
                continue
            else:
                if header is None:
                    raise ValueError(f"{path}: sequence data before the first '>' header")
                chunks.append(line.replace(b" ", b""))
    if header is not None:
        records.append(Record(header, b"".join(chunks)))
    return records


def write_fasta(records: Iterable[Record], path: str, width: int = 95, append: bool = True) -> None:
    """src/API.jl:234-241 write_results: APPENDS records, sequence wrapped at `width`."""
    with open(path, "ab" if append else "wb") as fh:
        for rec in records:
            fh.write(b">" + rec.description.encode() + b"\n")
            seq = rec.sequence.upper()
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + b"\n")
