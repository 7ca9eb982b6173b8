"""Minimal FASTA reader/writer (biopython is not a dependency of this engine)."""
from typing import Iterator, List, Tuple


def read_fasta(path: str) -> Iterator[Tuple[str, str]]:
    """Yields (name, sequence) for every record; the name is the header up to the first whitespace."""
    name, chunks = None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                fields = line[1:].split()
                name, chunks = (fields[0] if fields else ""), []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def length_buckets(seqs: List[str], batch_size: int) -> List[List[int]]:
    """Indices grouped into batches of similar length (longest first): the reference pads every batch to its
    longest member and computes the pads [REF evo/scoring.py:19-31]; bucketing keeps that waste small."""
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
