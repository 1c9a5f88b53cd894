"""Minimal bigWig reader: per-base values of a genomic range.

``bean run --scale-by-acc --acc-bw-path`` reads the accessibility track through
``pyBigWig.open(path).values(chrom, start, end)`` (``bean/preprocessing/utils.py:70-147``);
pyBigWig is not available here, so the part of the format that call needs is decoded directly:
header, chromosome B+ tree, R-tree index, (zlib-compressed) data sections of the three section
types (bedGraph, variableStep, fixedStep).  Zoom levels are never used - ``values`` is exact.
Format: Kent et al., "BigWig and BigBed: enabling browsing of large distributed datasets",
Bioinformatics 2010, and the UCSC ``bbiFile.h`` / ``bwgInternal.h`` layout it documents.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

BIGWIG_MAGIC = 0x888FFC26
CHROM_TREE_MAGIC = 0x78CA8C91
RTREE_MAGIC = 0x2468ACE0


class BigWigFile:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        head = self._f.read(64)
        magic = struct.unpack("<I", head[:4])[0]
        if magic == BIGWIG_MAGIC:
            self._e = "<"
        elif struct.unpack(">I", head[:4])[0] == BIGWIG_MAGIC:
            self._e = ">"
        else:
            raise ValueError(f"{path} is not a bigWig file")
        e = self._e
        (_, self.version, self.n_zoom, self.chrom_tree_offset, self.full_data_offset, self.full_index_offset,
         _fc, _dfc, _asql, self.total_summary_offset, self.uncompress_buf_size, _res) = struct.unpack(
            e + "IHHQQQHHQQIQ", head)
        self._chroms = self._read_chroms()

    # ------------------------------------------------------------------ header pieces
    def _read_chroms(self) -> Dict[str, Tuple[int, int]]:
        e, f = self._e, self._f
        f.seek(self.chrom_tree_offset)
        magic, _block, key_size, _val_size, _count, _res = struct.unpack(e + "IIIIQQ", f.read(32))
        if magic != CHROM_TREE_MAGIC:
            raise ValueError("bad chromosome tree")
        out: Dict[str, Tuple[int, int]] = {}

        def node(offset):
            f.seek(offset)
            is_leaf, _r, count = struct.unpack(e + "BBH", f.read(4))
            items = [f.read(key_size + 8) for _ in range(count)]
            for it in items:
                key = it[:key_size].rstrip(b"\x00").decode()
                if is_leaf:
                    cid, size = struct.unpack(e + "II", it[key_size:])
                    out[key] = (cid, size)
                else:
                    node(struct.unpack(e + "Q", it[key_size:])[0])

        node(self.chrom_tree_offset + 32)
        return out

    def chroms(self) -> Dict[str, int]:
        return {k: v[1] for k, v in self._chroms.items()}

    def total_summary(self) -> Dict[str, float]:
        """validCount, minVal, maxVal, sumData, sumSquares over the whole file."""
        self._f.seek(self.total_summary_offset)
        n, mn, mx, sm, sq = struct.unpack(self._e + "Qdddd", self._f.read(40))
        return {"validCount": n, "minVal": mn, "maxVal": mx, "sumData": sm, "sumSquares": sq}

    # ------------------------------------------------------------------ index
    def _blocks(self, cid: int, start: int, end: int) -> List[Tuple[int, int]]:
        e, f = self._e, self._f
        f.seek(self.full_index_offset)
        magic = struct.unpack(e + "I", f.read(4))[0]
        if magic != RTREE_MAGIC:
            raise ValueError("bad R-tree index")
        f.read(44)  # blockSize, itemCount, bounds, endFileOffset, itemsPerSlot, reserved
        found: List[Tuple[int, int]] = []

        def overlaps(sc, sb, ec, eb):
            return (sc, sb) < (cid, end) and (ec, eb) > (cid, start)

        def node(offset):
            f.seek(offset)
            is_leaf, _r, count = struct.unpack(e + "BBH", f.read(4))
            if is_leaf:
                raw = f.read(32 * count)
                for i in range(count):
                    sc, sb, ec, eb, off, size = struct.unpack_from(e + "IIIIQQ", raw, 32 * i)
                    if overlaps(sc, sb, ec, eb):
                        found.append((off, size))
            else:
                raw = f.read(24 * count)
                kids = []
                for i in range(count):
                    sc, sb, ec, eb, off = struct.unpack_from(e + "IIIIQ", raw, 24 * i)
                    if overlaps(sc, sb, ec, eb):
                        kids.append(off)
                for off in kids:
                    node(off)

        node(self.full_index_offset + 48)
        return found

    # ------------------------------------------------------------------ data
    def intervals(self, chrom: str, start: int, end: int):
        """(start, end, value) runs overlapping [start, end), clipped to it."""
        if chrom not in self._chroms:
            raise KeyError(f"{chrom} is not in {self.path}: {list(self._chroms)}")
        cid, _size = self._chroms[chrom]
        e = self._e
        for off, size in self._blocks(cid, start, end):
            self._f.seek(off)
            raw = self._f.read(size)
            if self.uncompress_buf_size > 0:
                raw = zlib.decompress(raw)
            b_cid, b_start, _b_end, step, span, kind, _r, count = struct.unpack_from(e + "IIIIIBBH", raw, 0)
            if b_cid != cid:
                continue
            pos = 24
            if kind == 1:  # bedGraph
                arr = np.frombuffer(raw, dtype=np.dtype([("s", e + "u4"), ("e", e + "u4"), ("v", e + "f4")]),
                                    count=count, offset=pos)
                s_, e_, v_ = arr["s"].astype(np.int64), arr["e"].astype(np.int64), arr["v"]
            elif kind == 2:  # variableStep
                arr = np.frombuffer(raw, dtype=np.dtype([("s", e + "u4"), ("v", e + "f4")]), count=count, offset=pos)
                s_ = arr["s"].astype(np.int64)
                e_, v_ = s_ + span, arr["v"]
            elif kind == 3:  # fixedStep
                v_ = np.frombuffer(raw, dtype=e + "f4", count=count, offset=pos)
                s_ = b_start + step * np.arange(count, dtype=np.int64)
                e_ = s_ + span
            else:
                raise ValueError(f"unknown bigWig section type {kind}")
            keep = (e_ > start) & (s_ < end)
            for a, b, v in zip(np.maximum(s_[keep], start), np.minimum(e_[keep], end), v_[keep]):
                yield int(a), int(b), float(v)

    def values(self, chrom: str, start: int, end: int) -> np.ndarray:
        """Per-base signal of [start, end) as float32-valued floats, NaN where the file has no data
        (``pyBigWig.bigWigFile.values``)."""
        if chrom not in self._chroms:
            raise RuntimeError("Invalid interval bounds!")
        size = self._chroms[chrom][1]
        if start < 0 or end > size or start >= end:
            raise RuntimeError("Invalid interval bounds!")
        out = np.full(end - start, np.nan, dtype=np.float64)
        for a, b, v in self.intervals(chrom, start, end):
            out[a - start:b - start] = v
        return out

    def close(self):
        self._f.close()


def open_bigwig(path: str) -> BigWigFile:
    return BigWigFile(path)
