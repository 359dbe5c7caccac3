"""Signature / cluster objects <-> Structure-of-Arrays tables (include/svx.h)."""
import math

import numpy as np

from . import _abi
from ._abi import (SVX_BND, SVX_DEL, SVX_DUP_INT, SVX_DUP_TAN, SVX_INS, SVX_INV, TYPE_CODE, TYPE_NAMES,
                   INV_DIRECTIONS, INV_DIR_CODE, SRC_NAMES, SigTable)
from .signatures import (SignatureDeletion, SignatureInsertion, SignatureInversion, SignatureInsertionFrom,
                         SignatureDuplicationTandem, SignatureTranslocation, SignatureClusterUniLocal,
                         SignatureClusterBiLocal)

_I32_MIN, _I32_MAX = -(1 << 31), (1 << 31) - 1


class Interner(object):
    """name <-> dense id"""

    def __init__(self, names=()):
        self.names = list(names)
        self.ids = {n: i for i, n in enumerate(self.names)}

    def __call__(self, name):
        i = self.ids.get(name)
        if i is None:
            i = self.ids[name] = len(self.names)
            self.names.append(name)
        return i


def _coord(x):
    if isinstance(x, bool) or not isinstance(x, (int, np.integer)):
        raise TypeError("signature coordinates must be integers on the GPU path (got %r)" % (x,))
    if not (_I32_MIN <= x <= _I32_MAX):
        raise OverflowError("signature coordinate %r outside int32" % (x,))
    return int(x)


def sigtable_from_objects(sigs, contigs=None, reads=None):
    """list of Signature objects (ours or the reference's: attribute access only) -> SigTable.
    Returns (table, contig Interner, read Interner)."""
    contigs = contigs if contigs is not None else Interner()
    reads = reads if reads is not None else Interner()
    n = len(sigs)
    seqs = []
    t = SigTable(n)
    off = 0
    for i, s in enumerate(sigs):
        code = TYPE_CODE[s.type]
        t.type[i] = code
        t.key[i] = i
        sig_src = s.signature
        t.src[i] = 0 if sig_src == "cigar" else 1
        t.read_id[i] = reads(s.read)
        t.contig2[i] = -1
        if code == SVX_BND:
            t.contig[i] = contigs(s.contig1)
            t.start[i] = _coord(s.pos1)
            t.end[i] = _coord(s.pos1 + 1)
            t.contig2[i] = contigs(s.contig2)
            t.pos2[i] = _coord(s.pos2)
            t.aux[i] = (1 if s.direction1 == "rev" else 0) | (2 if s.direction2 == "rev" else 0)
        elif code == SVX_DUP_INT:
            t.contig[i] = contigs(s.contig1)
            t.start[i], t.end[i] = _coord(s.start), _coord(s.end)
            t.contig2[i] = contigs(s.contig2)
            t.pos2[i] = _coord(s.pos)
        else:
            t.contig[i] = contigs(s.contig)
            t.start[i], t.end[i] = _coord(s.start), _coord(s.end)
            if code == SVX_INV:
                t.aux[i] = INV_DIR_CODE[s.direction]
            elif code == SVX_DUP_TAN:
                t.pos2[i] = _coord(s.copies)
                t.aux[i] = 1 if s.fully_covered else 0
            elif code == SVX_INS:
                c = _abi.encode_bases(s.sequence or "")
                seqs.append(c)
                off += c.size
        t.seq_off[i + 1] = off
    t.seq = np.concatenate(seqs) if seqs else np.zeros(1, dtype=np.uint8)
    if t.seq.size == 0:
        t.seq = np.zeros(1, dtype=np.uint8)
    return t, contigs, reads


def objects_from_sigtable(t, references, read_names):
    out = []
    dirs = ("fwd", "rev")
    for i in range(t.n):
        code = int(t.type[i])
        src = SRC_NAMES[t.src[i]]
        read = read_names[t.read_id[i]]
        c = references[t.contig[i]]
        s, e = int(t.start[i]), int(t.end[i])
        if code == SVX_DEL:
            o = SignatureDeletion(c, s, e, src, read)
        elif code == SVX_INS:
            o = SignatureInsertion(c, s, e, src, read, t.sequence(i))
        elif code == SVX_INV:
            o = SignatureInversion(c, s, e, src, read, INV_DIRECTIONS[t.aux[i]])
        elif code == SVX_DUP_TAN:
            o = SignatureDuplicationTandem(c, s, e, int(t.pos2[i]), bool(t.aux[i] & 1), src, read)
        elif code == SVX_DUP_INT:
            o = SignatureInsertionFrom(c, s, e, references[t.contig2[i]], int(t.pos2[i]), src, read)
        else:
            o = SignatureTranslocation(c, s, dirs[t.aux[i] & 1], references[t.contig2[i]], int(t.pos2[i]),
                                       dirs[(t.aux[i] >> 1) & 1], src, read)
        out.append(o)
    return out


def _none_if_nan(x):
    return None if math.isnan(x) else float(x)


def cluster_objects(ct, sig_objects, references):
    """ClusterTable -> the 6-tuple cluster_sv_signatures returns (src/svim/SVIM_CLUSTER.py:26):
    (DEL, INS, INV, DUP_TAN, DUP_INT, BND)."""
    dirs = ("fwd", "rev")
    by_type = [[] for _ in range(6)]
    for k in range(ct.n):
        code = int(ct.type[k])
        members = [sig_objects[j] for j in ct.members[ct.member_off[k]:ct.member_off[k + 1]]]
        sp, po = _none_if_nan(ct.std_span[k]), _none_if_nan(ct.std_pos[k])
        name = TYPE_NAMES[code]
        if code <= SVX_INV:
            o = SignatureClusterUniLocal(references[ct.contig[k]], int(ct.start[k]), int(ct.end[k]),
                                         float(ct.score[k]), int(ct.size[k]), members, name, sp, po)
        else:
            o = SignatureClusterBiLocal(references[ct.contig[k]], int(ct.start[k]), int(ct.end[k]),
                                        references[ct.contig2[k]], int(ct.start2[k]), int(ct.end2[k]),
                                        float(ct.score[k]), int(ct.size[k]), members, name, sp, po)
            if code == SVX_BND:
                o.direction1 = dirs[ct.aux[k] & 1]
                o.direction2 = dirs[(ct.aux[k] >> 1) & 1]
        by_type[code].append(o)
    return (by_type[SVX_DEL], by_type[SVX_INS], by_type[SVX_INV], by_type[SVX_DUP_TAN], by_type[SVX_DUP_INT],
            by_type[SVX_BND])


def genome_arrays(path_or_dict, references):
    """FASTA (plain or .gz) or {name: sequence} -> (off int64[n+1], codes uint8) indexed like `references`.
    Contigs absent from the FASTA get length 0 (fetch() on them yields '')."""
    if isinstance(path_or_dict, dict):
        seqs = path_or_dict
    else:
        import gzip
        seqs = {}
        name = None
        chunks = None
        opener = gzip.open if str(path_or_dict).endswith(".gz") else open
        with opener(path_or_dict, "rb") as fh:
            for line in fh:
                if line.startswith(b">"):
                    if name is not None:
                        seqs[name] = b"".join(chunks)
                    name = line[1:].split()[0].decode("ascii")
                    chunks = []
                elif name is not None:
                    chunks.append(line.strip())
            if name is not None:
                seqs[name] = b"".join(chunks)
    off = np.zeros(len(references) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(references):
        s = seqs.get(r, b"")
        c = _abi.encode_bases(s)
        parts.append(c)
        off[i + 1] = off[i] + c.size
    codes = np.concatenate(parts) if parts else np.zeros(1, dtype=np.uint8)
    if codes.size == 0:
        codes = np.zeros(1, dtype=np.uint8)
    return off, codes
