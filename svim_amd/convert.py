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


def _coord_list(values, what):
    """list of Python / numpy integers -> int32 array; the same errors as _coord for anything else"""
    for x in values:
        if type(x) is not int:
            _coord(x)                                 # raises for non-integers, accepts numpy integers
    a = np.asarray(values, dtype=np.int64) if values else np.zeros(0, dtype=np.int64)
    if a.size and (a.min() < _I32_MIN or a.max() > _I32_MAX):
        bad = a[(a < _I32_MIN) | (a > _I32_MAX)][0]
        raise OverflowError("signature coordinate %r outside int32" % (int(bad),))
    return a.astype(np.int32)


def sigtable_from_objects(sigs, contigs=None, reads=None):
    """list of Signature objects (ours or the reference's: attribute access only) -> SigTable.
    Returns (table, contig Interner, read Interner).  Column lists are filled in one pass over the objects and turned into arrays
    once (a numpy scalar store per field costs ~10x a list append)."""
    contigs = contigs if contigs is not None else Interner()
    reads = reads if reads is not None else Interner()
    n = len(sigs)
    typ, src, rid, c1, st, en, c2, p2, aux, seq_parts, seq_len = [], [], [], [], [], [], [], [], [], [], []
    for s in sigs:
        code = TYPE_CODE[s.type]
        typ.append(code)
        src.append(0 if s.signature == "cigar" else 1)
        rid.append(reads(s.read))
        ln = 0
        if code == SVX_BND:
            c1.append(contigs(s.contig1)); st.append(s.pos1); en.append(s.pos1 + 1)
            c2.append(contigs(s.contig2)); p2.append(s.pos2)
            aux.append((1 if s.direction1 == "rev" else 0) | (2 if s.direction2 == "rev" else 0))
        elif code == SVX_DUP_INT:
            c1.append(contigs(s.contig1)); st.append(s.start); en.append(s.end)
            c2.append(contigs(s.contig2)); p2.append(s.pos); aux.append(0)
        else:
            c1.append(contigs(s.contig)); st.append(s.start); en.append(s.end); c2.append(-1)
            if code == SVX_INV:
                p2.append(0); aux.append(INV_DIR_CODE[s.direction])
            elif code == SVX_DUP_TAN:
                p2.append(s.copies); aux.append(1 if s.fully_covered else 0)
            else:
                p2.append(0); aux.append(0)
                if code == SVX_INS:
                    q = s.sequence or ""
                    seq_parts.append(q)
                    ln = len(q)
        seq_len.append(ln)
    t = SigTable(n)
    t.type[:] = np.asarray(typ, dtype=np.uint8) if n else 0
    t.key[:] = np.arange(n, dtype=np.uint64)
    t.src[:] = np.asarray(src, dtype=np.uint8) if n else 0
    t.read_id[:] = np.asarray(rid, dtype=np.int32) if n else 0
    t.contig[:] = np.asarray(c1, dtype=np.int32) if n else 0
    t.contig2[:] = np.asarray(c2, dtype=np.int32) if n else 0
    t.aux[:] = np.asarray(aux, dtype=np.uint8) if n else 0
    t.start[:] = _coord_list(st, "start")
    t.end[:] = _coord_list(en, "end")
    t.pos2[:] = _coord_list(p2, "pos2")
    t.seq_off[1:] = np.cumsum(np.asarray(seq_len, dtype=np.int64)) if n else 0
    codes = _abi.encode_bases("".join(seq_parts)) if seq_parts else np.zeros(0, dtype=np.uint8)
    t.seq = codes if codes.size else np.zeros(1, dtype=np.uint8)
    return t, contigs, reads


def objects_from_sigtable(t, references, read_names):
    """SigTable -> list of Signature objects.  The columns are converted to Python lists and the inserted sequences decoded in one
    go; the loop then only calls the constructors."""
    n = t.n
    typ, srcs, aux = t.type[:n].tolist(), t.src[:n].tolist(), t.aux[:n].tolist()
    rid, c1, c2 = t.read_id[:n].tolist(), t.contig[:n].tolist(), t.contig2[:n].tolist()
    st, en, p2 = t.start[:n].tolist(), t.end[:n].tolist(), t.pos2[:n].tolist()
    off = t.seq_off[:n + 1].tolist()
    all_seq = _abi.decode_bases(t.seq[:off[n]]) if n and off[n] else ""
    dirs = ("fwd", "rev")
    out = []
    add = out.append
    for i in range(n):
        code = typ[i]
        src = SRC_NAMES[srcs[i]]
        read = read_names[rid[i]]
        c = references[c1[i]]
        if code == SVX_DEL:
            add(SignatureDeletion(c, st[i], en[i], src, read))
        elif code == SVX_INS:
            add(SignatureInsertion(c, st[i], en[i], src, read, all_seq[off[i]:off[i + 1]]))
        elif code == SVX_INV:
            add(SignatureInversion(c, st[i], en[i], src, read, INV_DIRECTIONS[aux[i]]))
        elif code == SVX_DUP_TAN:
            add(SignatureDuplicationTandem(c, st[i], en[i], p2[i], bool(aux[i] & 1), src, read))
        elif code == SVX_DUP_INT:
            add(SignatureInsertionFrom(c, st[i], en[i], references[c2[i]], p2[i], src, read))
        else:
            add(SignatureTranslocation(c, st[i], dirs[aux[i] & 1], references[c2[i]], p2[i], dirs[(aux[i] >> 1) & 1], src, read))
    return out


def _none_if_nan(x):
    return None if math.isnan(x) else float(x)


def cluster_objects(ct, sig_objects, references):
    """ClusterTable -> the 6-tuple cluster_sv_signatures returns (src/svim/SVIM_CLUSTER.py:26):
    (DEL, INS, INV, DUP_TAN, DUP_INT, BND)."""
    dirs = ("fwd", "rev")
    by_type = [[] for _ in range(6)]
    n = ct.n
    typ, aux, size = ct.type[:n].tolist(), ct.aux[:n].tolist(), ct.size[:n].tolist()
    c1, st, en = ct.contig[:n].tolist(), ct.start[:n].tolist(), ct.end[:n].tolist()
    c2, st2, en2 = ct.contig2[:n].tolist(), ct.start2[:n].tolist(), ct.end2[:n].tolist()
    score, sspan, spos = ct.score[:n].tolist(), ct.std_span[:n].tolist(), ct.std_pos[:n].tolist()
    moff = ct.member_off[:n + 1].tolist()
    mem = ct.members[:moff[n] if n else 0].tolist()
    for k in range(n):
        code = typ[k]
        members = [sig_objects[j] for j in mem[moff[k]:moff[k + 1]]]
        sp, po = _none_if_nan(sspan[k]), _none_if_nan(spos[k])
        name = TYPE_NAMES[code]
        if code <= SVX_INV:
            o = SignatureClusterUniLocal(references[c1[k]], st[k], en[k], score[k], size[k], members, name, sp, po)
        else:
            o = SignatureClusterBiLocal(references[c1[k]], st[k], en[k], references[c2[k]], st2[k], en2[k], score[k], size[k], members,
                                        name, sp, po)
            if code == SVX_BND:
                o.direction1 = dirs[aux[k] & 1]
                o.direction2 = dirs[(aux[k] >> 1) & 1]
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
