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
    """SigTable -> list of Signature objects.  Vectorised per type: the columns of one type become Python lists (names looked up
    through object arrays), one constructor call per row via map(), and the objects are scattered back into list order."""
    n = t.n
    out = [None] * n
    if n == 0:
        return out
    typ = t.type[:n]
    refs = np.asarray(list(references) + [None], dtype=object)            # index -1 (no second contig) -> None
    names = np.asarray(read_names, dtype=object) if not isinstance(read_names, np.ndarray) else read_names
    srcs = np.asarray(SRC_NAMES, dtype=object)
    dirs = np.asarray(("fwd", "rev"), dtype=object)
    inv_dirs = np.asarray(INV_DIRECTIONS, dtype=object)
    for code in range(6):
        idx = np.nonzero(typ == code)[0]
        if idx.size == 0:
            continue
        c1 = refs[t.contig[idx]].tolist()
        st, en = t.start[idx].tolist(), t.end[idx].tolist()
        src = srcs[t.src[idx]].tolist()
        rd = names[t.read_id[idx]].tolist()
        if code == SVX_DEL:
            objs = map(SignatureDeletion, c1, st, en, src, rd)
        elif code == SVX_INS:
            off = t.seq_off
            lo, hi = off[idx].tolist(), off[idx + 1].tolist()
            all_seq = _abi.decode_bases(t.seq[:int(off[n])]) if int(off[n]) else ""
            objs = map(SignatureInsertion, c1, st, en, src, rd, [all_seq[a:b] for a, b in zip(lo, hi)])
        elif code == SVX_INV:
            objs = map(SignatureInversion, c1, st, en, src, rd, inv_dirs[t.aux[idx]].tolist())
        elif code == SVX_DUP_TAN:
            objs = map(SignatureDuplicationTandem, c1, st, en, t.pos2[idx].tolist(), (t.aux[idx] & 1).astype(bool).tolist(), src, rd)
        elif code == SVX_DUP_INT:
            objs = map(SignatureInsertionFrom, c1, st, en, refs[t.contig2[idx]].tolist(), t.pos2[idx].tolist(), src, rd)
        else:
            aux = t.aux[idx]
            objs = map(SignatureTranslocation, c1, st, dirs[aux & 1].tolist(), refs[t.contig2[idx]].tolist(), t.pos2[idx].tolist(),
                       dirs[(aux >> 1) & 1].tolist(), src, rd)
        for i, o in zip(idx.tolist(), objs):
            out[i] = o
    return out


def object_from_row(t, i, references, read_names):
    """One Signature object from row i of a SigTable (SignatureList's single-element access)."""
    code = int(t.type[i])
    src = SRC_NAMES[int(t.src[i])]
    read = read_names[int(t.read_id[i])]
    c = references[int(t.contig[i])]
    st, en, aux = int(t.start[i]), int(t.end[i]), int(t.aux[i])
    dirs = ("fwd", "rev")
    if code == SVX_DEL:
        return SignatureDeletion(c, st, en, src, read)
    if code == SVX_INS:
        return SignatureInsertion(c, st, en, src, read, _abi.decode_bases(t.seq[int(t.seq_off[i]):int(t.seq_off[i + 1])]))
    if code == SVX_INV:
        return SignatureInversion(c, st, en, src, read, INV_DIRECTIONS[aux])
    if code == SVX_DUP_TAN:
        return SignatureDuplicationTandem(c, st, en, int(t.pos2[i]), bool(aux & 1), src, read)
    if code == SVX_DUP_INT:
        return SignatureInsertionFrom(c, st, en, references[int(t.contig2[i])], int(t.pos2[i]), src, read)
    return SignatureTranslocation(c, st, dirs[aux & 1], references[int(t.contig2[i])], int(t.pos2[i]), dirs[(aux >> 1) & 1], src, read)


def _none_if_nan(x):
    return None if math.isnan(x) else float(x)


def cluster_objects_range(ct, lo, hi, sig_objects, references):
    """Rows [lo, hi) of a ClusterTable -> SignatureCluster objects.  `members` stays an index list until it is read
    (signatures.py: the cluster classes resolve it against sig_objects on first access)."""
    dirs = ("fwd", "rev")
    out = []
    typ, aux, size = ct.type[lo:hi].tolist(), ct.aux[lo:hi].tolist(), ct.size[lo:hi].tolist()
    c1, st, en = ct.contig[lo:hi].tolist(), ct.start[lo:hi].tolist(), ct.end[lo:hi].tolist()
    c2, st2, en2 = ct.contig2[lo:hi].tolist(), ct.start2[lo:hi].tolist(), ct.end2[lo:hi].tolist()
    score, sspan, spos = ct.score[lo:hi].tolist(), ct.std_span[lo:hi].tolist(), ct.std_pos[lo:hi].tolist()
    moff = ct.member_off[lo:hi + 1].tolist()
    for k in range(hi - lo):
        code = typ[k]
        members = (sig_objects, ct.members[moff[k]:moff[k + 1]])            # resolved lazily
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
        out.append(o)
    return out


def cluster_objects(ct, sig_objects, references):
    """ClusterTable -> the 6-tuple cluster_sv_signatures returns (src/svim/SVIM_CLUSTER.py:26):
    (DEL, INS, INV, DUP_TAN, DUP_INT, BND), each a lazy ClusterList view (the table is grouped by type in SVX_* order)."""
    from .lazy import ClusterList
    bounds = [0]
    for c in ct.type_count:
        bounds.append(bounds[-1] + int(c))
    views = [ClusterList(ct, bounds[k], bounds[k + 1], sig_objects, references) for k in range(6)]
    return (views[SVX_DEL], views[SVX_INS], views[SVX_INV], views[SVX_DUP_TAN], views[SVX_DUP_INT], views[SVX_BND])


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
