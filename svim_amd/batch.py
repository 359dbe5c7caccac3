"""Host batcher: alignment records -> the Structure-of-Arrays record batch of include/svx.h.

This is the host half of rows a1-a3 of the scope table: iteration, query-name
grouping (src/svim/SVIM_COLLECT.py:8-41,108) and SA-tag string parsing (:44-71) stay on the host because
they are string work; everything numeric - filters, CIGAR-derived coordinates, signature emission -
happens on the device from the packed arrays built here.
"""
import logging

import numpy as np

from . import _abi
from ._abi import BATCH_DTYPES, SVX_FLAG_SKIP, ptr
from .records import parse_cigar_string

SVX_FLAG_SA = 0x4000     # segment rows were rebuilt from the SA tag: void them if the primary is hard-clipped


def contig_ranks(references):
    """rank of each contig name in Python str order (the order sorted(key=get_key) sees)."""
    order = sorted(range(len(references)), key=lambda i: references[i])
    rank = np.zeros(max(1, len(references)), dtype=np.int32)
    for r, i in enumerate(order):
        rank[i] = r
    return rank


def pack_bases(seq):
    """str -> 4-bit packed bytes in BAM layout (high nibble first)."""
    codes = _abi.encode_bases(seq)
    if codes.size & 1:
        codes = np.concatenate([codes, np.zeros(1, dtype=np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


class HostBatch(object):
    """numpy SoA of one record batch + the string tables needed to turn results back into objects."""

    def __init__(self):
        self.arrays = {}
        self.n_rec = 0
        self.n_seg = 0
        self.read_names = []
        self.references = []
        self._keep = None

    def struct(self):
        b = _abi.Batch()
        b.on_device = 0
        b.n_rec = self.n_rec
        b.n_seg = self.n_seg
        b.n_contig = len(self.references)
        for k in BATCH_DTYPES:
            setattr(b, k, ptr(self.arrays[k]))
        self._keep = b
        return b

    @property
    def contig_rank(self):
        return self.arrays["contig_rank"]

    def nbytes(self):
        return sum(a.nbytes for a in self.arrays.values())

    def pinned(self):
        """The same batch with its arrays in page-locked memory the library owns (svim_amd._lib.host_empty): svx_collect then uploads them in place, without the
        bounce pass through the library's own buffers that pageable numpy memory takes (include/svx.h).  A batcher that fills the arrays it got from host_empty
        in the first place spares this copy too."""
        from ._lib import host_empty
        out = HostBatch()
        out.n_rec, out.n_seg, out.read_names, out.references = self.n_rec, self.n_seg, self.read_names, self.references
        for k, a in self.arrays.items():
            a = np.ascontiguousarray(a)
            b = host_empty(a.size, a.dtype)
            b[...] = a.reshape(-1)
            out.arrays[k] = b
        return out


def _parse_sa(sa_value, bam):
    """SA tag string -> list of (tid, pos0, reverse, mapq, cigar_tuples); follows
    src/svim/SVIM_COLLECT.py:55-85 (6-field check, pos-1, strand, mapq overflow -> 0)."""
    out = []
    for element in sa_value.split(";"):
        if element == "":
            continue
        fields = element.split(",")
        if len(fields) != 6:
            logging.warning('SA tag does not consist of 6 fields. This could be a sign of invalid characters '
                            '(e.g. commas or semicolons) in a chromosome name of the reference genome.')
            continue
        rname = fields[0]
        pos = int(fields[1])
        strand = fields[2]
        cigar = fields[3]
        mapq = int(fields[4])
        int(fields[5])          # NM: parsed (and may raise) like the reference, otherwise unused
        if mapq < 0 or mapq > 255:
            mapq = 0
        out.append((bam.get_tid(rname), pos - 1, 0 if strand == "+" else 1, mapq, parse_cigar_string(cigar)))
    return out


def build_batch(bam, options, mode="coordinate", records=None):
    """Build the record batch for analyze_alignment_file_coordsorted / _querysorted.

    bam: object with fetch(until_eof=True), get_tid(name), references (our AlignmentFile or pysam's).
    """
    min_mapq = int(getattr(options, "min_mapq", 20))
    if records is None:
        # Ctrl-C while the file is read: stop there and go on with what has been read (src/svim/SVIM_COLLECT.py:126-128,164-166 break their loops the same
        # way; the step after COLLECT runs on the signatures of the reads processed so far)
        recs = []
        try:
            for a in bam.fetch(until_eof=True):
                recs.append(a)
        except KeyboardInterrupt:
            logging.warning('Execution interrupted by user. Stop detection and continue with next step..')
            if mode != "coordinate" and recs:
                # query-name order: the reference only processes COMPLETE read groups (bam_iterator yields a group when the next name appears, :8-41) - the
                # records of the group the interrupt fell into are dropped
                last = recs[-1].query_name
                while recs and recs[-1].query_name == last:
                    recs.pop()
    else:
        recs = list(records)
    n = len(recs)
    hb = HostBatch()
    hb.n_rec = n
    hb.references = list(getattr(bam, "references", []))
    name_id = {}
    flag = np.zeros(n, dtype=np.uint16)
    tid = np.zeros(n, dtype=np.int32)
    pos = np.zeros(n, dtype=np.int32)
    mapq = np.zeros(n, dtype=np.uint8)
    lseq = np.zeros(n, dtype=np.int32)
    read_id = np.zeros(n, dtype=np.int32)
    order = np.zeros(n, dtype=np.uint32)
    seg_order = np.zeros(n, dtype=np.uint32)
    cigar_off = np.zeros(n + 1, dtype=np.uint64)
    seq_off = np.zeros(n + 1, dtype=np.uint64)
    seg_off = np.zeros(n + 1, dtype=np.uint32)
    cig = []
    seqs = []
    segs = [[] for _ in range(n)]      # rows: (tid, pos, rev, mapq, lseq, cigar)

    for i, a in enumerate(recs):
        nm = a.query_name
        rid = name_id.get(nm)
        if rid is None:
            rid = name_id[nm] = len(hb.read_names)
            hb.read_names.append(nm)
        read_id[i] = rid
        flag[i] = a.flag & 0x0fff
        tid[i] = a.reference_id
        pos[i] = a.reference_start
        mapq[i] = a.mapping_quality
        ct = a.cigartuples or []
        cig.extend((l << 4) | op for op, l in ct)
        cigar_off[i + 1] = len(cig)
        s = a.query_sequence
        if s:
            pb = pack_bases(s)
            seqs.append(pb)
            lseq[i] = len(s)
            seq_off[i + 1] = seq_off[i] + np.uint64(pb.size)
        else:
            seq_off[i + 1] = seq_off[i]

    if mode == "coordinate":
        for i, a in enumerate(recs):
            order[i] = 2 * i
            seg_order[i] = 2 * i + 1
            f = int(flag[i])
            if f & (4 | 256 | 2048) or mapq[i] < min_mapq:
                continue
            try:
                sa = a.get_tag("SA")
            except KeyError:
                continue
            flag[i] |= SVX_FLAG_SA
            L = int(lseq[i])
            segs[i] = [(t, p, r, q, L, c) for (t, p, r, q, c) in _parse_sa(sa, bam)]
    elif mode == "queryname":
        # bam_iterator grouping (src/svim/SVIM_COLLECT.py:8-41): consecutive records sharing query_name
        slot = 0
        i = 0
        while i < n:
            j = i
            while j < n and recs[j].query_name == recs[i].query_name:
                j += 1
            prim, sup = [], []
            for k in range(i, j):
                f = int(flag[k])
                if f & 256:
                    pass
                elif f & 2048:
                    sup.append(k)
                else:
                    prim.append(k)
            ok = len(prim) == 1 and not (flag[prim[0]] & 4) and mapq[prim[0]] >= min_mapq
            for k in range(i, j):
                flag[k] |= SVX_FLAG_SKIP
            if ok:
                p = prim[0]
                good = [k for k in sup if not (flag[k] & 4) and mapq[k] >= min_mapq]
                flag[p] &= ~np.uint16(SVX_FLAG_SKIP)
                order[p] = slot
                for q, k in enumerate(good):
                    flag[k] &= ~np.uint16(SVX_FLAG_SKIP)
                    order[k] = slot + 1 + q
                    segs[p].append((int(tid[k]), int(pos[k]), 1 if flag[k] & 16 else 0, int(mapq[k]), int(lseq[k]),
                                    recs[k].cigartuples or []))
                seg_order[p] = slot + 1 + len(good)
                slot += len(good) + 2
            i = j
    else:
        raise ValueError("mode must be 'coordinate' or 'queryname'")

    n_seg = sum(len(x) for x in segs)
    hb.n_seg = n_seg
    seg_tid = np.zeros(max(1, n_seg), dtype=np.int32)
    seg_pos = np.zeros(max(1, n_seg), dtype=np.int32)
    seg_rev = np.zeros(max(1, n_seg), dtype=np.uint8)
    seg_mapq = np.zeros(max(1, n_seg), dtype=np.uint8)
    seg_lseq = np.zeros(max(1, n_seg), dtype=np.int32)
    seg_cigar_off = np.zeros(n_seg + 1, dtype=np.uint64)
    scig = []
    k = 0
    for i in range(n):
        seg_off[i] = k
        for (t, p, r, q, L, c) in segs[i]:
            seg_tid[k], seg_pos[k], seg_rev[k], seg_mapq[k], seg_lseq[k] = t, p, r, q, L
            scig.extend((l << 4) | op for op, l in c)
            seg_cigar_off[k + 1] = len(scig)
            k += 1
    seg_off[n] = k
    A = hb.arrays
    A["flag"], A["tid"], A["pos"], A["mapq"], A["lseq"], A["read_id"] = flag, tid, pos, mapq, lseq, read_id
    A["order"], A["seg_order"], A["cigar_off"], A["seq_off"], A["seg_off"] = order, seg_order, cigar_off, seq_off, seg_off
    A["cigar"] = np.array(cig if cig else [0], dtype=np.uint32)
    A["seq"] = np.concatenate(seqs) if seqs else np.zeros(1, dtype=np.uint8)
    A["seg_tid"], A["seg_pos"], A["seg_rev"], A["seg_mapq"], A["seg_lseq"] = seg_tid, seg_pos, seg_rev, seg_mapq, seg_lseq
    A["seg_cigar_off"] = seg_cigar_off
    A["seg_cigar"] = np.array(scig if scig else [0], dtype=np.uint32)
    A["contig_rank"] = contig_ranks(hb.references)
    return hb
