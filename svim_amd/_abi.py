"""ctypes mirror of include/svx.h (struct layouts + helpers to wrap numpy arrays)."""
import ctypes as C

import numpy as np

SVX_DEL, SVX_INS, SVX_INV, SVX_DUP_TAN, SVX_BND, SVX_DUP_INT = range(6)
TYPE_NAMES = ("DEL", "INS", "INV", "DUP_TAN", "BND", "DUP_INT")
TYPE_CODE = {n: i for i, n in enumerate(TYPE_NAMES)}
INV_DIRECTIONS = ("left_fwd", "left_rev", "right_fwd", "right_rev", "all")
INV_DIR_CODE = {n: i for i, n in enumerate(INV_DIRECTIONS)}
SRC_NAMES = ("cigar", "suppl")
SVX_FLAG_SKIP = 0x8000
NIBBLE = "=ACMGRSVTWYHKDBN"

ERRORS = {-1: "SVX_E_NODEVICE", -2: "SVX_E_HIP", -3: "SVX_E_ARG", -4: "SVX_E_CAPACITY", -5: "SVX_E_STATE"}

# translation table: ASCII (any case) -> 4-bit code; 255 marks symbols outside the BAM alphabet
_ENC = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(NIBBLE):
    _ENC[ord(_c)] = _i
    _ENC[ord(_c.lower())] = _i
_DEC = np.frombuffer(NIBBLE.encode("ascii"), dtype=np.uint8)
_DEC_TABLE = bytes(NIBBLE.encode("ascii")[i & 15] for i in range(256))        # bytes.translate table: code -> symbol (codes are < 16)


def encode_bases(s):
    """str/bytes -> uint8 codes (upper-cased); ValueError for symbols outside '=ACMGRSVTWYHKDBN'."""
    if isinstance(s, str):
        s = s.encode("ascii")
    codes = _ENC[np.frombuffer(s, dtype=np.uint8)]
    if codes.size and codes.max() == 255:
        bad = sorted(set(chr(b) for b, c in zip(s, codes) if c == 255))
        raise ValueError("sequence symbol(s) %r outside the IUPAC/BAM alphabet %r" % (bad, NIBBLE))
    return codes


def decode_bases(codes):
    # bytes.translate: one C loop over the codes (a numpy look-up of the same table costs 20 ms per 5 M bases - half of what building a table's objects took)
    return np.ascontiguousarray(codes, dtype=np.uint8).tobytes().translate(_DEC_TABLE).decode("ascii")


class Params(C.Structure):
    _fields_ = [("min_mapq", C.c_int32), ("min_sv_size", C.c_int32), ("max_sv_size", C.c_int32),
                ("segment_gap_tolerance", C.c_int32), ("segment_overlap_tolerance", C.c_int32),
                ("all_bnds", C.c_int32), ("partition_max_distance", C.c_int64),
                ("position_distance_normalizer", C.c_double), ("edit_distance_normalizer", C.c_double),
                ("cluster_max_distance", C.c_double)]

    @classmethod
    def from_options(cls, o):
        g = lambda k, d: getattr(o, k, d)      # noqa: E731
        return cls(int(g("min_mapq", 20)), int(g("min_sv_size", 40)), int(g("max_sv_size", 100000)),
                   int(g("segment_gap_tolerance", 10)), int(g("segment_overlap_tolerance", 5)),
                   1 if g("all_bnds", False) else 0, int(g("partition_max_distance", 1000)),
                   float(g("position_distance_normalizer", 900)), float(g("edit_distance_normalizer", 1.0)),
                   float(g("cluster_max_distance", 0.5)))


_P = C.c_void_p


class Batch(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("n_rec", C.c_int64),
                ("flag", _P), ("tid", _P), ("pos", _P), ("mapq", _P), ("lseq", _P), ("read_id", _P),
                ("order", _P), ("seg_order", _P), ("cigar_off", _P), ("cigar", _P), ("seq_off", _P), ("seq", _P),
                ("seg_off", _P), ("n_seg", C.c_int64), ("seg_tid", _P), ("seg_pos", _P), ("seg_rev", _P),
                ("seg_mapq", _P), ("seg_lseq", _P), ("seg_cigar_off", _P), ("seg_cigar", _P),
                ("n_contig", C.c_int32), ("contig_rank", _P),
                ("seq_rng_off", _P), ("seq_rng_q0", _P), ("seq_rng_len", _P), ("seq_rng_byte", _P), ("n_seq_rng", C.c_int64)]


BATCH_DTYPES = dict(flag=np.uint16, tid=np.int32, pos=np.int32, mapq=np.uint8, lseq=np.int32, read_id=np.int32,
                    order=np.uint32, seg_order=np.uint32, cigar_off=np.uint64, cigar=np.uint32, seq_off=np.uint64,
                    seq=np.uint8, seg_off=np.uint32, seg_tid=np.int32, seg_pos=np.int32, seg_rev=np.uint8,
                    seg_mapq=np.uint8, seg_lseq=np.int32, seg_cigar_off=np.uint64, seg_cigar=np.uint32,
                    contig_rank=np.int32)


class SigView(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("n", C.c_int64), ("key", _P), ("type", _P), ("src", _P), ("aux", _P),
                ("contig", _P), ("start", _P), ("end", _P), ("contig2", _P), ("pos2", _P), ("read_id", _P),
                ("seq_off", _P), ("seq", _P)]


SIG_DTYPES = dict(key=np.uint64, type=np.uint8, src=np.uint8, aux=np.uint8, contig=np.int32, start=np.int32,
                  end=np.int32, contig2=np.int32, pos2=np.int32, read_id=np.int32)


class Genome(C.Structure):
    _fields_ = [("on_device", C.c_int32), ("n_contig", C.c_int32), ("off", _P), ("codes", _P)]


class ClusterView(C.Structure):
    _fields_ = [("n", C.c_int64), ("type_count", C.c_int64 * 6), ("type", _P), ("contig", _P), ("start", _P),
                ("end", _P), ("contig2", _P), ("start2", _P), ("end2", _P), ("aux", _P), ("score", _P),
                ("std_span", _P), ("std_pos", _P), ("size", _P), ("member_off", _P), ("members", _P),
                ("n_members", C.c_int64)]


class AlnIndex(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_contig", C.c_int32), ("reserved", C.c_int32), ("contig_first", _P), ("contig_len", _P),
                ("pos", _P), ("end", _P), ("flag", _P), ("mapq", _P), ("name_id", _P)]


CLU_DTYPES = dict(type=np.uint8, contig=np.int32, start=np.int32, end=np.int32, contig2=np.int32, start2=np.int32,
                  end2=np.int32, aux=np.uint8, score=np.float64, std_span=np.float64, std_pos=np.float64,
                  size=np.int32)


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("t_collect_ms", "t_cluster_ms", "t_cigar_scan_ms", "t_segments_ms",
                                          "t_sort_ms", "t_partition_ms", "t_edit_ms", "t_linkage_ms", "t_gather_ms")] + \
               [(n, C.c_int64) for n in ("n_rec_used", "n_ops", "n_seg", "n_seg_ops", "n_sig", "n_bnd_side",
                                         "n_ins_bases", "n_partitions", "n_large_partitions", "n_pairs",
                                         "n_edit_pairs", "n_edit_cells", "n_clusters", "n_hap_bytes",
                                         "n_edit_wordcols_issued", "n_edit_wordcols_useful", "n_edit_wordcols_retry",
                                         "n_edit_wordcols_band")] + [("edit_guess", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def ptr(a):
    """numpy array (C-contiguous) or torch tensor or int -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())          # torch tensor


class SigTable(object):
    """Host-side signature table (numpy SoA) - the currency between COLLECT, CLUSTER and the Python objects."""
    __slots__ = tuple(SIG_DTYPES) + ("seq_off", "seq", "n")

    def __init__(self, n, n_seq=0):
        self.n = n
        for k, dt in SIG_DTYPES.items():
            setattr(self, k, np.zeros(n, dtype=dt))
        self.seq_off = np.zeros(n + 1, dtype=np.int64)
        self.seq = np.zeros(max(1, n_seq), dtype=np.uint8)

    def view(self):
        v = SigView()
        v.on_device = 0
        v.n = self.n
        for k in SIG_DTYPES:
            setattr(v, k, ptr(getattr(self, k)))
        v.seq_off = ptr(self.seq_off)
        v.seq = ptr(self.seq)
        return v

    def sequence(self, i):
        return decode_bases(self.seq[self.seq_off[i]:self.seq_off[i + 1]])

    def equal(self, other, with_key=True):
        if self.n != other.n:
            return False
        for k in SIG_DTYPES:
            if k == "key" and not with_key:
                continue
            if not np.array_equal(getattr(self, k), getattr(other, k)):
                return False
        if not np.array_equal(self.seq_off, other.seq_off):
            return False
        m = int(self.seq_off[self.n])
        return np.array_equal(self.seq[:m], other.seq[:m])

    def first_difference(self, other):
        if self.n != other.n:
            return "n: %d != %d" % (self.n, other.n)
        for k in list(SIG_DTYPES) + ["seq_off"]:
            a, b = getattr(self, k), getattr(other, k)
            if not np.array_equal(a, b):
                i = int(np.nonzero(a != b)[0][0])
                return "%s[%d]: %r != %r" % (k, i, a[i], b[i])
        m = int(self.seq_off[self.n])
        if not np.array_equal(self.seq[:m], other.seq[:m]):
            return "seq differs"
        return None


class ClusterTable(object):
    __slots__ = tuple(CLU_DTYPES) + ("member_off", "members", "n", "n_members", "type_count")

    def __init__(self, n, n_members):
        self.n, self.n_members = n, n_members
        for k, dt in CLU_DTYPES.items():
            setattr(self, k, np.zeros(max(1, n), dtype=dt))
        self.member_off = np.zeros(n + 1, dtype=np.int64)
        self.members = np.zeros(max(1, n_members), dtype=np.int32)
        self.type_count = [0] * 6

    def view(self):
        v = ClusterView()
        v.n = self.n
        v.n_members = self.n_members
        for k in CLU_DTYPES:
            setattr(v, k, ptr(getattr(self, k)))
        v.member_off = ptr(self.member_off)
        v.members = ptr(self.members)
        return v

    def finish(self, v):
        self.type_count = list(v.type_count)
        for k in CLU_DTYPES:
            setattr(self, k, getattr(self, k)[:self.n])
        self.members = self.members[:self.n_members]

    def first_difference(self, other, rtol=0.0):
        if self.n != other.n:
            return "n: %d != %d" % (self.n, other.n)
        if list(self.type_count) != list(other.type_count):
            return "type_count %r != %r" % (self.type_count, other.type_count)
        for k in CLU_DTYPES:
            a, b = getattr(self, k), getattr(other, k)
            if a.dtype == np.float64:
                same = (np.isnan(a) & np.isnan(b)) | (a == b) if rtol == 0.0 else \
                    (np.isnan(a) & np.isnan(b)) | (np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b)))
            else:
                same = a == b
            if not same.all():
                i = int(np.nonzero(~same)[0][0])
                return "%s[%d]: %r != %r" % (k, i, a[i], b[i])
        if not np.array_equal(self.member_off, other.member_off):
            return "member_off differs"
        if not np.array_equal(self.members, other.members):
            return "members differ"
        return None
