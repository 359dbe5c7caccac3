"""ctypes loader for the parity oracle (oracle/libsvx_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from svim_amd import _abi
from svim_amd._abi import ClusterTable, SigTable, ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsvx_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.svo_edit_distance.restype = C.c_int32
    return _LIB


class Oracle(object):
    def __init__(self):
        self.L = lib()
        self.ctx = C.c_void_p()
        self.L.svo_ctx_create(C.byref(self.ctx))
        self._keep = []

    def __del__(self):
        try:
            self.L.svo_ctx_destroy(self.ctx)
        except Exception:
            pass

    def collect(self, hb, params):
        b = hb.struct() if hasattr(hb, "struct") else hb
        rc = self.L.svo_collect(self.ctx, C.byref(b), C.byref(params))
        assert rc == 0, "svo_collect failed (%d)" % rc
        n, ns, nb = C.c_int64(), C.c_int64(), C.c_int64()
        self.L.svo_collect_count(self.ctx, C.byref(n), C.byref(ns), C.byref(nb))
        out = []
        for which, cnt, nseq in ((0, n.value, ns.value), (1, nb.value, 0)):
            t = SigTable(cnt, nseq)
            v = t.view()
            self.L.svo_collect_fetch(self.ctx, which, C.byref(v))
            out.append(t)
        return out[0], out[1]

    def set_genome(self, off, codes):
        g = _abi.Genome(0, len(off) - 1, ptr(off), ptr(codes))
        self._keep = [off, codes]
        self.L.svo_set_genome(self.ctx, C.byref(g))

    def set_threads(self, n):
        """worker threads for the pair distances of cluster() (test infrastructure: full-size parity runs; the results do not depend on n)"""
        self.L.svo_set_threads(self.ctx, int(n))

    def set_chain(self, fn):
        """fn(phase, words): phase 0 fills the six per-type stream start positions into `words` (list of 6 ints, in place), phase 1 receives the six
        end positions - the checker's stand-in for the rank exchange the product does inside svx_cluster (svx_cluster_set_ranks)"""
        if fn is None:
            self._chain_cb = None
            self.L.svo_cluster_set_chain(self.ctx, None, None)
            return
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64))

        def tramp(user, phase, words):
            try:
                w = [int(words[i]) for i in range(6)]
                fn(int(phase), w)
                if phase == 0:
                    for i in range(6):
                        words[i] = int(w[i])
                else:
                    self._last_chain_end = list(w)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._chain_cb = proto(tramp)
        self.L.svo_cluster_set_chain(self.ctx, self._chain_cb, None)

    def cluster(self, params, contig_rank, table=None, source=2):
        v = table.view() if table is not None else _abi.SigView()
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        rc = self.L.svo_cluster(self.ctx, source, C.byref(v), len(rank), ptr(rank), C.byref(params))
        assert rc == 0
        n, nm = C.c_int64(), C.c_int64()
        self.L.svo_cluster_count(self.ctx, C.byref(n), C.byref(nm))
        ct = ClusterTable(n.value, nm.value)
        cv = ct.view()
        self.L.svo_cluster_fetch(self.ctx, C.byref(cv))
        ct.finish(cv)
        return ct

    def stats(self):
        s = _abi.Stats()
        self.L.svo_get_stats(self.ctx, C.byref(s))
        return s.as_dict()

    # ---- single-function hooks ----
    def cigar_indel(self, tuples, min_length):
        c = np.array([(l << 4) | op for op, l in tuples] or [0], dtype=np.uint32)
        n = len(tuples)
        o_ref = np.zeros(max(1, n), dtype=np.int64)
        o_read = np.zeros(max(1, n), dtype=np.int64)
        o_len = np.zeros(max(1, n), dtype=np.int32)
        o_del = np.zeros(max(1, n), dtype=np.uint8)
        m = C.c_int64()
        self.L.svo_cigar_indel(ptr(c), C.c_int64(n), C.c_int32(min_length), ptr(o_ref), ptr(o_read), ptr(o_len),
                               ptr(o_del), C.byref(m))
        return [(int(o_ref[i]), int(o_read[i]), int(o_len[i]), "DEL" if o_del[i] else "INS") for i in range(m.value)]

    def set_alignment_index(self, index):
        self._index = index                      # the C side borrows the arrays
        v = index.view()
        self.L.svo_set_alignment_index(self.ctx, C.byref(v))

    def genotype(self, mode, tid, start, end, member_off, member_names, min_mapq):
        n = len(tid)
        out = np.zeros(max(1, n), dtype=np.int32)
        tid = np.ascontiguousarray(tid, dtype=np.int32); start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32); member_off = np.ascontiguousarray(member_off, dtype=np.int64)
        member_names = np.ascontiguousarray(member_names, dtype=np.int32)
        self.L.svo_genotype(self.ctx, C.c_int32(mode), C.c_int64(n), ptr(tid), ptr(start), ptr(end), ptr(member_off),
                            ptr(member_names if member_names.size else np.zeros(1, np.int32)), C.c_int32(min_mapq), ptr(out))
        return out[:n]

    def edit_distance(self, a, b):
        ca, cb = _abi.encode_bases(a), _abi.encode_bases(b)
        ca = np.ascontiguousarray(np.concatenate([ca, np.zeros(1, np.uint8)]))
        cb = np.ascontiguousarray(np.concatenate([cb, np.zeros(1, np.uint8)]))
        return int(self.L.svo_edit_distance(ptr(ca), C.c_int64(len(a)), ptr(cb), C.c_int64(len(b))))

    def linkage_fcluster(self, n, d, cutoff, want_z=False):
        d = np.ascontiguousarray(d, dtype=np.float64)
        lab = np.zeros(n, dtype=np.int32)
        Z = np.zeros((max(1, n - 1), 4), dtype=np.float64)
        self.L.svo_linkage_fcluster(C.c_int32(n), ptr(d), C.c_double(cutoff), ptr(lab), ptr(Z))
        return (lab, Z) if want_z else lab

    def getrandbits(self, seed, k, count):
        out = np.zeros(count, dtype=np.uint32)
        self.L.svo_getrandbits(C.c_uint32(seed), C.c_int(k), C.c_int64(count), ptr(out))
        return out

    def sample_sequence(self, seed, ns):
        ns = np.array(ns, dtype=np.int64)
        out = np.zeros(len(ns) * 100, dtype=np.int32)
        self.L.svo_sample_sequence(C.c_uint32(seed), C.c_int64(len(ns)), ptr(ns), ptr(out))
        return out.reshape(len(ns), 100)

    def span_position_distance(self, table, i, j, params):
        v = table.view()
        out = C.c_double()
        self.L.svo_span_position_distance(self.ctx, C.byref(v), C.c_int64(i), C.c_int64(j), C.byref(params), C.byref(out))
        return out.value

    def form_partitions(self, table, contig_rank, max_distance):
        v = table.view()
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        sidx = np.zeros(max(1, table.n), dtype=np.int64)
        pid = np.zeros(max(1, table.n), dtype=np.int64)
        self.L.svo_form_partitions(C.byref(v), ptr(rank), C.c_int64(max_distance), ptr(sidx), ptr(pid))
        return sidx[:table.n], pid[:table.n]
