"""ctypes binding of svim_amd/libsvx.so (the C ABI declared in include/svx.h).

There is no CPU fallback: if the HIP library is missing, or no MI355X is visible, every entry point of the
package raises (SvxError) instead of silently computing somewhere else.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi
from ._abi import ClusterTable, SigTable, ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("SVX_LIB") or os.path.join(_HERE, "libsvx.so")      # SVX_LIB: an experiment build (tools/build_variants.sh)
_LIB = None
_ENGINES = {}

SYMBOLS = ["svx_ctx_create", "svx_ctx_destroy", "svx_last_error", "svx_version", "svx_get_stats", "svx_stream", "svx_cluster_partitions_fetch", "svx_memcpy_d2h", "svx_memcpy_h2d", "svx_dev_alloc", "svx_dev_free", "svx_host_alloc", "svx_host_free", "svx_device_synchronize", "svx_selftest_prims", "svx_bam_set_device_decode",
           "svx_collect", "svx_collect_count", "svx_collect_fetch", "svx_collect_accumulate", "svx_collect_set_slot_base", "svx_set_genome", "svx_cluster",
           "svx_cluster_count", "svx_cluster_fetch", "svx_cluster_set_ranks", "svx_cluster_abort_ranks", "svx_cluster_stream_positions",
           "svx_set_alignment_index", "svx_genotype",
           "svx_cigar_indel", "svx_edit_distance", "svx_linkage_fcluster", "svx_pair_distances",
           "svx_bam_open", "svx_bam_close", "svx_bam_header", "svx_bam_read_batch", "svx_bam_read_names", "svx_bam_set_seq_filter", "svx_bam_rewind", "svx_bam_seek", "svx_bam_set_gpu_inflate", "svx_bam_gpu_inflate_stats",
           "svx_inflater_create", "svx_inflater_destroy", "svx_inflater_staging", "svx_inflater_enqueue", "svx_inflater_wait",
           "svx_inflater_run"]


class SvxError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")])
    return _LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise SvxError("svim_amd/libsvx.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C svim_amd/csrc`); the GPU path has no CPU fallback")
        L = C.CDLL(_LIB_PATH)
        L.svx_last_error.restype = C.c_char_p
        L.svx_stream.restype = C.c_void_p
        _LIB = L
    return _LIB


def host_empty(n, dtype):
    """numpy array of n elements in page-locked memory the LIBRARY owns (svx_host_alloc, include/svx.h): svx_collect uploads such an array without the bounce
    pass a pageable numpy array costs.  The block goes back to the library when the array (and every view of it) is gone.  Raises SvxError when the memory
    cannot be had (no GPU): callers that only want the speed-up fall back to numpy themselves."""
    import weakref
    L = lib()
    L.svx_host_alloc.restype = C.c_void_p
    L.svx_host_alloc.argtypes = [C.c_uint64]
    L.svx_host_free.argtypes = [C.c_void_p]
    dt = np.dtype(dtype)
    nbytes = max(1, int(n) * dt.itemsize)
    p = L.svx_host_alloc(C.c_uint64(nbytes))
    if not p:
        raise SvxError("svx_host_alloc(%d) failed: %s" % (nbytes, L.svx_last_error().decode("utf-8", "replace")))
    buf = (C.c_uint8 * nbytes).from_address(p)
    weakref.finalize(buf, L.svx_host_free, C.c_void_p(p))
    return np.frombuffer(buf, dtype=dt, count=int(n))


def _check(rc, what):
    if rc != 0:
        msg = lib().svx_last_error().decode("utf-8", "replace")
        raise SvxError("%s failed: %s (%s)" % (what, _abi.ERRORS.get(rc, rc), msg))


class Engine(object):
    """One libsvx context on one GPU."""

    def __init__(self, device=0):
        self.L = lib()
        self.ctx = C.c_void_p()
        _check(self.L.svx_ctx_create(C.c_int(device), C.byref(self.ctx)), "svx_ctx_create")
        self.device = device
        self._keep = []
        self.collect_generation = 0          # bumped by every svx_collect: identifies which tables the context holds (svim_amd/lazy.py)

    def close(self):
        if self.ctx:
            self.L.svx_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- COLLECT ----
    def collect(self, hb, params, fetch=True):
        b = hb.struct() if hasattr(hb, "struct") else hb
        self.collect_generation += 1
        _check(self.L.svx_collect(self.ctx, C.byref(b), C.byref(params)), "svx_collect")
        if not fetch:
            return None
        return self.fetch_signatures(0), self.fetch_signatures(1)

    def accumulate(self, on):
        """on: every following collect() appends to the resident signature lists (the batches of one file); off: one batch per call"""
        _check(self.L.svx_collect_accumulate(self.ctx, C.c_int(1 if on else 0)), "svx_collect_accumulate")
        self.collect_generation += 1

    def set_slot_base(self, base):
        _check(self.L.svx_collect_set_slot_base(self.ctx, C.c_uint64(int(base))), "svx_collect_set_slot_base")

    def collect_counts(self):
        n, ns, nb = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self.L.svx_collect_count(self.ctx, C.byref(n), C.byref(ns), C.byref(nb)), "svx_collect_count")
        return n.value, ns.value, nb.value

    def fetch_signatures(self, which):
        n, ns, nb = self.collect_counts()
        t = SigTable(n if which == 0 else nb, ns if which == 0 else 0)
        v = t.view()
        _check(self.L.svx_collect_fetch(self.ctx, which, C.byref(v)), "svx_collect_fetch")
        return t

    # ---- CLUSTER ----
    def set_genome(self, off, codes, on_device=False):
        g = _abi.Genome(1 if on_device else 0, len(off) - 1 if not on_device else int(off.shape[0]) - 1, ptr(off), ptr(codes))
        self._keep = [off, codes]
        _check(self.L.svx_set_genome(self.ctx, C.byref(g)), "svx_set_genome")

    def cluster(self, params, contig_rank, table=None, source=2, fetch=True):
        v = table.view() if (table is not None and hasattr(table, "view")) else (table if table is not None else _abi.SigView())
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        _check(self.L.svx_cluster(self.ctx, source, C.byref(v), len(rank), ptr(rank), C.byref(params)), "svx_cluster")
        if not fetch:
            return None
        return self.fetch_clusters()

    def partitions(self):
        """partitions of the last cluster() call (svx_cluster_partitions_fetch): list of lists of signature indices, in the order they were formed"""
        n, npart = C.c_int64(), C.c_int64()
        _check(self.L.svx_cluster_partitions_fetch(self.ctx, C.byref(n), C.byref(npart), None, None), "svx_cluster_partitions_fetch")
        if n.value <= 0:
            return []
        sidx, start = np.zeros(n.value, dtype=np.uint32), np.zeros(npart.value + 1, dtype=np.int64)
        _check(self.L.svx_cluster_partitions_fetch(self.ctx, None, None, ptr(sidx), ptr(start)), "svx_cluster_partitions_fetch")
        return [[int(i) for i in sidx[int(start[k]):int(start[k + 1])]] for k in range(npart.value)]

    def fetch_clusters(self):
        n, nm = C.c_int64(), C.c_int64()
        _check(self.L.svx_cluster_count(self.ctx, C.byref(n), C.byref(nm)), "svx_cluster_count")
        ct = ClusterTable(n.value, nm.value)
        cv = ct.view()
        _check(self.L.svx_cluster_fetch(self.ctx, C.byref(cv)), "svx_cluster_fetch")
        ct.finish(cv)
        return ct

    def set_alignment_index(self, index):
        """index: svim_amd.SVIM_genotyping.AlignmentIndex (arrays are copied to the device)"""
        v = index.view()
        _check(self.L.svx_set_alignment_index(self.ctx, C.byref(v)), "svx_set_alignment_index")

    def genotype(self, mode, tid, start, end, member_off, member_names, min_mapq):
        n = len(tid)
        out = np.zeros(max(1, n), dtype=np.int32)
        tid = np.ascontiguousarray(tid, dtype=np.int32); start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32); member_off = np.ascontiguousarray(member_off, dtype=np.int64)
        member_names = np.ascontiguousarray(member_names, dtype=np.int32)
        _check(self.L.svx_genotype(self.ctx, C.c_int32(mode), C.c_int64(n), ptr(tid), ptr(start), ptr(end), ptr(member_off),
                                   ptr(member_names if member_names.size else np.zeros(1, np.int32)), C.c_int32(min_mapq), ptr(out)), "svx_genotype")
        return out[:n]

    def set_ranks(self, rank, world, allgather=None):
        """Contig-sharded ranks (svx_cluster_set_ranks): this engine is rank `rank` of `world`; allgather(send: bytes) -> bytes of all ranks, rank-major
        (len(send) * world) - or an object that also has gather_into(send_addr, recv_addr, nbytes) - is the transport svx_cluster uses to find where its random.sample streams start.  allgather None / world 1: single rank."""
        if allgather is None or world <= 1:
            self._ag_cb = None
            _check(self.L.svx_cluster_set_ranks(self.ctx, 0, 1, None, None), "svx_cluster_set_ranks")
            return
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

        into = getattr(allgather, "gather_into", None)      # a transport that works on the buffers themselves (multigpu.TorchAllGather)

        def tramp(user, send, recv, nbytes):
            try:
                if into is not None and nbytes > 0:
                    into(int(send), int(recv), int(nbytes))
                    return 0
                got = allgather(C.string_at(send, nbytes))
                if len(got) != nbytes * world:
                    raise ValueError("all-gather callback returned %d bytes, expected %d" % (len(got), nbytes * world))
                C.memmove(recv, got, len(got))
                return 0
            except Exception:                       # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._ag_cb = proto(tramp)
        _check(self.L.svx_cluster_set_ranks(self.ctx, int(rank), int(world), self._ag_cb, None), "svx_cluster_set_ranks")

    def abort_ranks(self):
        """tell the other ranks this one will not reach svx_cluster (they fail instead of waiting in the rank exchange)"""
        _check(self.L.svx_cluster_abort_ranks(self.ctx), "svx_cluster_abort_ranks")

    def stream_positions(self):
        """(start[6], end[6]): where each type's random.sample stream started / stopped on this rank in the last cluster call"""
        a, b = (C.c_int64 * 6)(), (C.c_int64 * 6)()
        _check(self.L.svx_cluster_stream_positions(self.ctx, a, b), "svx_cluster_stream_positions")
        return list(a), list(b)

    def stats(self):
        s = _abi.Stats()
        _check(self.L.svx_get_stats(self.ctx, C.byref(s)), "svx_get_stats")
        return s.as_dict()

    def stream(self):
        return self.L.svx_stream(self.ctx)

    def selftest_prims(self, n, begin_bit=0, end_bit=64, seed=1):
        """the library's own radix sort + exclusive scan on n pseudo-random elements against std::stable_sort / a serial sum (raises on a difference)"""
        _check(self.L.svx_selftest_prims(self.ctx, C.c_int64(n), C.c_int32(begin_bit), C.c_int32(end_bit), C.c_uint64(seed)), "svx_selftest_prims")

    # ---- single-function entry points ----
    def cigar_indel(self, tuples, min_length):
        c = np.array([(l << 4) | op for op, l in tuples] or [0], dtype=np.uint32)
        n = len(tuples)
        o_ref = np.zeros(max(1, n), dtype=np.int64)
        o_read = np.zeros(max(1, n), dtype=np.int64)
        o_len = np.zeros(max(1, n), dtype=np.int32)
        o_del = np.zeros(max(1, n), dtype=np.uint8)
        m = C.c_int64()
        _check(self.L.svx_cigar_indel(self.ctx, ptr(c), C.c_int64(n), C.c_int32(min_length), ptr(o_ref), ptr(o_read),
                                      ptr(o_len), ptr(o_del), C.byref(m)), "svx_cigar_indel")
        return [(int(o_ref[i]), int(o_read[i]), int(o_len[i]), "DEL" if o_del[i] else "INS") for i in range(m.value)]

    def edit_distances(self, pairs):
        """pairs: list of (a, b) strings -> list of unit-cost global edit distances."""
        chunks, a_off, b_off = [], [0], [0]
        pos = 0
        # layout: all a strings, then all b strings; a_off/b_off index into the same code array
        for a, _ in pairs:
            c = _abi.encode_bases(a)
            chunks.append(c)
            pos += c.size
            a_off.append(pos)
        b_off = [pos]
        for _, b in pairs:
            c = _abi.encode_bases(b)
            chunks.append(c)
            pos += c.size
            b_off.append(pos)
        codes = np.concatenate(chunks + [np.zeros(1, np.uint8)])
        a_off = np.array(a_off, dtype=np.int64)
        b_off = np.array(b_off, dtype=np.int64)
        out = np.zeros(max(1, len(pairs)), dtype=np.int32)
        _check(self.L.svx_edit_distance(self.ctx, C.c_int64(len(pairs)), ptr(codes), ptr(a_off), ptr(b_off), ptr(out)),
               "svx_edit_distance")
        return [int(x) for x in out[:len(pairs)]]

    def pair_distances(self, table, pairs, params):
        """span_position_distance of (i, j) pairs of a host SigTable through the device path -> float64 array"""
        ia = np.ascontiguousarray([i for i, _ in pairs], dtype=np.int64)
        ib = np.ascontiguousarray([j for _, j in pairs], dtype=np.int64)
        out = np.zeros(max(1, len(pairs)), dtype=np.float64)
        v = table.view()
        _check(self.L.svx_pair_distances(self.ctx, C.byref(v), C.c_int64(len(pairs)), ptr(ia), ptr(ib), C.byref(params), ptr(out)), "svx_pair_distances")
        return out[:len(pairs)]

    def linkage_fcluster(self, problems, cutoff):
        """problems: list of (n, condensed distance array) -> list of label arrays (1-based, scipy numbering)."""
        ns = np.array([p[0] for p in problems], dtype=np.int32)
        d_off = np.zeros(len(problems) + 1, dtype=np.int64)
        l_off = np.zeros(len(problems) + 1, dtype=np.int64)
        for i, (n, d) in enumerate(problems):
            d_off[i + 1] = d_off[i] + n * (n - 1) // 2
            l_off[i + 1] = l_off[i] + n
        d = np.concatenate([np.asarray(p[1], dtype=np.float64).ravel() for p in problems] + [np.zeros(1)])
        lab = np.zeros(max(1, int(l_off[-1])), dtype=np.int32)
        _check(self.L.svx_linkage_fcluster(self.ctx, C.c_int64(len(problems)), ptr(ns), ptr(d_off), ptr(d), C.c_double(cutoff),
                                           ptr(l_off), ptr(lab)), "svx_linkage_fcluster")
        return [lab[l_off[i]:l_off[i + 1]].copy() for i in range(len(problems))]


def engine(device=None):
    """Process-wide engine for `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    e = _ENGINES.get(device)
    if e is None:
        e = _ENGINES[device] = Engine(device)
    return e


def bgzf_blocks(path):
    """(payload bytes, ISIZE) of every BGZF block of a file: the raw DEFLATE stream between the block header and its CRC32 / ISIZE trailer"""
    import struct
    with open(path, "rb") as fh:
        data = fh.read()
    at, out = 0, []
    while at + 18 <= len(data):
        if data[at:at + 2] != b"\x1f\x8b":
            raise ValueError("not a BGZF block at %d" % at)
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        p, bsize = at + 12, None
        while p + 4 <= at + 12 + xlen:
            si, sl = data[p:p + 2], struct.unpack_from("<H", data, p + 2)[0]
            if si == b"BC":
                bsize = struct.unpack_from("<H", data, p + 4)[0]
            p += 4 + sl
        if bsize is None:
            raise ValueError("no BC subfield")
        blen = bsize + 1
        out.append((data[at + 12 + xlen:at + blen - 8], struct.unpack_from("<I", data, at + blen - 4)[0]))
        at += blen
    return out


class Inflater(object):
    """svx_inflater: BGZF payloads inflated on the GPU, one wavefront per block (svim_amd/csrc/bgzf.hip).  No CPU fallback."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.L.svx_inflater_staging.restype = C.c_void_p
        _check(self.L.svx_inflater_create(C.c_int(device), C.byref(self.h)), "svx_inflater_create")
        self.kernel_ms = 0.0

    def close(self):
        if self.h:
            self.L.svx_inflater_destroy(self.h)
            self.h = None

    def inflate(self, blocks):
        """blocks: [(payload bytes, isize)] -> the inflated stream as a numpy uint8 array (host); self.kernel_ms = duration of the launch"""
        n = len(blocks)
        in_off = np.zeros(max(1, n), dtype=np.uint64)
        clen = np.array([len(b) for b, _ in blocks] or [0], dtype=np.uint32)
        isize = np.array([s for _, s in blocks] or [0], dtype=np.uint32)
        out_at = np.zeros(max(1, n), dtype=np.uint64)
        at = 0
        for i, (b, _) in enumerate(blocks):
            in_off[i] = at
            at += (len(b) + 7) & ~7
        if n > 1:
            out_at[1:n] = np.cumsum(isize[:n - 1].astype(np.uint64))
        total_out = int(isize[:n].astype(np.uint64).sum()) if n else 0
        stage = self.L.svx_inflater_staging(self.h, C.c_int(0), C.c_uint64(max(at, 8)))
        if not stage:
            raise SvxError("svx_inflater_staging failed")
        buf = (C.c_uint8 * max(at, 8)).from_address(stage)
        view = np.frombuffer(buf, dtype=np.uint8)
        for i, (b, _) in enumerate(blocks):
            o = int(in_off[i])
            view[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
        out = np.zeros(max(1, total_out), dtype=np.uint8)
        ms = C.c_float(0)
        _check(self.L.svx_inflater_run(self.h, C.c_int64(n), ptr(in_off), ptr(clen), ptr(isize), ptr(out_at), C.c_uint64(at), ptr(out),
                                       C.c_uint64(total_out), C.c_int(0), C.byref(ms)), "svx_inflater_run")
        self.kernel_ms = float(ms.value)
        return out[:total_out]
