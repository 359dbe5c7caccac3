"""ctypes wrapper of the native BAM front-end (svim_amd/csrc/bamio.cpp): BGZF inflate + record decode into the record
batch, on the host, without pysam.  Needs only libsvx.so - no GPU - so it is usable (and tested) on CPU."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import SvxError, lib


class NativeBam(object):
    """BAM reader with the handful of pysam.AlignmentFile members SVIM's COLLECT driver uses (references, lengths,
    get_tid/getrname, header['HD']['SO']) plus read_batch(), which yields ready-made svx_batch structs."""

    def __init__(self, path, threads=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.svx_bam_open(path.encode(), C.c_int(threads), C.byref(self.h))
        if rc != 0:
            raise SvxError("svx_bam_open(%r) failed: %s" % (path, self.L.svx_last_error().decode()))
        n = C.c_int32()
        names, lens, so = C.c_char_p(), C.POINTER(C.c_int32)(), C.c_char_p()
        blob = C.c_void_p()
        self.L.svx_bam_header(self.h, C.byref(n), C.byref(blob), C.byref(lens), C.byref(so))
        # names: NUL separated, n of them
        out, p = [], blob.value
        for _ in range(n.value):
            s = C.string_at(p)
            out.append(s.decode("ascii"))
            p += len(s) + 1
        self.references = out
        self.lengths = [lens[i] for i in range(n.value)]
        self.sort_order = (so.value or b"").decode("ascii")
        self.header = {"HD": {"SO": self.sort_order}} if self.sort_order else {}
        self._tid = {r: i for i, r in enumerate(self.references)}
        self.filename = path

    def get_tid(self, name):
        return self._tid.get(name, -1)

    def getrname(self, tid):
        return self.references[tid]

    get_reference_name = getrname

    def set_seq_filter(self, min_ins_len):
        """coordinate mode: keep only the SEQ ranges COLLECT can read (svx_bam_set_seq_filter); pass options.min_sv_size"""
        rc = self.L.svx_bam_set_seq_filter(self.h, C.c_int(int(min_ins_len)))
        if rc != 0:
            raise SvxError("svx_bam_set_seq_filter failed")

    def set_gpu_inflate(self, device):
        """BGZF inflate shared between the GPU (device >= 0) and the host's cores (svx_bam_set_gpu_inflate); device < 0 switches it off"""
        if self.L.svx_bam_set_gpu_inflate(self.h, C.c_int(int(device))) != 0:
            raise SvxError("svx_bam_set_gpu_inflate failed: %s" % self.L.svx_last_error().decode())

    def set_device_decode(self, device):
        """coordinate mode: BGZF inflate, record discovery and decode on GPU `device` (svx_bam_set_device_decode); read_batch then returns
        batches whose arrays live in HBM.  device < 0: back to the host reader"""
        if self.L.svx_bam_set_device_decode(self.h, C.c_int(int(device))) != 0:
            raise SvxError("svx_bam_set_device_decode failed: %s" % self.L.svx_last_error().decode())
        self.device_decode = int(device) >= 0

    def gpu_inflate_stats(self):
        g, c, ms = C.c_int64(), C.c_int64(), C.c_double()
        self.L.svx_bam_gpu_inflate_stats(self.h, C.byref(g), C.byref(c), C.byref(ms))
        return {"gpu_blocks": g.value, "cpu_blocks": c.value, "gpu_kernel_ms": ms.value}

    def seek(self, voff, last_tid=-2):
        """continue at BGZF virtual offset `voff`; records beyond reference id `last_tid` end the reading (svx_bam_seek)"""
        if self.L.svx_bam_seek(self.h, C.c_uint64(int(voff)), C.c_int32(int(last_tid))) != 0:
            raise SvxError("svx_bam_seek failed: %s" % self.L.svx_last_error().decode())

    def rewind(self):
        """back to the first record; buffers, threads and interned names are kept (svx_bam_rewind)"""
        if self.L.svx_bam_rewind(self.h) != 0:
            raise SvxError("svx_bam_rewind failed: %s" % self.L.svx_last_error().decode())

    def read_batch(self, max_records, min_mapq, mode="coordinate"):
        """-> (svx_batch struct with host pointers owned by the reader, n_records); n_records == 0 at EOF."""
        b = _abi.Batch()
        n = C.c_int64()
        rc = self.L.svx_bam_read_batch(self.h, C.c_int64(max_records), C.c_int(0 if mode == "coordinate" else 1), C.c_int(min_mapq),
                                       C.byref(b), C.byref(n))
        if rc != 0:
            raise SvxError("svx_bam_read_batch failed: %s" % self.L.svx_last_error().decode())
        return b, n.value

    def batch_arrays(self, b):
        """numpy copies of a batch returned by read_batch (tests / inspection)."""
        n, ns = b.n_rec, b.n_seg
        if b.on_device:
            return self._device_batch_arrays(b)

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
        A = {}
        for k in ("flag", "tid", "pos", "mapq", "lseq", "read_id", "order", "seg_order"):
            A[k] = arr(getattr(b, k), n, _abi.BATCH_DTYPES[k])
        A["cigar_off"] = arr(b.cigar_off, n + 1, np.uint64)
        A["cigar"] = arr(b.cigar, int(A["cigar_off"][-1]) if n else 0, np.uint32)
        A["seq_off"] = arr(b.seq_off, n + 1, np.uint64)
        A["seq"] = arr(b.seq, int(A["seq_off"][-1]) if n else 0, np.uint8)
        A["seg_off"] = arr(b.seg_off, n + 1, np.uint32)
        for k in ("seg_tid", "seg_pos", "seg_rev", "seg_mapq", "seg_lseq"):
            A[k] = arr(getattr(b, k), ns, _abi.BATCH_DTYPES[k])
        A["seg_cigar_off"] = arr(b.seg_cigar_off, ns + 1, np.uint64)
        A["seg_cigar"] = arr(b.seg_cigar, int(A["seg_cigar_off"][-1]) if ns else 0, np.uint32)
        A["contig_rank"] = arr(b.contig_rank, b.n_contig, np.int32)
        if b.seq_rng_off:
            nr = int(b.n_seq_rng)
            A["seq_rng_off"] = arr(b.seq_rng_off, n + 1, np.uint32)
            A["seq_rng_q0"], A["seq_rng_len"] = arr(b.seq_rng_q0, nr, np.int32), arr(b.seq_rng_len, nr, np.int32)
            A["seq_rng_byte"] = arr(b.seq_rng_byte, nr, np.uint64)
        return A

    def _device_batch_arrays(self, b):
        """a device-resident batch (svx_bam_set_device_decode) copied to the host in the layout of a host batch: offsets rebased to the batch, one
        packed SEQ per record (the device batch points into the inflated stream instead)"""
        n, ns = int(b.n_rec), int(b.n_seg)

        def arr(ptr, count, dt, skip=0):
            out = np.zeros(count, dtype=dt)
            if count:
                base = C.cast(ptr, C.c_void_p).value
                src = C.c_void_p(base + skip * np.dtype(dt).itemsize)
                if self.L.svx_memcpy_d2h(out.ctypes.data_as(C.c_void_p), src, C.c_uint64(out.nbytes)) != 0:
                    raise SvxError("svx_memcpy_d2h of %d x %s from %#x + %d elements failed: %s" % (count, np.dtype(dt).name, base or 0, skip, self.L.svx_last_error().decode()))
            return out
        A = {}
        for k in ("flag", "tid", "pos", "mapq", "lseq", "read_id", "order", "seg_order"):
            A[k] = arr(getattr(b, k), n, _abi.BATCH_DTYPES[k])
        co = arr(b.cigar_off, n + 1, np.uint64)
        A["cigar"] = arr(b.cigar, int(co[-1] - co[0]) if n else 0, np.uint32, skip=int(co[0]) if n else 0)
        A["cigar_off"] = (co - co[0]).astype(np.uint64) if n else np.zeros(1, np.uint64)
        so = arr(b.seq_off, n + 1, np.uint64)
        nb = (A["lseq"].astype(np.int64) + 1) // 2
        parts = []
        if n:
            lo, hi = int(so[0]), int(so[n - 1] + nb[n - 1])
            raw = arr(b.seq, hi - lo, np.uint8, skip=lo)
            parts = [raw[int(so[i]) - lo:int(so[i]) - lo + int(nb[i])] for i in range(n)]
        A["seq"] = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        A["seq_off"] = np.concatenate([[0], np.cumsum(nb)]).astype(np.uint64)
        sg = arr(b.seg_off, n + 1, np.uint32)
        s0, s1 = (int(sg[0]), int(sg[-1])) if n else (0, 0)
        A["seg_off"] = (sg - sg[0]).astype(np.uint32) if n else np.zeros(1, np.uint32)
        for k in ("seg_tid", "seg_pos", "seg_rev", "seg_mapq", "seg_lseq"):
            A[k] = arr(getattr(b, k), s1 - s0, _abi.BATCH_DTYPES[k], skip=s0)
        sc = arr(b.seg_cigar_off, s1 - s0 + 1, np.uint64, skip=s0) if ns else np.zeros(1, np.uint64)
        A["seg_cigar"] = arr(b.seg_cigar, int(sc[-1] - sc[0]), np.uint32, skip=int(sc[0]))
        A["seg_cigar_off"] = (sc - sc[0]).astype(np.uint64)
        A["contig_rank"] = arr(b.contig_rank, b.n_contig, np.int32)
        return A

    def read_names(self):
        n, blob, ln = C.c_int64(), C.c_void_p(), C.c_int64()
        self.L.svx_bam_read_names(self.h, C.byref(n), C.byref(blob), C.byref(ln))
        if n.value == 0:
            return []
        raw = C.string_at(blob, ln.value)
        return raw[:-1].decode("ascii").split("\0")

    def close(self):
        if self.h:
            self.L.svx_bam_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
