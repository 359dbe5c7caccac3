"""End-to-end driver for BAM inputs: native reader -> record batches -> svx_collect (accumulating on the device) -> svx_cluster.

Replaces the loop of src/svim/SVIM_COLLECT.py:132-167 + cluster_sv_signatures for a file on disk.  Three things keep the GPU fed
(VERDICT r01 item 4):
  * the reader inflates and decodes on all host cores (svim_amd/csrc/bamio.cpp) and hands out batches from two alternating array
    sets, so batch i+1 is read WHILE batch i is uploaded and collected (BamPipeline: one reader thread, one GPU thread; both spend
    their time inside libsvx, i.e. outside the GIL);
  * only the bases COLLECT can read cross PCIe (svx_bam_set_seq_filter: insertions >= min_sv_size + whole split-read primaries);
  * the signature lists of all batches accumulate in HBM (svx_collect_accumulate), CLUSTER starts from them without any transfer.

bench.py uses run_bam() for `--bam/--fasta` (the real configs[2]-[4] inputs) and end_to_end_sample() for the `end_to_end` block of
the default line: a BAM written from a slice of the synthetic batch (write_bam_from_batch - test / bench infrastructure, not on
the product path).
"""
import logging
import os
import threading
import time
import zlib

import numpy as np

from . import _abi, _lib, convert


def effective_cpus():
    """CPUs this process may use: affinity mask capped by the cgroup CPU quota (what `os.cpu_count()` does not tell)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


class BamPipeline(object):
    """reader thread || GPU thread over one BAM file; results stay resident in the engine (accumulated lists)."""

    def __init__(self, path, options, engine, threads=0, batch_records=200_000, mode="coordinate", sparse_seq=True, regions=None, gpu_inflate=None,
                 device_decode=None):
        from .bamio import NativeBam
        self.bam = NativeBam(path, threads=threads)
        dev = getattr(engine, "device", None)
        if device_decode is None:                                   # default on a GPU engine, either sort order (SVX_BAM_DEVICE_DECODE=0: host reader)
            device_decode = dev is not None and mode in ("coordinate", "queryname") and gpu_inflate is not False and os.environ.get("SVX_BAM_DEVICE_DECODE", "1") != "0"
        self.device_decode = bool(device_decode)
        self.options, self.eng, self.mode, self.batch_records = options, engine, mode, batch_records
        self.params = _abi.Params.from_options(options)
        if self.device_decode:
            # inflate, record discovery and decode on the engine's GPU: batches arrive device-resident, only the compressed file crosses PCIe
            self.bam.set_device_decode(int(dev))
        else:
            if gpu_inflate is None and os.environ.get("SVX_BAM_GPU_INFLATE", "1") != "0":
                gpu_inflate = dev                                   # the engine's GPU helps with the inflate (SVX_BAM_GPU_INFLATE=0: host only)
            if gpu_inflate is not None and gpu_inflate is not False:
                self.bam.set_gpu_inflate(int(gpu_inflate))          # BGZF inflate shared between that GPU and the host's cores
            if sparse_seq and mode == "coordinate":
                self.bam.set_seq_filter(int(getattr(options, "min_sv_size", 40)))
        self.regions = regions            # [(virtual offset, last reference id)]: the contig runs this rank reads (None: the whole file)
        self.region_slots = []            # per region: (first local emission slot, records)
        self.stats = {}
        self.interrupted = False          # run() was ended by Ctrl-C (KeyboardInterrupt): the results are those of the batches collected until then

    def run(self):
        bam, eng, p = self.bam, self.eng, self.params
        min_mapq = int(getattr(self.options, "min_mapq", 20))
        free = threading.Semaphore(2)                 # the reader owns two array sets: at most one batch ahead of the GPU thread
        ready, box, err = threading.Semaphore(0), [], []
        t_read = [0.0]
        stop = threading.Event()                      # set when the GPU thread gives up: the reader must leave libsvx before the handle is closed

        def reader():
            try:
                for region in (self.regions if self.regions is not None else [None]):
                    if stop.is_set():
                        break
                    if region is not None:
                        bam.seek(region[0], region[1])
                    while True:
                        free.acquire()
                        if stop.is_set():
                            break
                        t0 = time.perf_counter()
                        b, n = bam.read_batch(self.batch_records, min_mapq, self.mode)
                        t_read[0] += time.perf_counter() - t0
                        if n == 0:
                            free.release()
                            break
                        box.append((b, n, region))
                        ready.release()
                if not stop.is_set():
                    free.acquire()
                box.append((None, 0, None))
                ready.release()
            except BaseException as e:                 # surfaces in the GPU thread
                err.append(e)
                box.append((None, 0, None))
                ready.release()

        eng.accumulate(True)
        th = threading.Thread(target=reader, daemon=True)
        t_start = time.perf_counter()
        th.start()
        n_rec, slot_base, t_gpu, t_wait, n_batches = 0, 0, 0.0, 0.0, 0
        k = 0
        self.region_slots = []
        cur_region = object()
        try:
            while True:
                t0 = time.perf_counter()
                ready.acquire()
                t_wait += time.perf_counter() - t0
                b, n, region = box[k]
                k += 1
                if err:
                    raise err[0]
                if n == 0:
                    break
                if region is not cur_region:
                    cur_region = region
                    self.region_slots.append([slot_base, 0])
                t0 = time.perf_counter()
                eng.set_slot_base(slot_base)
                # A KeyboardInterrupt that arrives while the call is inside libsvx is raised when the call RETURNS (ctypes): the batch's signatures are in the
                # accumulated lists by then, so it is counted before the interrupt goes on to the handler below.  (A collect that fails raises SvxError.)
                def counted():
                    nonlocal t_gpu, slot_base, n_rec, n_batches
                    self.region_slots[-1][1] += n
                    t_gpu += time.perf_counter() - t0
                    slot_base += 2 * n + 2
                    n_rec += n
                    n_batches += 1
                try:
                    eng.collect(b, p, fetch=False)
                except KeyboardInterrupt:
                    counted()
                    raise
                counted()
                free.release()
        except KeyboardInterrupt:
            # src/svim/SVIM_COLLECT.py:126-128,164-166: an interrupt ends the reading, the pipeline goes on with what was collected - here the batches whose
            # svx_collect has completed (their signatures are in the accumulated lists on the device, and the counters above say so); the batch being READ is dropped
            logging.warning('Execution interrupted by user. Stop detection and continue with next step..')
            self.interrupted = True
        finally:
            # whatever happened (a failing collect, an interrupt): the reader thread leaves libsvx BEFORE anyone may close the handle it reads from
            stop.set()
            free.release()
            free.release()
            th.join()
        t_collect_done = time.perf_counter()
        self.stats = dict(records=n_rec, batches=n_batches, t_collect_wall=t_collect_done - t_start, t_reader_busy=t_read[0], t_gpu_collect=t_gpu,
                          t_gpu_waits_for_reader=t_wait)
        try:
            self.stats["inflate"] = self.bam.gpu_inflate_stats()       # cumulative over the passes of this reader
        except Exception:
            self.stats["inflate"] = None
        return n_rec

    def cluster(self, genome=None):
        """CLUSTER from the accumulated lists.  genome: (off, codes[, on_device]) or None when already set"""
        if genome is not None:
            self.eng.set_genome(*genome)
        from .batch import contig_ranks
        t0 = time.perf_counter()
        self.eng.cluster(self.params, contig_ranks(self.bam.references), source=0, fetch=False)
        self.stats["t_cluster_wall"] = time.perf_counter() - t0

    def rewind(self):
        """back to the first record for another pass: buffers, worker threads and read names are kept - the state a long file is in
        after its first few batches (no first-touch allocation on the host or the device)"""
        self.bam.rewind()

    def close(self):
        self.eng.accumulate(False)
        self.bam.close()


# ---------------------------------------------------------------------------------------------------------------------
# BAM writer from a Structure-of-Arrays batch (bench / test infrastructure)
# ---------------------------------------------------------------------------------------------------------------------
_CIG = "MIDNSHP=XB"


def _bgzf_block(payload, level=1):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(payload) + co.flush()
    head = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(comp) + 25).to_bytes(2, "little")
    return head + comp + (zlib.crc32(payload) & 0xffffffff).to_bytes(4, "little") + len(payload).to_bytes(4, "little")


def write_bam_from_batch(path, hb, references, lengths, name_fmt="r%08d", threads=None, qual_seed=None):
    """HostBatch (numpy SoA; e.g. DeviceBatch.slice_records) -> coordinate-sorted BAM file: fixed fields, CIGAR, SEQ, QUAL 0xff (= absent; with
    qual_seed: random Phred values, normal(18, 8) clipped to 1..50 - a file that deflates like one with base qualities), and an
    SA tag rebuilt from the segment rows of every primary that has them.  Returns (n_records, uncompressed bytes)."""
    from concurrent.futures import ThreadPoolExecutor
    A = hb.arrays
    n = hb.n_rec
    cig_off = A["cigar_off"].astype(np.int64)
    seq_off = A["seq_off"].astype(np.int64)
    lseq = A["lseq"].astype(np.int64)
    flag = (A["flag"].astype(np.int64) & 0x0fff)
    seg_off = A["seg_off"].astype(np.int64)
    scig_off = A["seg_cigar_off"].astype(np.int64)
    # SA strings of the records that own segment rows
    sa = {}
    for i in np.nonzero(seg_off[1:n + 1] > seg_off[:n])[0].tolist():
        parts = []
        for r in range(int(seg_off[i]), int(seg_off[i + 1])):
            words = A["seg_cigar"][int(scig_off[r]):int(scig_off[r + 1])]
            cg = "".join("%d%s" % (int(w) >> 4, _CIG[int(w) & 15]) for w in words)
            parts.append("%s,%d,%s,%s,%d,0" % (references[int(A["seg_tid"][r])], int(A["seg_pos"][r]) + 1, "-" if A["seg_rev"][r] else "+", cg,
                                               int(A["seg_mapq"][r])))
        sa[i] = ("SAZ" + ";".join(parts) + ";").encode("ascii") + b"\0"
    name_len = len((name_fmt % 0).encode()) + 1
    n_cig = (cig_off[1:n + 1] - cig_off[:n])
    if n and int(n_cig.max()) > 65535:
        raise ValueError("write_bam_from_batch: CIGARs beyond 65535 operations need the CG tag (svim_amd.records.write_bam handles them)")
    sa_len = np.zeros(n, dtype=np.int64)
    for i, s in sa.items():
        sa_len[i] = len(s)
    rec_len = 32 + name_len + 4 * n_cig + (lseq + 1) // 2 + lseq + sa_len
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(rec_len + 4, out=off[1:])
    head_text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (r, l) for r, l in zip(references, lengths))).encode()
    head = b"BAM\1" + len(head_text).to_bytes(4, "little") + head_text + len(references).to_bytes(4, "little")
    for r, l in zip(references, lengths):
        nm = r.encode() + b"\0"
        head += len(nm).to_bytes(4, "little") + nm + int(l).to_bytes(4, "little")
    if qual_seed is None:
        buf = np.full(len(head) + int(off[n]), 0xff, dtype=np.uint8)       # QUAL bytes are 0xff: pre-filled
    else:                                                                 # (everything but QUAL is overwritten below)
        buf = np.clip(np.rint(np.random.default_rng(qual_seed).normal(18.0, 8.0, len(head) + int(off[n]))), 1, 50).astype(np.uint8)
    buf[:len(head)] = np.frombuffer(head, dtype=np.uint8)
    base = len(head)
    core = np.zeros((n, 9), dtype="<i4")                                  # block_size, refID, pos, (l_name|mapq|bin), (n_cig|flag), l_seq, next refID, next pos, tlen
    core[:, 0] = rec_len
    core[:, 1] = A["tid"][:n]
    core[:, 2] = A["pos"][:n]
    core[:, 3] = name_len | (A["mapq"][:n].astype(np.int64) << 8) | (4680 << 16)
    core[:, 4] = n_cig | (flag[:n] << 16)
    core[:, 5] = lseq[:n]
    core[:, 6] = -1
    core[:, 7] = -1
    core_b = core.view(np.uint8).reshape(n, 36)
    read_id = A["read_id"]
    cig_b = A["cigar"].view(np.uint8)
    seq_b = A["seq"]
    for i in range(n):
        o = base + int(off[i])
        buf[o:o + 36] = core_b[i]
        o += 36
        nm = (name_fmt % int(read_id[i])).encode() + b"\0"
        buf[o:o + name_len] = np.frombuffer(nm, dtype=np.uint8)
        o += name_len
        c0, c1 = 4 * int(cig_off[i]), 4 * int(cig_off[i + 1])
        buf[o:o + c1 - c0] = cig_b[c0:c1]
        o += c1 - c0
        nb = (int(lseq[i]) + 1) // 2
        s0 = int(seq_off[i])
        buf[o:o + nb] = seq_b[s0:s0 + nb]
        o += nb + int(lseq[i])
        s = sa.get(i)
        if s is not None:
            buf[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    raw = buf.tobytes()
    step = 65280
    chunks = [raw[i:i + step] for i in range(0, len(raw), step)]
    with ThreadPoolExecutor(max_workers=threads or min(32, os.cpu_count() or 1)) as ex:
        blocks = list(ex.map(_bgzf_block, chunks))
    with open(path, "wb") as fh:
        for b in blocks:
            fh.write(b)
        fh.write(_bgzf_block(b""))                                        # EOF marker block
    return n, len(raw)


def write_bam_from_device_batch(path, batch, references, lengths, lo=0, hi=None, name_prefix="r", name_digits=8, threads=None, qual_seed=None,
                                slab_bytes=384 << 20, level=1, index=False):
    """The same file write_bam_from_batch makes (fixed fields, CIGAR, SEQ, QUAL 0xff or - with qual_seed - random Phred values, SA tags from the
    segment rows), for records [lo, hi) of a DeviceBatch whose arrays live on the GPU: the uncompressed stream is put together THERE slab by slab
    (ragged gathers over torch tensors), compressed by the host's threads, and written as it comes - a million records (27 GB of stream) need neither
    a per-record Python loop nor the whole stream in host memory.  index: also <path>.bai (records.write_bai).  Bench / test infrastructure.
    Returns (n_records, uncompressed bytes)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    t = batch.t
    dev = t["cigar"].device
    hi = batch.n_rec if hi is None else hi
    n = hi - lo
    i64 = torch.int64
    cig_off = t["cigar_off"][lo:hi + 1].to(i64)
    seq_off = t["seq_off"][lo:hi + 1].to(i64)
    lseq = t["lseq"][lo:hi].to(i64)
    n_cig = cig_off[1:] - cig_off[:-1]
    if n and int(n_cig.max().item()) > 65535:
        raise ValueError("write_bam_from_device_batch: CIGARs beyond 65535 operations need the CG tag (svim_amd.records.write_bam handles them)")
    # SA strings of the records that own segment rows (host loop over those records only)
    seg_off = t["seg_off"][lo:hi + 1].cpu().numpy().astype(np.int64)
    owners = np.nonzero(seg_off[1:] > seg_off[:-1])[0]
    sa_len = np.zeros(n, dtype=np.int64)
    sa_parts = []
    if owners.size:
        s0, s1 = int(seg_off[0]), int(seg_off[-1])
        st, sp, sr, sm = (t[k][s0:s1].cpu().numpy() for k in ("seg_tid", "seg_pos", "seg_rev", "seg_mapq"))
        sco = t["seg_cigar_off"][s0:s1 + 1].cpu().numpy().astype(np.int64)
        scg = t["seg_cigar"][int(sco[0]):int(sco[-1])].cpu().numpy()
        sco = sco - sco[0]
        for i in owners.tolist():
            parts = []
            for r in range(int(seg_off[i]) - s0, int(seg_off[i + 1]) - s0):
                words = scg[int(sco[r]):int(sco[r + 1])]
                cg = "".join("%d%s" % (int(w) >> 4, _CIG[int(w) & 15]) for w in words)
                parts.append("%s,%d,%s,%s,%d,0" % (references[int(st[r])], int(sp[r]) + 1, "-" if sr[r] else "+", cg, int(sm[r])))
            b = ("SAZ" + ";".join(parts) + ";").encode("ascii") + b"\0"
            sa_parts.append(b)
            sa_len[i] = len(b)
    sa_blob = torch.frombuffer(bytearray(b"".join(sa_parts) or b"\0"), dtype=torch.uint8).to(dev)
    sa_len_t = torch.as_tensor(sa_len, device=dev)
    sa_src = torch.cumsum(sa_len_t, 0) - sa_len_t
    name_len = len(name_prefix) + name_digits + 1
    nb = (lseq + 1) // 2
    rec_len = 32 + name_len + 4 * n_cig + nb + lseq + sa_len_t
    head_text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (r, l) for r, l in zip(references, lengths))).encode()
    head = b"BAM\1" + len(head_text).to_bytes(4, "little") + head_text + len(references).to_bytes(4, "little")
    for r, l in zip(references, lengths):
        nm = r.encode() + b"\0"
        head += len(nm).to_bytes(4, "little") + nm + int(l).to_bytes(4, "little")
    cig_bytes = t["cigar"].view(torch.uint8)
    seq_bytes = t["seq"]
    flag = t["flag"][lo:hi].to(i64) & 0x0fff
    gen = None
    if qual_seed is not None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(qual_seed))
    prefix = torch.tensor(list(name_prefix.encode()), dtype=torch.uint8, device=dev)
    pow10 = torch.tensor([10 ** k for k in range(name_digits - 1, -1, -1)], dtype=i64, device=dev)

    def ragged(dst, dst_start, src, src_start, lens):
        tot = int(lens.sum().item())
        if tot == 0:
            return
        rep = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)
        within = torch.arange(tot, device=dev) - (torch.cumsum(lens, 0) - lens)[rep]
        dst[dst_start[rep] + within] = src[src_start[rep] + within]

    # slabs of about slab_bytes of stream
    rl_host = (rec_len + 4).cpu().numpy()
    bounds = [0]
    acc = 0
    for i, l in enumerate(rl_host.tolist()):
        acc += l
        if acc >= slab_bytes:
            bounds.append(i + 1)
            acc = 0
    if bounds[-1] != n:
        bounds.append(n)
    step = 65280
    total_raw = len(head)
    pending = head
    pool = ThreadPoolExecutor(max_workers=threads or effective_cpus())
    block_at = []                                                         # file offset of every BGZF block (for the index)
    with open(path, "wb") as fh:
        def flush(data, final):
            k = len(data) if final else (len(data) // step) * step
            chunks = [data[i:i + step] for i in range(0, k, step)]
            for b in pool.map(lambda c: _bgzf_block(c, level), chunks):
                block_at.append(fh.tell())
                fh.write(b)
            return data[k:]
        for a, b in zip(bounds[:-1], bounds[1:]):
            m = b - a
            rl = rec_len[a:b]
            off = torch.cumsum(rl + 4, 0) - (rl + 4)
            tot = int((rl + 4).sum().item())
            if gen is None:
                buf = torch.full((tot,), 0xff, dtype=torch.uint8, device=dev)
            else:
                buf = torch.normal(18.0, 8.0, (tot,), generator=gen, device=dev).round_().clamp_(1, 50).to(torch.uint8)
            core = torch.zeros((m, 9), dtype=torch.int32, device=dev)
            core[:, 0] = rl.to(torch.int32)
            core[:, 1] = t["tid"][lo + a:lo + b]
            core[:, 2] = t["pos"][lo + a:lo + b]
            core[:, 3] = (name_len | (t["mapq"][lo + a:lo + b].to(i64) << 8) | (4680 << 16)).to(torch.int32)
            core[:, 4] = (n_cig[a:b] | (flag[a:b] << 16)).to(torch.int32)
            core[:, 5] = lseq[a:b].to(torch.int32)
            core[:, 6] = -1
            core[:, 7] = -1
            ar36 = torch.arange(36, device=dev)
            buf[(off[:, None] + ar36[None, :]).reshape(-1)] = core.view(torch.uint8).reshape(-1)
            ids = t["read_id"][lo + a:lo + b].to(i64)
            digits = ((ids[:, None] // pow10[None, :]) % 10 + 48).to(torch.uint8)
            names = torch.cat([prefix[None, :].expand(m, -1), digits, torch.zeros((m, 1), dtype=torch.uint8, device=dev)], dim=1)
            buf[((off + 36)[:, None] + torch.arange(name_len, device=dev)[None, :]).reshape(-1)] = names.reshape(-1)
            at = off + 36 + name_len
            ragged(buf, at, cig_bytes, 4 * cig_off[a:b], 4 * n_cig[a:b])
            at = at + 4 * n_cig[a:b]
            ragged(buf, at, seq_bytes, seq_off[a:b], nb[a:b])
            at = at + nb[a:b] + lseq[a:b]
            ragged(buf, at, sa_blob, sa_src[a:b], sa_len_t[a:b])
            data = pending + buf.cpu().numpy().tobytes()
            total_raw += tot
            del buf
            pending = flush(data, False)
        flush(pending, True)
        block_at.append(fh.tell())
        fh.write(_bgzf_block(b""))                                        # EOF marker block
    pool.shutdown()
    if index:
        from .records import write_bai
        starts = len(head) + np.concatenate([[0], np.cumsum(rl_host)[:-1]]) if n else np.zeros(0, dtype=np.int64)
        tids = t["tid"][lo:hi].cpu().numpy()
        write_bai(path + ".bai", len(references), list(zip(tids.tolist(), starts.tolist())), total_raw, block_at, step)
    return n, total_raw


# ---------------------------------------------------------------------------------------------------------------------
# measurements
# ---------------------------------------------------------------------------------------------------------------------
def _timed_bam_passes(path, opts, eng, genome, passes=1, threads=0, batch_records=200_000, sparse_seq=True, gpu_inflate=None, device_decode=None):
    """-> list of (records, wall seconds, pipeline stats, engine stats, (n_sig, n_seq, n_bnd)) per pass over ONE reader: the first pass
    pays every first-touch allocation (host buffers, device buffers), later passes are the steady state of a long file."""
    pipe = BamPipeline(path, opts, eng, threads=threads, batch_records=batch_records, sparse_seq=sparse_seq, gpu_inflate=gpu_inflate, device_decode=device_decode)
    out = []
    try:
        for k in range(passes):
            t0 = time.perf_counter()
            if k:
                pipe.rewind()                       # INSIDE the clock: rewinding inflates the first chunk of the file (svx_bam_rewind -> ensure)
            n = pipe.run()
            pipe.cluster(genome if k == 0 else None)
            wall = time.perf_counter() - t0
            out.append((n, wall, dict(pipe.stats), eng.stats(), eng.collect_counts()))
    finally:
        pipe.close()
    return out


def _median_pass(runs):
    """(cold pass, median of the warm passes) of _timed_bam_passes"""
    warm = sorted(runs[1:], key=lambda r: r[1])
    return runs[0], warm[len(warm) // 2]


def _inflate_delta(runs, k):
    """inflate counters of pass k alone (the reader's counters are cumulative over its passes)"""
    cur = runs[k][2].get("inflate") or {}
    prev = (runs[k - 1][2].get("inflate") or {}) if k else {}
    return {key: cur.get(key, 0) - prev.get(key, 0) for key in ("gpu_blocks", "cpu_blocks", "gpu_kernel_ms")}


def end_to_end_sample(batch, g_off, genome, opts, device=0, resident_reads_per_s=None, n_records=180_000, n_records_qual=300_000, tmp_dir=None,
                      chunk_mb=2048):
    """bench.py's `end_to_end` block.  HEADLINE: the first n_records_qual records of the synthetic batch (coordinate order) written as a BAM file WITH base
    qualities (what every real ONT / HiFi file carries; random Phred values, deflate ~1.5 : 1), read by the device-resident reader in chunks of chunk_mb of
    inflated data (several chunks: the overlap of chunk i+1's inflate with chunk i's batches is inside the clock), COLLECT per batch, CLUSTER - wall clock
    from before rewind() to the end of CLUSTER, median of three warm passes, the cold first pass beside it, and the cost of turning the result tables into
    the reference's Python objects.  Side figures: the same records without qualities (QUAL 0xff, ~4 : 1: the round-3 headline), the host reader's routes
    on that file, host arrays in, the resident rate."""
    import tempfile
    import torch
    from . import lazy
    eng = _lib.Engine(device)
    n = min(int(n_records), batch.n_rec)
    nq = min(int(n_records_qual), batch.n_rec)
    hb = batch.slice_records(0, n)
    refs = list(hb.references)
    lens = [int(x) for x in (g_off[1:] - g_off[:-1]).tolist()]
    d = tempfile.mkdtemp(prefix="svx_e2e_", dir=tmp_dir)
    path, qpath = os.path.join(d, "sample.bam"), os.path.join(d, "sample_q.bam")
    gen = (g_off, genome, True)
    p = _abi.Params.from_options(opts)
    out = {"host_cores_visible": os.cpu_count(), "host_cpus_granted": effective_cpus()}
    old_chunk = os.environ.get("SVX_BAM_DEV_CHUNK_MB")
    try:
        # ---- headline: the file with base qualities ------------------------------------------------------------------------------------------------------
        t0 = time.perf_counter()
        _, q_raw = write_bam_from_device_batch(qpath, batch, refs, lens, 0, nq, qual_seed=7)
        t_write_q = time.perf_counter() - t0
        q_size = os.path.getsize(qpath)
        os.environ["SVX_BAM_DEV_CHUNK_MB"] = str(int(chunk_mb))
        per_batch = max(1000, nq // 16)
        runs = _timed_bam_passes(qpath, opts, eng, gen, passes=4, batch_records=per_batch)
        cold, med = _median_pass(runs)
        k = runs.index(med)
        inf = _inflate_delta(runs, k)
        n_read, wall, ps, st, counts = med
        gpu_share = inf["gpu_blocks"] / max(1, inf["gpu_blocks"] + inf["cpu_blocks"])
        gpu_rate = gpu_share * q_raw / max(inf["gpu_kernel_ms"] * 1e-3, 1e-9) / 1e6                  # MB/s of inflated output while the kernels run
        n_chunks = -(-q_raw // (int(chunk_mb) << 20))
        out["sample"] = ("first %d records of the batch as a BAM file WITH base qualities (%.0f MB, %.0f MB inflated, deflate ratio %.2f; written in %.1f s, untimed); "
                         "read in %d chunks of %d MB" % (nq, q_size / 1e6, q_raw / 1e6, q_raw / max(1, q_size), t_write_q, n_chunks, chunk_mb))
        out["bam_file_reads_per_s"] = n_read / wall
        out["bam_file_first_pass_reads_per_s"] = cold[0] / cold[1]          # cold: every host / device buffer is touched for the first time, the file is registered
        blk = {"records": n_read, "reads_per_s": n_read / wall, "wall_s": wall, "passes": "1 cold + 3 warm; the figure is the MEDIAN warm pass",
               "warm_pass_walls_s": sorted(r[1] for r in runs[1:]), "cold_pass_wall_s": cold[1], "chunks": int(n_chunks), "chunk_MB": int(chunk_mb),
               "bam_MB": q_size / 1e6, "inflated_MB": q_raw / 1e6, "deflate_ratio": q_raw / max(1, q_size), "bam_MB_per_s": q_size / wall / 1e6,
               "inflated_MB_per_s": q_raw / wall / 1e6, "batches": ps["batches"], "reader_busy_s": ps["t_reader_busy"], "gpu_collect_s": ps["t_gpu_collect"],
               "gpu_waits_for_reader_s": ps["t_gpu_waits_for_reader"], "cluster_s": ps["t_cluster_wall"], "signatures": counts[0],
               "qualities": "random Phred values, normal(18, 8) clipped to 1..50",
               "reader": "device-resident: BGZF inflate, record discovery, field / CIGAR / SA / name decode on the GPU (csrc/bamdev.hip)",
               "inflate_blocks_gpu": inf["gpu_blocks"], "inflate_blocks_host_cores": inf["cpu_blocks"], "inflate_kernel_ms": inf["gpu_kernel_ms"],
               "inflate_kernel_MB_per_s": gpu_rate, "clock": "perf_counter from before rewind() to the end of CLUSTER (first chunk included)",
               "bound_by": ("GPU (the k_bgzf_inflate launches sum to %.0f %% of the wall time - the launches of the inflater's three slots overlap since the staged input path of round 5, "
                            "so the sum can exceed 100 %%; COLLECT + CLUSTER %.0f %%)" % (100 * inf["gpu_kernel_ms"] * 1e-3 / wall, 100 * (ps["t_gpu_collect"] + ps["t_cluster_wall"]) / wall))
               if inf["gpu_kernel_ms"] * 1e-3 > 0.5 * wall else "host / PCIe (the GPU inflate is busy %.0f %% of the wall time)" % (100 * inf["gpu_kernel_ms"] * 1e-3 / wall)}
        # what it costs to LOOK at the result: the signature table and the six cluster lists as the reference's Python objects (svim_amd/lazy.py builds them on
        # first access; SVIM's own writers iterate them once)
        pipe = BamPipeline(qpath, opts, eng, batch_records=per_batch)
        try:
            pipe.run()
            pipe.cluster()
            names = pipe.bam.read_names()
            t0 = time.perf_counter()
            sl = lazy.SignatureList(eng.fetch_signatures(0), refs, names)
            bl = lazy.SignatureList(eng.fetch_signatures(1), refs, names)
            sl.materialise(); bl.materialise()
            lists = convert.cluster_objects(eng.fetch_clusters(), sl, refs)
            n_obj = len(sl) + len(bl) + sum(len(l.materialise()) for l in lists)
            t_obj = time.perf_counter() - t0
        finally:
            pipe.close()
        blk["objects_materialised"] = {"objects": n_obj, "seconds": t_obj, "reads_per_s_including_them": n_read / (wall + t_obj)}
        out["objects_materialised_reads_per_s"] = n_read / (wall + t_obj)
        out["bam_file_with_base_qualities"] = blk
        out["bam_file"] = blk
        os.remove(qpath)
        # ---- side figure: the same records without qualities (QUAL 0xff), read in chunks of the same size: the cold pass touches all three chunk slots of the
        # reader, so that the warm passes are warm (the slots rotate across rewind, csrc/bamio.cpp dev_last_slot) -----------------------------------------------
        t0 = time.perf_counter()
        _, raw_bytes = write_bam_from_device_batch(path, batch, refs, lens, 0, n)
        size = os.path.getsize(path)
        per_batch = max(1000, n // 6)
        runs = _timed_bam_passes(path, opts, eng, gen, passes=4, batch_records=per_batch)
        cold0, med0 = _median_pass(runs)
        inf0 = _inflate_delta(runs, runs.index(med0))
        share0 = inf0["gpu_blocks"] / max(1, inf0["gpu_blocks"] + inf0["cpu_blocks"])
        rate0 = share0 * raw_bytes / max(inf0["gpu_kernel_ms"] * 1e-3, 1e-9) / 1e6
        out["bam_file_without_base_qualities"] = {"records": med0[0], "reads_per_s": med0[0] / med0[1], "first_pass_reads_per_s": cold0[0] / cold0[1], "bam_MB": size / 1e6,
                                                  "inflated_MB": raw_bytes / 1e6, "deflate_ratio": raw_bytes / max(1, size), "inflated_MB_per_s": raw_bytes / med0[1] / 1e6,
                                                  "inflate_kernel_MB_per_s": rate0, "chunk_MB": int(chunk_mb), "note": "QUAL 0xff (absent): the round-3 headline file"}
        # the same file with the host reader (inflate shared GPU / host cores, record decode on the host's cores) and with the host's cores alone
        r = _median_pass(_timed_bam_passes(path, opts, eng, gen, passes=3, batch_records=per_batch, device_decode=False))[1]
        out["bam_file_host_decode_reads_per_s"] = r[0] / r[1]
        r = _median_pass(_timed_bam_passes(path, opts, eng, gen, passes=3, batch_records=per_batch, gpu_inflate=False, device_decode=False))[1]
        out["bam_file_host_inflate_only_reads_per_s"] = r[0] / r[1]
        host_only_rate = raw_bytes / r[1] / 1e6
        # sanity of the clock: a pass cannot be shorter than a third of the summed durations of its own inflate launches (they run on the inflater's
        # THREE slot streams and their tails overlap: 90 ms of launches in an 83 ms pass on the round-5 box), and the blocks
        # the host's cores took (the default input path stages the file through pinned buffers and lets the cores inflate from the back of every chunk) cannot exceed
        # what zlib delivers on the granted CPUs (1.5 GB/s of output per CPU is generous: 0.8 measured).  (Until round 4 the bound was "kernel rate + the host READER's
        # rate"; the host reader also decodes records, so that sum undercounts what cores that only inflate can add.)
        host_share_MBps = (1.0 - share0) * raw_bytes / med0[1] / 1e6
        sane = {"inflated_MB_per_s": raw_bytes / med0[1] / 1e6, "gpu_kernel_MB_per_s_while_running": rate0, "gpu_share_of_blocks": share0,
                "pass_seconds": med0[1], "gpu_inflate_launch_seconds_summed_in_that_pass": inf0["gpu_kernel_ms"] * 1e-3,
                "host_share_MB_per_s": host_share_MBps, "host_share_bound_MB_per_s": 1500.0 * effective_cpus(), "host_only_reader_MB_per_s": host_only_rate,
                "ok": med0[1] >= 0.98 * inf0["gpu_kernel_ms"] * 1e-3 / 3.0 and host_share_MBps <= 1500.0 * effective_cpus(), "file": "the one without qualities"}
        out["bam_file_without_base_qualities"]["sanity"] = sane
        assert sane["ok"], sane
        # (b) host arrays in, no file: pageable numpy arrays (every byte takes the library's bounce buffers), and the same arrays in page-locked memory the
        # library owns (svx_host_alloc: uploaded in place) - what a batcher that fills such arrays gets
        eng.accumulate(False)
        eng.set_genome(*gen)
        for key, batch in (("host_arrays_reads_per_s", hb), ("host_arrays_library_pinned_reads_per_s", hb.pinned())):
            eng.collect(batch, p, fetch=False)
            eng.cluster(p, batch.contig_rank, source=0, fetch=False)
            t0 = time.perf_counter()
            for _ in range(2):
                eng.collect(batch, p, fetch=False)
                eng.cluster(p, batch.contig_rank, source=0, fetch=False)
            torch.cuda.synchronize()
            out[key] = 2 * n / (time.perf_counter() - t0)
        out["host_arrays_signatures"] = eng.collect_counts()[0]
        out["resident_reads_per_s"] = resident_reads_per_s
    finally:
        if old_chunk is None:
            os.environ.pop("SVX_BAM_DEV_CHUNK_MB", None)
        else:
            os.environ["SVX_BAM_DEV_CHUNK_MB"] = old_chunk
        for f in (path, qpath):
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass
        eng.close()
    return out


def shard_plan(references, lengths, bai, rank, world):
    """Contig ownership (multigpu.assign_contigs) and the file regions this rank reads: maximal runs of consecutive reference ids that are
    its own (contigs without records do not interrupt a run) -> (owner[n_ref], [(virtual offset of the run's first record, last reference id,
    first reference id)])."""
    from .multigpu import assign_contigs
    owner = assign_contigs(references, lengths, world)
    n = len(references)
    runs, t = [], 0
    while t < n:
        if owner[t] == rank and bai[t] is not None:
            a = b = t
            while b + 1 < n and (owner[b + 1] == rank or bai[b + 1] is None):
                b += 1
            runs.append((bai[a][0], b, a))
            t = b + 1
        else:
            t += 1
    return owner, runs


def collect_cluster_bam_sharded(bam_path, opts, engine, adapter, rank, world, threads=0, batch_records=200_000, bai_path=None, gather_names=True):
    """One rank of a contig-sharded run over an indexed, coordinate-sorted BAM: read this rank's contig runs (svx_bam_seek), COLLECT
    them batch by batch (accumulating), exchange the region sizes so that emission slots are global, then multigpu.cluster_step.
    Returns (StepResult or None, pipeline, read-name list of this rank)."""
    import torch.distributed as dist
    from . import multigpu, records
    from .batch import contig_ranks
    from .bamio import NativeBam
    probe = NativeBam(bam_path, threads=1)
    refs, lens = probe.references, probe.lengths
    probe.close()
    bai = records.read_bai(bai_path or bam_path + ".bai")
    owner, runs = shard_plan(refs, lens, bai, rank, world)
    pipe = BamPipeline(bam_path, opts, engine, threads=threads, batch_records=batch_records, regions=[(v, last) for v, last, _ in runs])
    n_rec = pipe.run()
    # emission slots in file order: every region's local slot range moves to the sum of the spans of the regions before it in the file
    ends = [s for s, _ in pipe.region_slots[1:]] + [pipe.region_slots[-1][0] + 2 * pipe.region_slots[-1][1] + 2 * pipe.stats["batches"] + 2] if pipe.region_slots else []
    local = [(runs[k][2], pipe.region_slots[k][0], ends[k] - pipe.region_slots[k][0]) for k in range(len(pipe.region_slots))]   # (first tid, local start, span)
    everyone = [None] * world
    dist.all_gather_object(everyone, local)
    order = sorted((first_tid, r, k, span) for r, regs in enumerate(everyone) for k, (first_tid, _, span) in enumerate(regs))
    base, acc = {}, 0
    for first_tid, r, k, span in order:
        base[(r, k)] = acc
        acc += span
    key_runs = ([st for _, st, _ in local], [base[(rank, k)] for k in range(len(local))]) if local else ([0], [0])
    names = pipe.bam.read_names()
    index = None

    def names_of(ids):
        return [names[int(i)] for i in ids]

    def ids_of(nms):
        nonlocal index
        if index is None:
            index = {nm: i for i, nm in enumerate(names)}
        out = []
        for nm in nms:
            i = index.get(nm)
            if i is None:
                i = index[nm] = len(names)
                names.append(nm)
            out.append(i)
        return out
    # read ids are rank-local numbers (each rank interns the names of the records IT read, plus the names foreign rows arrive with): in the gathered
    # table rank r's ids are shifted by r * stride, and rank 0 receives every rank's name list - StepResult.read_name(id) resolves any member
    stride = (2 ** 31 - 1) // max(1, world)
    if len(names) >= stride:
        raise OverflowError("more than %d read names on one rank" % stride)
    res = multigpu.cluster_step(adapter, pipe.params, rank, world, np.arange(len(refs)), contig_ranks(refs), owner, key_runs=key_runs,
                                names_of=names_of, ids_of=ids_of, read_base=rank * stride)
    if gather_names and world > 1:
        everyone = [None] * world if rank == 0 else None
        dist.gather_object(list(names), everyone, dst=0)
        if rank == 0:
            res.set_read_names(everyone, stride)
    elif res is not None:
        res.set_read_names([list(names)], stride)
    pipe.stats["records"] = n_rec
    return res, pipe, names


def run_bam(bam_path, fasta_path, opts, rank=0, world=1, device=0, steps=1, warmup=0, threads=0):
    """bench.py --bam PATH --fasta PATH: read / op / signature / cluster counts and the end-to-end rates of a real
    coordinate-sorted BAM on ONE GPU (contig-sharded multi-GPU reading needs the .bai route)."""
    if not fasta_path:
        raise SystemExit("bench.py --bam needs --fasta (the reference genome the insertion haplotypes are built from)")
    eng = _lib.Engine(device)
    from .bamio import NativeBam
    nb = NativeBam(bam_path, threads=1)
    refs, lens, so = nb.references, nb.lengths, nb.sort_order
    nb.close()
    if so != "coordinate":
        raise SystemExit("bench.py --bam: coordinate-sorted input expected (header says %r)" % so)
    t0 = time.perf_counter()
    off, codes = convert.genome_arrays(fasta_path, refs)
    t_genome = time.perf_counter() - t0
    eng.set_genome(off, codes)
    if world > 1:
        return _run_bam_sharded(bam_path, fasta_path, opts, eng, rank, world, device, steps, warmup, threads, refs, t_genome)
    runs = _timed_bam_passes(bam_path, opts, eng, None, passes=warmup + steps, threads=threads)
    n, wall, ps, st, counts = min(runs[warmup:], key=lambda r: r[1])
    size = os.path.getsize(bam_path)
    ct = eng.fetch_clusters()
    return {"metric": "aligned reads/sec through COLLECT+CLUSTER", "value": n / wall, "unit": "reads/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * wall, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "file",
            "dtype": "u32/i32 CIGAR + u8 bases, int64 positions, f64 distances",
            "config": {"workload": "BAM file %s (%d contigs, %.1f MB) + FASTA %s, END TO END: BGZF inflate + decode + H2D + COLLECT + CLUSTER" % (
                os.path.basename(bam_path), len(refs), size / 1e6, os.path.basename(fasta_path)), "options": "SVIM alignment-mode defaults"},
            "counts": {"records": n, "reads_used": st["n_rec_used"], "signatures": counts[0], "clusters": ct.n,
                       "clusters_by_type": dict(zip(_abi.TYPE_NAMES, [int(x) for x in ct.type_count])), "partitions": st["n_partitions"],
                       "large_partitions": st["n_large_partitions"], "edit_pairs": st["n_edit_pairs"]},
            "end_to_end": {"bam_file_reads_per_s": n / wall, "bam_MB_per_s": size / wall / 1e6, "reader_busy_s": ps["t_reader_busy"],
                           "gpu_collect_s": ps["t_gpu_collect"], "gpu_waits_for_reader_s": ps["t_gpu_waits_for_reader"], "cluster_s": ps["t_cluster_wall"],
                           "genome_load_s": t_genome, "host_cores_visible": os.cpu_count(), "host_cpus_granted": effective_cpus()}}


def _run_bam_sharded(bam_path, fasta_path, opts, eng, rank, world, device, steps, warmup, threads, refs, t_genome):
    """bench.py --gpus N --bam: contig-sharded ranks over one indexed BAM (needs <bam>.bai); value = records of ALL ranks / max wall."""
    import torch
    import torch.distributed as dist
    from . import multigpu
    dev = "cuda:%d" % device
    adapter = multigpu.SvxAdapter(eng, dev)
    best = None
    for it in range(warmup + steps):
        torch.cuda.synchronize(); multigpu.barrier()
        t0 = time.perf_counter()
        res, pipe, _ = collect_cluster_bam_sharded(bam_path, opts, eng, adapter, rank, world, threads=threads)
        torch.cuda.synchronize(); multigpu.barrier()
        wall = time.perf_counter() - t0
        n_local = pipe.stats["records"]
        pipe.close()
        tt = torch.tensor([wall, float(n_local)], dtype=torch.float64, device=dev)
        walls = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(walls, tt)
        wall = max(float(w[0]) for w in walls)
        n_all = int(sum(float(w[1]) for w in walls))
        if it >= warmup and (best is None or wall < best[0]):
            best = (wall, n_all, res)
    if rank != 0:
        return None
    wall, n_all, res = best
    ct = res.to_host()
    return {"metric": "aligned reads/sec through COLLECT+CLUSTER", "value": n_all / wall, "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * wall, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "file",
            "dtype": "u32/i32 CIGAR + u8 bases, int64 positions, f64 distances",
            "config": {"workload": "BAM file %s (%d contigs) + FASTA %s, END TO END on %d contig-sharded ranks (.bai seek per contig run)" % (
                os.path.basename(bam_path), len(refs), os.path.basename(fasta_path), world), "options": "SVIM alignment-mode defaults"},
            "counts": {"records": n_all, "signatures": int(sum(res.sig_counts)), "clusters": ct.n,
                       "clusters_by_type": dict(zip(_abi.TYPE_NAMES, [int(x) for x in ct.type_count]))},
            "end_to_end": {"bam_file_reads_per_s": n_all / wall, "genome_load_s": t_genome, "host_cores_visible": os.cpu_count(), "host_cpus_granted": effective_cpus()}}
