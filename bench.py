#!/usr/bin/env python3
"""bench.py - COLLECT+CLUSTER throughput on MI355X (BASELINE.json metric: aligned reads/sec through
COLLECT+CLUSTER; SV signatures/sec clustered).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (svx_collect + svx_cluster through the C ABI) over one synthetic record batch
that is already resident in HBM: BASELINE.json configs[1] - 1M synthetic ONT reads (N50 20 kb), one 250 Mb contig,
planted DEL/INS/INV (svim_amd/devsynth.py).  Weak scaling: every rank owns its own batch of that size (its own
contig); signatures are all-gathered over RCCL so that partitioning and the sequential random.sample stream see
the whole list, the quadratic per-partition work is sharded by partition index, and the cluster tables are gathered
at the end (svim_amd/distributed.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def options():
    return types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10,
                                 segment_overlap_tolerance=5, partition_max_distance=1000, position_distance_normalizer=900,
                                 edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)


def cpu_baseline(batch, genome, params, eng=None, budget_s=20.0):
    """The oracle (single-threaded C restatement of the reference algorithm, kind 'port') on a bounded, contiguous
    slice of the same batch (contiguous in coordinate order = full local coverage, so per-partition work is
    representative)."""
    from oracle import oracle as om
    orc = om.Oracle()
    g = genome.cpu().numpy()
    orc.set_genome(np.array([0, g.size], dtype=np.int64), g)
    n = min(batch.n_rec, 4000)
    best = None
    for _ in range(4):
        hb = batch.slice_records(0, n)
        t0 = time.perf_counter()
        sig, _ = orc.collect(hb, params)
        t1 = time.perf_counter()
        ct = orc.cluster(params, np.zeros(1, np.int32), source=0)
        t2 = time.perf_counter()
        st = orc.stats()
        best = dict(n_rec=n, used=st["n_rec_used"], n_sig=sig.n, t_collect=t1 - t0, t_cluster=t2 - t1, n_clusters=ct.n,
                    edit_cells=st["n_edit_cells"], ops=st["n_ops"])
        if (t2 - t0) >= budget_s / 4 or n >= batch.n_rec:
            break
        n = min(batch.n_rec, n * 4)
    t = best["t_collect"] + best["t_cluster"]
    parity = None
    if eng is not None:
        # the same slice through the HIP path (host batch this time): the bench line carries its own parity evidence
        gs, _ = eng.collect(hb, params)
        gc = eng.cluster(params, np.zeros(1, np.int32), source=0)
        parity = {"records": best["n_rec"], "signatures_identical": gs.first_difference(sig) is None,
                  "clusters_identical": gc.first_difference(ct, rtol=1e-12) is None}
    return {"parity_vs_gpu_on_sample": parity, "value": best["used"] / t, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d records of the same batch (coordinate order): %d reads used, %d signatures, %d clusters; "
                      "collect %.2f s + cluster %.2f s on 1 host core (of %d)" % (
                          best["n_rec"], best["used"], best["n_sig"], best["n_clusters"], best["t_collect"], best["t_cluster"],
                          os.cpu_count()),
            "signatures_per_s": best["n_sig"] / max(best["t_cluster"], 1e-9),
            "cigar_ops_per_s": best["ops"] / max(best["t_collect"], 1e-9),
            "edit_cells_per_s": best["edit_cells"] / max(best["t_cluster"], 1e-9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--n50", type=int, default=20000)
    ap.add_argument("--contig-len", type=int, default=250_000_000)
    ap.add_argument("--sites", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    # SVX_BENCH_FORCE_DIST=1 exercises the multi-GPU code path (RCCL exchange + sharded clustering) with any world size
    use_dist = world > 1 or os.environ.get("SVX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from svim_amd import _abi, _lib, devsynth
    p = _abi.Params.from_options(options())
    t0 = time.perf_counter()
    batch, genome, meta = devsynth.make_batch(n_reads=args.reads, n50=args.n50, contig_len=args.contig_len, n_sites=args.sites,
                                              seed=2 + rank, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    eng = _lib.Engine(local_rank)
    if use_dist:
        from svim_amd import distributed as D
        g_off, g_all = D.all_gather_genomes(genome, dev)
        eng.set_genome(g_off, g_all, on_device=True)
    else:
        g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device=dev)
        eng.set_genome(g_off, genome, on_device=True)
    rank_arr = np.zeros(1, dtype=np.int32)
    bstruct = batch.struct()

    def step():
        eng.collect(bstruct, p, fetch=False)
        if use_dist:
            from svim_amd import distributed as D
            D.device_pipeline_step(eng, p, rank, world, dev)
        else:
            eng.cluster(p, rank_arr, source=0, fetch=False)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        s0 = time.perf_counter()
        step()
        per_step.append(time.perf_counter() - s0)
    barrier()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([st["n_rec_used"], st["n_sig"], st["n_ops"]], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_used, tot_sig, tot_ops = (int(x) for x in cnt.tolist())
    else:
        tot_used, tot_sig, tot_ops = st["n_rec_used"], st["n_sig"], st["n_ops"]
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    ms_per_step = 1e3 * elapsed / args.steps
    reads_per_s = tot_used * args.steps / elapsed
    # ---- roofline of the HBM-bound kernel (k_cigar_scan), algorithmic bytes per SURVEY.md section 8(d) ----
    n_ins = st["n_ins_bases"]
    bytes_collect = 32 * st["n_rec_used"] + 4 * st["n_ops"] + 16 * st["n_seg"] + 4 * st["n_seg_ops"] + 32 * st["n_sig"] + n_ins // 2
    scan_s = st["t_cigar_scan_ms"] * 1e-3
    scan_bytes = 32 * st["n_rec_used"] + 4 * st["n_ops"] + 16 * st["n_seg"] + 4 * st["n_seg_ops"] + 32 * st["n_sig"]
    achieved = scan_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    # HBM traffic of that kernel from the PMC counters (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction):
    # measured once for the default workload and committed under profiles/; null for any other workload
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "traffic_k_cigar_scan.json")) as fh:
            tj = json.load(fh)
        if tj.get("workload_cigar_ops") == meta["n_ops"]:
            traffic = tj["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        traffic = None
    roofline = {"kernel": "k_cigar_scan", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic,
                "algorithmic_bytes_per_launch": scan_bytes, "kernel_ms": st["t_cigar_scan_ms"]}
    kernels = {
        "k_cigar_scan_ms": st["t_cigar_scan_ms"], "k_segments_ms": st["t_segments_ms"], "collect_order_ms": st["t_sort_ms"],
        "collect_gather_ms": st["t_gather_ms"], "collect_total_ms": st["t_collect_ms"],
        "cluster_partition_sample_ms": st["t_partition_ms"], "cluster_edit_distance_ms": st["t_edit_ms"],
        "cluster_linkage_ms": st["t_linkage_ms"], "cluster_total_ms": st["t_cluster_ms"],
        "edit_distance_gcups": (st["n_edit_cells"] / (st["t_edit_ms"] * 1e-3) / 1e9) if st["t_edit_ms"] > 0 else None,
        "edit_pairs": st["n_edit_pairs"], "edit_cells": st["n_edit_cells"], "pair_distances": st["n_pairs"],
        "bytes_collect_model": bytes_collect,
    }
    out = {
        "metric": "aligned reads/sec through COLLECT+CLUSTER", "value": reads_per_s, "unit": "reads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32/i32 CIGAR + u8 bases, int64 positions, f64 distances", "data": "synthetic",
        "config": {"workload": "configs[1]: %d synthetic ONT reads per GPU (N50 %d), single %d Mb contig, DEL/INS/INV"
                               % (args.reads, args.n50, args.contig_len // 1_000_000),
                   "records_per_gpu": meta["n_records"], "cigar_ops_per_gpu": meta["n_ops"], "planted_sites": meta["n_sites"],
                   "parallelism": "1 process/GPU, records sharded, signature columns all-gathered, partitions owned by origin rank", "options": "SVIM alignment-mode defaults"},
        "signatures_per_s": tot_sig * args.steps / elapsed,
        "signatures_per_s_cluster_only": st["n_sig"] / (st["t_cluster_ms"] * 1e-3) if st["t_cluster_ms"] > 0 else None,
        "counts": {"reads_used": tot_used, "signatures": tot_sig, "cigar_ops": tot_ops, "partitions": st["n_partitions"],
                   "large_partitions": st["n_large_partitions"], "clusters": st["n_clusters"], "ins_bases": n_ins},
        "roofline": roofline, "kernels": kernels, "synth_seconds": t_gen,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch, genome, p, eng)
        out["speedup_vs_cpu_port"] = reads_per_s / world / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
