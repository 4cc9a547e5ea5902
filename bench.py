#!/usr/bin/env python3
"""bench.py — reads/s clustered on the BASELINE.json config-2 workload (3000 reads / 50 Mb, k=11 w=15,
fast mode) on N MI355X GPUs of one node.

One "step" = one pass of the hot path over one sorted batch whose minimizer SoA is already resident
in HBM: index build + shared-minimizer scoring + mapped-ratio resolve + decisions back on the host
(ioc_cluster_resident).  Batches shard one per GPU with no data-path collective (weak scaling); ranks
only meet at the timing barrier.  Inputs are synthetic (isonclust2_amd/synth.py, seed = 1 + rank) and
are prepared by the product's own GPU sort stage (ioc_qual_scores / ioc_extract_minimizers), never by
the oracle.  The oracle appears only in the cpu_baseline leg (rank 0, N=1): it is timed on the same
batch on one host core and doubles as a full-size parity check.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def prepare_resident_batch(ctx, api, synth, config, seed, k, w, mode="fast"):
    """raw reads -> GPU quality scores -> stable sort -> GPU HPC/minimizers -> resident queries."""
    rs = synth.generate_config(config, seed=seed)
    score, err = ctx.qual_scores(rs.offs, rs.qual, k)                       # FillQualScores
    order = np.argsort(-score, kind="stable")                               # SortByQualScores
    lens = np.diff(rs.offs)[order]
    so = np.zeros(rs.n + 1, np.int64)
    so[1:] = np.cumsum(lens)
    starts = rs.offs[:-1][order]
    idx = np.repeat(starts - so[:-1], lens) + np.arange(so[-1])
    ex = ctx.extract_minimizers(so, rs.seq[idx], rs.qual[idx], k, w)        # PrepareSortedBatch
    p = api.default_params(k, w, mode)
    ctx.set_params(p)
    # gates of the clustering loop (src/cluster.cpp:116-160); MinQual default 7.0
    keep = (ex["status"] == 0) & (score[order] >= 0) & (-10 * np.log10(err[order]) > 7.0)
    cell = np.array([api.host_err_cell(e) if kp else 1 for e, kp in zip(ex["hpc_err"], keep)], np.uint8)
    need = np.array([api.host_min_total(h, p.mapped_threshold) if kp else 0xFFFFFFFE
                     for h, kp in zip(ex["hpc_len"], keep)], np.uint32)
    ctx.queries_from_extracted(keep, cell, need)
    ctx.left_load(0, None, None, None, None)
    if mode == "sahlin":   # BASELINE.json configs[2]: the alignment fallback needs the raw sequences
        ctx.resident_set_sequences(rs.seq[idx], so, err[order])
    ex.update(score=score[order], raw_err=err[order])
    ctx._last_extract = ex
    return rs, order, int(ex["off_rev"][-1])


def resident_to_clustered(ctx, api, pipeline, rs, order, cls, strand, rank):
    """ClusteredBatch (representative records + membership + MinDB) of the batch just clustered."""
    import numpy as np
    n_min = ctx.timings()["n_minimizers"]
    mn, ps = ctx.extracted_download(int(n_min))
    ex = ctx._last_extract
    view = dict(off_fwd=ex["off_fwd"], off_rev=ex["off_rev"], min_val=mn, min_pos=ps,
                raw_len=np.diff(rs.offs)[order].astype(np.uint32), hpc_len=ex["hpc_len"],
                score=ex["score"], raw_err=ex["raw_err"], hpc_err=ex["hpc_err"],
                state=np.zeros(rs.n, np.uint8), min_qual=7.0)
    keys, offs, post = ctx.index_export()
    ok = cls >= 0
    ncl = int(cls.max()) + 1 if ok.any() else 0
    rep_entry = np.full(ncl, -1, np.int64)
    for i in np.nonzero(ok)[0][::-1]:
        rep_entry[cls[i]] = i            # first entry of each cluster = its creator
    base = rank * 10_000_000
    return pipeline.ClusteredBatch(rep_view=pipeline.gather_records(view, rep_entry),
                                   member_cls=cls[ok].astype(np.int32),
                                   member_read=(base + order[ok]).astype(np.int64),
                                   member_strand=strand[ok].astype(np.int32), mindb=(keys, offs, post),
                                   depth=0, batch_start=base, batch_end=base + rs.n - 1)


def cpu_baseline_sahlin_sample(rs, order, cls, strand, k, w, sample):
    """Sahlin mode on one host core is dominated by 16.7 kb x 16.7 kb alignments (~1 s each): the oracle
    runs on the first `sample` reads of the sorted batch only (its aligner hook calls the product's host
    aligner, parasail being absent); the GPU result on the same sub-batch is the parity check."""
    import ctypes as C
    from oracle import pyoracle as po
    from isonclust2_amd import _lib, api
    from tests.helpers import oracle_sorted_batch
    sub = rs.subset(order[:sample]) if hasattr(rs, "subset") else None
    if sub is None:
        return None
    B, view = oracle_sorted_batch(sub, k, w)
    L = _lib.load()
    CB = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char), C.c_int)
    fn = CB(lambda read, nread, rep, nrep, go, ge, out, cap:
            L.ioc_host_align(read, nread, rep, nrep, 2, -2, go, ge, C.cast(out, C.c_char_p), cap, None))
    po.lib().orc_set_aligner(C.cast(fn, C.c_void_p))
    try:
        t0 = time.perf_counter()
        st = B.cluster(mode="sahlin", stats=True)
        dt = time.perf_counter() - t0
    finally:
        po.lib().orc_set_aligner(None)
    acl, ast = B.assignments(sub.n)
    ocl, ost = acl[view["orig"]], ast[view["orig"]]
    # the same sub-batch through the product (own context: the benchmark's resident batch stays untouched)
    seqs = [sub.read(int(i))[0] for i in view["orig"]]
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    v = dict(view)
    v.update(raw_seq=b"".join(seqs), raw_off=off)
    c2 = api.Context(0)
    gcl, gst, _ = c2.cluster_batch(api.default_params(k, w, "sahlin"), v)
    c2.close()
    mism = int(np.count_nonzero((ocl != gcl) | (ost != gst)))
    return dt, st, mism, sub.n


def cpu_baseline_and_parity(rs, order, cls, strand, k, w):
    """Oracle (CPU restatement, 1 core) on the SAME batch: timing of the ClusterSortedReads region,
    exact M/H/C_s counts for the roofline, and a full-size parity check of the GPU result."""
    from oracle import pyoracle as po
    from tests.helpers import oracle_sorted_batch
    B, view = oracle_sorted_batch(rs, k, w)
    assert np.array_equal(view["orig"], order), "sort order differs from the oracle's"
    t0 = time.perf_counter()
    st = B.cluster(mode="fast", stats=True)
    dt = time.perf_counter() - t0
    acl, ast = B.assignments(rs.n)
    ocl, ost = acl[view["orig"]], ast[view["orig"]]
    mism = int(np.count_nonzero((ocl != cls) | (ost != strand)))
    return dt, st, mism


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--mode", default="fast", choices=["fast", "sahlin"],
                    help="fast = BASELINE.json configs[1] (default); sahlin = configs[2] (GPU alignment fallback)")
    ap.add_argument("--cpu-sample", type=int, default=40, help="sahlin: reads in the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--merge", action="store_true",
                    help="after the timed region: all-gather the clustered batches and left-fold merge them on rank 0")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    k, w = 11, 15

    import torch
    from isonclust2_amd import api, synth

    dist = None
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev          # one process per GPU; (gloo smoke runs may share a device)
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=a.backend)
    dev = torch.device("cuda", dev_index) if (dist is None or a.backend == "nccl") else torch.device("cpu")

    ctx = api.Context(dev_index)
    rs, order, n_min = prepare_resident_batch(ctx, api, synth, a.config, 1 + rank, k, w, a.mode)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(a.warmup):
        cls, strand, st = ctx.cluster_resident()
    barrier()
    ms_score = ms_build = ms_resolve = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cls, strand, st = ctx.cluster_resident()
        tm = ctx.timings()          # HIP events recorded on the launch stream around each phase
        ms_score += tm["ms_score"]
        ms_build += tm["ms_build"]
        ms_resolve += tm["ms_resolve"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nreads = torch.tensor([rs.n], dtype=torch.int64, device=dev)
        dist.all_reduce(nreads, op=dist.ReduceOp.SUM)
        total_reads = int(nreads.item())
    else:
        total_reads = rs.n
    ms_score /= a.steps
    ms_build /= a.steps
    ms_resolve /= a.steps

    if rank == 0:
        value = total_reads * a.steps / elapsed
        cpu = None
        parity = None
        M = tm["n_minimizers"]
        # H = postings the reference's GetMinimizerHits traverses on this batch, counted on the device
        # from the final clustering (instrumentation launch, untimed); C_s = survivor candidates
        H = ctx.count_reference_postings()
        Cs = tm["n_mapped_evals"]
        h_source = "device count (ioc_count_reference_postings)"
        if world == 1 and not a.no_cpu_baseline and a.mode == "sahlin":
            res = cpu_baseline_sahlin_sample(rs, order, cls, strand, k, w, a.cpu_sample)
            if res is not None:
                dt, ost, mism, ns = res
                cpu = {"value": ns / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                       "sample": f"first {ns} reads of the sorted {a.config} batch, sahlin mode, oracle -O3 -msse3 with the "
                                 "product's host aligner behind its aligner hook (parasail absent), 1 run"}
                parity = {"entries": ns, "mismatches": mism, "oracle_aln_invoked": ost["aln_invoked"]}
        elif world == 1 and not a.no_cpu_baseline:
            dt, ost, mism = cpu_baseline_and_parity(rs, order, cls, strand, k, w)
            cpu = {"value": rs.n / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": f"the full {a.config} batch ({rs.n} reads), ClusterSortedReads region, oracle -O3 -msse3, 1 run"}
            parity = {"entries": rs.n, "mismatches": mism, "clusters": st["n_clusters"],
                      "tie_replays": st["n_tie_replays"], "oracle_postings": ost["postings"],
                      "device_postings": H}
            Cs = ost["mapped_calls"]
            h_source += "; equals the oracle's count" if H == ost["postings"] else "; DIFFERS from the oracle's count"
        # algorithmic bytes of one scoring launch: 12 B per minimizer probed +
        # 8 B index row per probe + 4 B per posting traversed + 16 B per surviving candidate
        alg_bytes = None
        roof = None
        if H:
            alg_bytes = 12 * M + 8 * M + 4 * H + 16 * Cs
            ach = alg_bytes / (ms_score * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_score", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg_bytes,
                    "kernel_ms": ms_score, "counts": {"M": M, "H": H, "C_s": Cs, "H_source": h_source}}
            tfile = os.path.join(ROOT, "profiles", "k_score_traffic.json")
            if os.path.exists(tfile):
                try:
                    roof["traffic"] = json.load(open(tfile)).get("hbm_bytes_per_launch")
                except Exception:
                    pass
        out = {
            "metric": f"reads/s clustered ({a.mode} mode, k=11 w=15, 3000-read / 50 Mb batch per GPU)",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{a.config}: {rs.tag}; one sorted batch per GPU, minimizer SoA resident in HBM",
                       "mode": a.mode, "k": k, "w": w, "reads_per_gpu": rs.n, "minimizers_per_gpu": int(n_min),
                       "parallelism": f"batch-shard x{world}, no data-path collective"},
            "phase_ms": {"index_build": ms_build, "score": ms_score, "resolve": ms_resolve,
                         "resolve_sweeps": tm["resolve_iters"]},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
        }
        if a.mode == "sahlin":
            out["alignment"] = {"reads_aligned": st["n_aln_invoked"], "pairs": st["n_aln_pairs"],
                                "rounds": st["aln_rounds"], "order_dependent": st["n_aln_order_dep"]}
        print(json.dumps(out), flush=True)
    if a.merge:
        # config 4: RCCL all-gather of every rank's clustered batch, then the reference's left fold
        # ((b0 + b1) + b2) ... with ioc_cluster_merge on rank 0 (untimed extra, reported on stderr)
        from isonclust2_amd import dist as idist
        from isonclust2_amd import pipeline
        cb = resident_to_clustered(ctx, api, pipeline, rs, order, cls, strand, rank)
        t1 = time.perf_counter()
        allb = idist.allgather_clustered(cb, dist)
        t2 = time.perf_counter()
        if rank == 0:
            merged = idist.fold_merge(ctx, api.default_params(k, w, "fast"), allb)
            t3 = time.perf_counter()
            print(json.dumps({"merge": {"batches": len(allb), "clusters_in": [b.n_clusters for b in allb],
                                        "clusters_out": merged.n_clusters, "allgather_ms": (t2 - t1) * 1e3,
                                        "fold_ms": (t3 - t2) * 1e3}}), file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
