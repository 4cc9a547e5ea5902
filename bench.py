#!/usr/bin/env python3
"""bench.py — reads/s clustered on BASELINE.json's 3000-read / 50 Mb batch (k=11 w=15) on N MI355X GPUs.

Headline (`value`) = sahlin mode, the mode BASELINE.json's metric names (configs[2]): index build +
shared-minimizer scoring + mapped-ratio resolve + the alignment fallback (GPU, batched) + decisions back on
the host.  The fast-mode result of the SAME resident batch (configs[1]: no alignment) rides along in
`fast_mode`, with the HBM roofline of the scoring kernels in `roofline`; `roofline_align` is the integer-VALU
issue roofline of the forward DP kernel that dominates a sahlin step.

One "step" = one pass of the hot path (ioc_cluster_resident) over one sorted batch whose minimizer SoA and raw
sequences are already resident in HBM.  Batches shard one per GPU with no data-path collective (weak
scaling); ranks only meet at the timing barrier.  Inputs are synthetic (isonclust2_amd/synth.py): every rank
generates and prepares its OWN copy of the same batch (seed 1) — weak scaling with the per-GPU work fixed, so that
the max over ranks measures the system and not the spread between batches (other seeds of the same shape take
115-149 ms: some need a second alignment round); --rank-seeds gives rank r the batch of seed 1 + r instead.
Inputs are prepared by the product's own GPU sort stage (ioc_qual_scores / ioc_extract_minimizers), never
by the oracle.  The oracle appears only in the cpu_baseline legs (rank 0, N=1): timed on one host core — the
full batch in fast mode (doubling as a full-size parity check), a bounded sample in sahlin mode.

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


VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # 39.3 T int32 lane-ops/s: 1024 SIMDs x 16 lanes x 2.4 GHz (one
                                               # wave64 int32 VALU instruction = 4 cycles of its SIMD)
ALIGN_VALU_PER_CELL = 5                        # fwd_cells (query-profile kernel): add with byte select, max3, sub, max, max (ioc_align_gpu.hip)


def timed_steps(ctx, torch, dist, dev, steps, warmup):
    """W untimed + K timed passes of the hot path over the resident batch; returns the last result, the
    wall time (max over ranks) and the per-phase HIP-event averages."""
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(warmup):
        cls, strand, st = ctx.cluster_resident()
    barrier()
    acc = dict(ms_score=0.0, ms_build=0.0, ms_resolve=0.0, ms_align_fwd=0.0, ms_align_trace=0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        cls, strand, st = ctx.cluster_resident()
        tm = ctx.timings()          # HIP events recorded on the launch stream around each phase
        for k in acc:
            acc[k] += tm[k]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for k in acc:
        acc[k] /= steps
    return cls, strand, st, tm, elapsed, acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--mode", default="both", choices=["both", "fast", "sahlin"],
                    help="both (default): headline = sahlin mode (BASELINE.json's metric, configs[2]) with the fast-mode "
                         "result (configs[1]) of the same batch beside it; fast / sahlin = that mode only")
    ap.add_argument("--cpu-sample", type=int, default=30, help="sahlin: reads in the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rank-seeds", action="store_true", help="rank r clusters the batch of seed 1 + r (default: every rank the batch of seed 1)")
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
    want_fast = a.mode in ("both", "fast")
    want_sahlin = a.mode in ("both", "sahlin")
    rs, order, n_min = prepare_resident_batch(ctx, api, synth, a.config, 1 + (rank if a.rank_seeds else 0), k, w,
                                              "sahlin" if want_sahlin else "fast")
    if dist is not None:
        nreads = torch.tensor([rs.n], dtype=torch.int64, device=dev)
        dist.all_reduce(nreads, op=dist.ReduceOp.SUM)
        total_reads = int(nreads.item())
    else:
        total_reads = rs.n
    single = world == 1 and not a.no_cpu_baseline

    fast = None
    if want_fast:
        ctx.set_params(api.default_params(k, w, "fast"))
        cls, strand, st, tm, elapsed, acc = timed_steps(ctx, torch, dist, dev, a.steps, a.warmup)
        fast = {"value": total_reads * a.steps / elapsed, "unit": "reads/s", "ms_per_step": elapsed / a.steps * 1e3,
                "phase_ms": {"index_build": acc["ms_build"], "score": acc["ms_score"], "resolve": acc["ms_resolve"],
                             "resolve_sweeps": tm["resolve_iters"]},
                "clusters": st["n_clusters"]}
        fast_res = (cls, strand, st, tm, acc)
    sah = None
    if want_sahlin:
        ctx.set_params(api.default_params(k, w, "sahlin"))
        cls, strand, st, tm, elapsed, acc = timed_steps(ctx, torch, dist, dev, a.steps, a.warmup)
        sah = {"value": total_reads * a.steps / elapsed, "unit": "reads/s", "ms_per_step": elapsed / a.steps * 1e3,
               "phase_ms": {"index_build": acc["ms_build"], "score": acc["ms_score"], "resolve_last": acc["ms_resolve"],
                            "align_fwd": acc["ms_align_fwd"], "align_trace": acc["ms_align_trace"]},
               "clusters": st["n_clusters"],
               "alignment": {"reads_aligned": st["n_aln_invoked"], "pairs": st["n_aln_pairs"],
                             "rounds": st["aln_rounds"], "order_dependent": st["n_aln_order_dep"],
                             "cells": tm["n_align_cells"],
                             "refused_by_packed_kernel": tm.get("n_align_refused", 0)}}
        sah_res = (cls, strand, st, tm, acc)

    if rank == 0:
        out_extra = {}
        # ---- fast mode (configs[1]): HBM roofline of the scoring kernels + full-batch CPU baseline / parity ----
        roof = None
        if fast is not None:
            ctx.set_params(api.default_params(k, w, "fast"))
            fcls, fstrand, fst = ctx.cluster_resident()     # untimed: leaves the fast-mode clustering on the device
            M = fast_res[3]["n_minimizers"]
            # H = postings the reference's GetMinimizerHits traverses on this batch, counted on the device
            # from the final clustering (instrumentation launch, untimed); C_s = survivor candidates
            H = ctx.count_reference_postings()
            Cs = fast_res[3]["n_mapped_evals"]
            h_source = "device count (ioc_count_reference_postings)"
            if single:
                dt, ost, mism = cpu_baseline_and_parity(rs, order, fcls, fstrand, k, w)
                fast["cpu_baseline"] = {"value": rs.n / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                                        "sample": f"the full {a.config} batch ({rs.n} reads), fast mode, ClusterSortedReads "
                                                  "region, oracle -O3 -msse3, 1 run"}
                fast["parity"] = {"entries": rs.n, "mismatches": mism, "clusters": fst["n_clusters"],
                                  "tie_replays": fst["n_tie_replays"], "oracle_postings": ost["postings"],
                                  "device_postings": H}
                Cs = ost["mapped_calls"]
                h_source += "; equals the oracle's count" if H == ost["postings"] else "; DIFFERS from the oracle's count"
            # algorithmic bytes of one scoring launch: 12 B per minimizer probed +
            # 8 B index row per probe + 4 B per posting traversed + 16 B per surviving candidate
            if H:
                ms_score = fast_res[4]["ms_score"]
                alg_bytes = 12 * M + 8 * M + 4 * H + 16 * Cs
                ach = alg_bytes / (ms_score * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": "k_score (k_partition_mins + k_score_part + k_score_compact)",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg_bytes,
                        "kernel_ms": ms_score, "counts": {"M": M, "H": H, "C_s": Cs, "H_source": h_source}}
                tfile = os.path.join(ROOT, "profiles", "k_score_traffic.json")
                if os.path.exists(tfile):
                    try:
                        roof["traffic"] = json.load(open(tfile)).get("hbm_bytes_per_launch")
                    except Exception:
                        pass
        # ---- sahlin mode (configs[2]): the step is dominated by the alignment fallback's forward DP, an
        # integer-VALU kernel (no HBM traffic to speak of, no MFMA): its issue-rate roofline beside the HBM one ----
        roof_aln = None
        if sah is not None:
            cells = sah_res[3]["n_align_cells"]
            ms_fwd = sah_res[4]["ms_align_fwd"]
            if cells and ms_fwd > 0:
                ach = cells * ALIGN_VALU_PER_CELL / (ms_fwd * 1e-3) / 1e12
                roof_aln = {"bound": "valu-int32", "kernel": "k_align_fwd", "achieved": ach, "peak": VALU_PEAK_TOPS,
                            "unit": "T lane-op/s", "frac": ach / VALU_PEAK_TOPS, "kernel_ms": ms_fwd,
                            "cells": cells, "valu_per_cell": ALIGN_VALU_PER_CELL,
                            "gcells_per_s": cells / (ms_fwd * 1e-3) / 1e9}
            if single:
                res = cpu_baseline_sahlin_sample(rs, order, sah_res[0], sah_res[1], k, w, a.cpu_sample)
                if res is not None:
                    dt, ost, mism, ns = res
                    sah["cpu_baseline"] = {"value": ns / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                                           "sample": f"first {ns} reads of the sorted {a.config} batch, sahlin mode, oracle -O3 "
                                                     "-msse3 with the product's host aligner behind its aligner hook "
                                                     "(parasail absent), 1 run"}
                    sah["parity"] = {"entries": ns, "mismatches": mism, "oracle_aln_invoked": ost["aln_invoked"]}
        head, head_mode = (sah, "sahlin") if sah is not None else (fast, "fast")
        out = {
            "metric": f"reads/s clustered ({head_mode} mode, k=11 w=15)",
            "value": head["value"], "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{a.config} = BASELINE.json configs[{2 if head_mode == 'sahlin' else 1}]: {rs.tag}; one sorted "
                                   "3000-read / 50 Mb batch per GPU, minimizer SoA and sequences resident in HBM",
                       "mode": head_mode, "k": k, "w": w, "reads_per_gpu": rs.n, "minimizers_per_gpu": int(n_min),
                       "parallelism": f"batch-shard x{world}, no data-path collective",
                       "rank_batches": "seed 1 + rank" if a.rank_seeds else "every rank its own copy of the seed-1 batch"},
            "phase_ms": head["phase_ms"],
            "roofline": roof if roof is not None else roof_aln,
            "cpu_baseline": head.get("cpu_baseline"), "parity": head.get("parity"),
        }
        if head_mode == "sahlin":
            out["alignment"] = sah["alignment"]
            out["roofline_align"] = roof_aln
            if fast is not None:
                out["fast_mode"] = fast       # BASELINE.json configs[1] on the same batch
        print(json.dumps(out), flush=True)
    if a.merge:
        # config 4: RCCL all-gather of every rank's clustered batch, then the reference's left fold
        # ((b0 + b1) + b2) ... with ioc_cluster_merge on rank 0 (untimed extra, reported on stderr)
        from isonclust2_amd import dist as idist
        from isonclust2_amd import pipeline
        ctx.set_params(api.default_params(k, w, "fast"))
        cls, strand, st = ctx.cluster_resident()
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
