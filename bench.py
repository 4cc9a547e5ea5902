#!/usr/bin/env python3
"""bench.py — reads/s clustered on BASELINE.json's 3000-read / 50 Mb batch (k=11 w=15) on N MI355X GPUs.

Headline (`value`) = sahlin mode, the mode BASELINE.json's metric names (configs[2]): index build +
shared-minimizer scoring + mapped-ratio resolve + the alignment fallback (GPU, batched) + decisions back on
the host, over ONE sorted batch whose minimizer SoA and raw sequences are ALREADY RESIDENT IN HBM when the timed
region starts (`value_region: "resident"`, the driver's contract).  The fast-mode result of the same resident batch
(configs[1]) rides along in `fast_mode`.  Beside it, the two regions SURVEY.md §8(d) defines:

  core   host arrays (the parsed batch in host RAM) -> assignments + MinDB back in host RAM; H2D of the 191 MB
         minimizer SoA (+ 50 MB of sequences in sahlin mode) and every D2H included (ioc_cluster_merge +
         ioc_index_export).  The >= 20x target is judged on this region.
  cli    the whole `isONclust2-hip cluster` process on the batch's .cer file (load, core, bookkeeping, save).

`roofline` = HBM roofline of the shared-minimizer scoring kernels (the kernel north_star asks the HBM fraction
of); `roofline_align` = the integer-VALU issue roofline of the forward DP kernel that dominates a sahlin step
(peak from the issue-rate microbenchmark tools/micro/valu_rate.hip, profiles/r02_valu_rate.txt).

N > 1: one process per GPU; rank r clusters ITS OWN batch (seed 1 + r: BASELINE.json configs[3] = 8 different
batches), no data-path collective, ranks meet at the timing barrier (weak scaling; `--same-seed` gives every rank
the seed-1 batch instead).  After the timed region the batches are MERGED (the one exchange step of the path):
representative records all-gathered, folded left to right; reported in `merge` with its own time.

CPU baseline (rank 0, N = 1): the oracle (CPU restatement, `-O3 -DNDEBUG -msse3`, libstdc++) on ONE pinned core:
fast mode = the full batch (doubling as a full-size parity check), sahlin mode = a bounded sample with the oracle's
own SCALAR aligner (not parasail's striped SIMD scan).  min over the runs that fit the time budget (<= 3).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
# int32 VALU issue peak: MEASURED with tools/micro/valu_rate.hip (profiles/r02_valu_rate.txt), not assumed; the
# file's JSON line overrides this default when present
VALU_PEAK_TOPS_DEFAULT = 256 * 4 * 16 * 2.4e9 / 1e12
ALIGN_VALU_PER_CELL = 5  # fwd_cells: add with byte select, max3, sub, max, max (ioc_align_gpu.hip)


def valu_rates():
    """The JSON line of the issue-rate microbenchmark (tools/micro/valu_rate.hip -> profiles/r04_valu_rate.txt): T lane-op/s per
    instruction, every wave at work for the same 40 ms of wall clock, best over 1-8 waves per SIMD."""
    for name in ("r04_valu_rate.txt", "r03_valu_rate.txt", "r02_valu_rate.txt"):
        try:
            for line in open(os.path.join(ROOT, "profiles", name)):
                if line.startswith("JSON "):
                    return json.loads(line[5:]), name
        except Exception:
            pass
    return {}, None


def valu_peak():
    """(peak T lane-op/s, source): the best rate any plain int32 VALU instruction reached in the microbenchmark."""
    d, name = valu_rates()
    ks = [d[k] for k in ("v_add_u32", "v_max_i32", "v_max3_i32") if k in d]
    if ks:
        return max(ks), f"measured: profiles/{name} (best of v_add_u32 / v_max_i32 / v_max3_i32 over 1-8 waves per SIMD)"
    return VALU_PEAK_TOPS_DEFAULT, "assumed 4 cycles per wave64 instruction per SIMD (no measurement file)"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class pinned:
    """taskset for the calling thread: one core for the duration of a CPU-baseline run."""

    def __enter__(self):
        self.old = os.sched_getaffinity(0)
        self.core = max(self.old)
        os.sched_setaffinity(0, {self.core})
        return self

    def __exit__(self, *a):
        os.sched_setaffinity(0, self.old)


def prepare(ctx, api, pipeline, synth, config, seed, k, w, rank):
    """raw reads -> GPU quality scores -> stable sort -> GPU HPC/minimizers: the sorted batch as host arrays
    (core / cli / merge regions) AND as resident queries (the headline's timed region)."""
    rs = synth.generate_config(config, seed=seed)
    sb, order = pipeline.sort_stage(ctx, rs, k, w, read_id_base=rs.n * rank, batch_nr=rank)
    v = sb.view
    p = api.default_params(k, w, "sahlin")
    ctx.set_params(p)
    # gates of the clustering loop (src/cluster.cpp:116-160)
    keep = (v["state"] == 0) & (v["score"] >= 0) & (v["raw_len"] >= 2 * k) & (v["hpc_len"] >= 2 * k)
    cell = np.array([api.host_err_cell(e) if kp else 1 for e, kp in zip(v["hpc_err"], keep)], np.uint8)
    need = np.array([api.host_min_total(h, p.mapped_threshold) if kp else 0xFFFFFFFE
                     for h, kp in zip(v["hpc_len"], keep)], np.uint32)
    ctx.queries_from_extracted(keep, cell, need)     # the extractor's output stays where it is: in HBM
    ctx.left_load(0, None, None, None, None)
    ctx.resident_set_sequences(v["raw_seq"], v["raw_off"], v["raw_err"])
    return rs, sb, order


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


def core_region(ctx, api, pipeline, sb, k, w, mode, runs):
    """SURVEY §8(d) *core*: the parsed batch in host RAM -> assignments + updated MinDB in host RAM, through
    ioc_cluster_merge + ioc_index_export; PCIe both ways inside the timed region."""
    p = api.default_params(k, w, mode)
    view = sb.view if mode != "fast" else {kk: vv for kk, vv in sb.view.items() if kk not in ("raw_seq", "raw_off")}
    ms, cb = [], None
    for _ in range(runs + 1):                      # first run untimed (allocations)
        tm = {}
        cb = pipeline.cluster_single(ctx, p, pipeline.SortedBatch(view=view, read_ids=sb.read_ids, batch_nr=sb.batch_nr,
                                                                  batch_start=sb.batch_start, batch_end=sb.batch_end), timing=tm)
        ms.append(tm["abi_ms"])                    # ioc_cluster_merge + ioc_index_export, nothing else
        parts = (tm["cluster_ms"], tm["export_ms"])
    ms = ms[1:]
    h2d = sum(np.asarray(view[x]).nbytes for x in ("min_val", "min_pos", "off_fwd", "off_rev", "hpc_len")) + (len(view.get("raw_seq", b"")))
    return {"ms_min": min(ms), "ms_mean": sum(ms) / len(ms), "runs": len(ms), "last_run_ms": {"ioc_cluster_merge": parts[0], "ioc_index_export": parts[1]},
            "h2d_bytes": int(h2d),
            "d2h_bytes": int(sum(a.nbytes for a in cb.mindb) + 5 * len(sb.read_ids))}, cb


def cli_region(rs, modes, runs=3):
    """SURVEY §8(d) *cli*: the whole `isONclust2-hip cluster -l batch.cer -o out.cer -x mode` process, `runs` processes back to
    back per mode — `one_shot`: the job runs in the process the caller started (ISONCLUST2_SERVE=0: runtime start, first-use
    costs and the driver taking the context back are part of every call); `served`: the command's default, the job is handed to
    the resident worker that the first `cluster` of a pipeline starts (csrc/cli/main.cpp "serve"; its first call is reported
    apart).  A fresh output path per run, as a pipeline's are."""
    cli = os.path.join(ROOT, "isonclust2_amd", "bin", "isONclust2-hip")
    if not os.path.exists(cli):
        return None
    d = tempfile.mkdtemp(prefix="ioc_bench_")
    keys = ("process_wall_ms", "cli_ms", "core_ms", "load_ms", "ctx_ms", "trim_ms", "bookkeeping_ms", "save_ms", "before_main_ms", "after_exit_ms")
    try:
        fq = os.path.join(d, "reads.fq")
        with open(fq, "wb") as f:
            for i in range(rs.n):
                s, q = rs.read(i)
                f.write(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
        t = time.perf_counter()
        subprocess.check_call([cli, "sort", "-B", "60000", "-M", str(rs.n), "-o", os.path.join(d, "sorted"), fq],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_sort = time.perf_counter() - t
        batch = os.path.join(d, "sorted", "batches", "isONbatch_0.cer")
        out = {"sort_process_s": t_sort, "batch_cer_MB": os.path.getsize(batch) / 1e6}
        nr = 0

        def one(mode, env):
            nonlocal nr
            nr += 1
            o = os.path.join(d, f"out{nr}.cer")
            t0, m0 = time.perf_counter(), time.monotonic() * 1e3
            r = subprocess.run([cli, "cluster", "-l", batch, "-o", o, "-x", mode], capture_output=True, text=True, env=env)
            wall, m1 = (time.perf_counter() - t0) * 1e3, time.monotonic() * 1e3
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            j = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
            if os.environ.get("IOC_BENCH_CLI_TRACE"):   # (developer aid, with IOC_TRACE=1: where a child spent its time)
                sys.stderr.write(f"--- cluster -x {mode} {'served' if env.get('ISONCLUST2_SERVE') != '0' else 'one-shot'}: wall {wall:.1f} ms\n" +
                                 "\n".join(l for l in r.stderr.splitlines() if l.startswith("[ioc]")) + "\n")
            j["process_wall_ms"] = wall
            if "t_begin_mono_ms" in j:   # (one-shot: time before main and after _exit; served: before the worker had the job / after it answered)
                j["before_main_ms"], j["after_exit_ms"] = j["t_begin_mono_ms"] - m0, m1 - j["t_end_mono_ms"]
            os.unlink(o)
            return j

        for mode in modes:
            res = {}
            for kind, env in (("one_shot", dict(os.environ, ISONCLUST2_STATS_JSON="1", ISONCLUST2_SERVE="0")),
                              ("served", dict(os.environ, ISONCLUST2_STATS_JSON="1", ISONCLUST2_SERVE_DIR=os.path.join(d, "srv")))):
                first = one(mode, env) if kind == "served" else None     # (starts the worker: a pipeline pays this once)
                rr = [one(mode, env) for _ in range(runs)]
                bad = [j for j in rr + ([first] if first else []) if "error" in j]
                if bad:
                    res[kind] = bad[0]
                    continue
                best = min(rr, key=lambda j: j["process_wall_ms"])
                res[kind] = {"process_wall_ms_min": best["process_wall_ms"], "process_wall_ms_all": [round(j["process_wall_ms"], 1) for j in rr],
                             "phases_of_that_run": {k: v for k, v in best.items() if not k.startswith("t_")}, "runs": runs,
                             "phases_all_runs": [{k: round(j[k], 1) for k in keys if k in j} for j in rr]}
                if first:
                    res[kind]["first_call_starting_the_worker"] = {k: round(first[k], 1) for k in keys if k in first}
                    subprocess.run([cli, "serve", "stop"], capture_output=True, env=env)
            # `cli.<mode>` = the command as a user runs it (its default: the job goes to the resident worker), the process that
            # does the job itself (ISONCLUST2_SERVE=0, what rounds 1 - 4 reported under these keys) beside it
            out[mode] = dict(res.get("served", {}), how="`isONclust2-hip cluster` as installed: the calling process hands the job to the resident "
                                                       "worker that the first call of a pipeline starts (`first_call_starting_the_worker`); "
                                                       "`one_shot` = ISONCLUST2_SERVE=0, the job runs in the calling process",
                             one_shot=res.get("one_shot"))
        return out
    finally:
        subprocess.call(["rm", "-rf", d])


def cpu_baseline_fast(rs, order, cls, strand, k, w, max_runs, budget_s):
    """Oracle (CPU restatement) on the SAME batch, one pinned core: timing of the ClusterSortedReads region (min over
    the runs), exact M/H/C_s counts for the roofline, and a full-size parity check of the GPU result."""
    from tests.helpers import oracle_sorted_batch
    times, st, mism = [], None, None
    t_all = time.perf_counter()
    with pinned() as pin:
        for _ in range(max_runs):
            B, view = oracle_sorted_batch(rs, k, w)
            assert np.array_equal(view["orig"], order), "sort order differs from the oracle's"
            t0 = time.perf_counter()
            st = B.cluster(mode="fast", stats=True)
            times.append(time.perf_counter() - t0)
            acl, ast = B.assignments(rs.n)
            ocl, ost = acl[view["orig"]], ast[view["orig"]]
            mism = int(np.count_nonzero((ocl != cls) | (ost != strand)))
            if time.perf_counter() - t_all + times[-1] > budget_s:
                break
    return min(times), times, st, mism, pin.core


def cpu_simd_score_pass(rs, order, n_pairs=24):
    """What the reference's alignment calls cost at least on this host: a 16-bit SSE2 striped score pass of the semi-global alignment
    (oracle/sg_striped.cpp, Farrar 2007 — parasail, which the reference calls, is absent from its tree; its sg_trace_scan_16 also
    writes a traceback table and re-runs saturated pairs in 32 bits) on pairs of neighbouring reads of the sorted batch, one pinned
    core.  Returns G cell updates per second."""
    import ctypes as C
    from oracle import pyoracle as po
    L = po.lib()
    cells, t_all, sat = 0, 0.0, 0
    with pinned():
        for x in range(n_pairs):
            a, _ = rs.read(int(order[2 * x]))
            b, _ = rs.read(int(order[2 * x + 1]))
            f = C.c_int32()
            t0 = time.perf_counter()
            L.orc_sg_striped16(a, len(a), b, len(b), 2, -2, 3, 1, C.byref(f))
            t_all += time.perf_counter() - t0
            cells += len(a) * len(b)
            sat += f.value
    return {"gcups": cells / t_all / 1e9, "pairs": n_pairs, "seconds": t_all, "saturated_pairs": sat}


def cpu_baseline_sahlin(ctx_factory, api, pipeline, rs, order, k, w, sample, max_runs, budget_s):
    """Sahlin mode on one host core is dominated by 16.7 kb x 16.7 kb alignments: the oracle (with its OWN scalar
    aligner — not parasail's SIMD scan, not the product's) runs on the first `sample` reads of the sorted batch; the
    product clusters the same sub-batch on the GPU for the parity check."""
    from tests.helpers import oracle_sorted_batch
    sub = rs.subset(order[:sample])
    times, st = [], None
    t_all = time.perf_counter()
    with pinned() as pin:
        for _ in range(max_runs):
            B, view = oracle_sorted_batch(sub, k, w)
            t0 = time.perf_counter()
            st = B.cluster(mode="sahlin", stats=True)
            times.append(time.perf_counter() - t0)
            acl, ast = B.assignments(sub.n)
            ocl, ost = acl[view["orig"]], ast[view["orig"]]
            if time.perf_counter() - t_all + times[-1] > budget_s:
                break
    c2 = ctx_factory()
    sb, o2 = pipeline.sort_stage(c2, sub, k, w)
    assert np.array_equal(o2, view["orig"])
    gcl, gst, _ = c2.cluster_batch(api.default_params(k, w, "sahlin"), sb.view)
    c2.close()
    mism = int(np.count_nonzero((ocl != gcl) | (ost != gst)))
    return min(times), times, st, mism, sub.n, pin.core


_NODE_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
os.sched_setaffinity(0, {{{core}}})
from isonclust2_amd import synth
from oracle import pyoracle as po
rs = synth.generate_config({config!r}, seed={seed})
R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
R.score_sort({k}, {w})
B = po.Batch(R, 0, rs.n - 1, po.default_params({k}, {w}))
print("READY", flush=True)
sys.stdin.readline()
t0 = time.perf_counter()
st = B.cluster(mode="fast")
dt = time.perf_counter() - t0
print(json.dumps({{"seed": {seed}, "core": {core}, "seconds": dt, "clusters": B.n_clusters(), "reads": rs.n}}), flush=True)
"""


def cpu_baseline_node(config, k, w, max_procs=8):
    """SURVEY §8(d) / BASELINE.md §3: the reference's `cluster` is single-threaded, so a NODE runs P single-batch processes on
    P cores (configs[3]: one batch per process, seeds 1..P).  Here: P oracle processes, each pinned to a core of its own,
    prepared first (reads, sort stage) and released together; node throughput = all reads / the slowest process."""
    cores = sorted(os.sched_getaffinity(0))
    P = min(max_procs, len(cores))
    if P < 1:
        return None
    procs = []
    try:
        for i in range(P):
            code = _NODE_CHILD.format(root=ROOT, core=cores[-1 - i], config=config, seed=1 + i, k=k, w=w)
            procs.append(subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for pr in procs:
            if pr.stdout.readline().strip() != "READY":
                raise RuntimeError("a CPU-baseline child did not come up")
        t0 = time.perf_counter()
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        res = [json.loads(pr.stdout.readline()) for pr in procs]
        wall = time.perf_counter() - t0
        for pr in procs:
            pr.wait(timeout=60)
    except Exception as e:  # noqa: BLE001
        for pr in procs:
            pr.kill()
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    reads = sum(r["reads"] for r in res)
    return {"value": reads / wall, "unit": "reads/s", "cores": P, "processes": P, "kind": "port", "mode": "fast", "cpu_model": cpu_model(),
            "seconds_per_process": [round(r["seconds"], 2) for r in res], "wall_s": round(wall, 2),
            "sample": f"{P} concurrent single-batch oracle processes (the batches of seeds 1..{P} of {config}: configs[3]'s shape), each pinned "
                      f"to one core, fast mode, ClusterSortedReads region, released together; value = {reads} reads / the wall time of "
                      "the slowest; the sahlin leg is not repeated at node level (a read that reaches the scalar aligner costs ~0.5 s: "
                      "P times the single-core sample at best)"}


def native_merge_leg(ctx, api, pipeline, idist, dist, torch, sb, cb, k, w, mode, torch_merge, timeout_s=240):
    """The same merge once more through the library's OWN multi-GPU binding (csrc/ioc_dist.cpp: ncclCommInitRank, ragged
    all-gather as grouped broadcasts, one-pass merge — all in C++), after re-clustering the batch so that its
    representatives' lists are device-resident again; then, in fast mode, the merge once replicated and once with scoring and
    resolve sharded over the ranks (ioc_set_shard over RCCL).  The library's collectives run in a thread with a deadline: a
    communicator that does not come up must not cost the bench line.  Every torch.distributed collective of this leg (the
    id's broadcast, the agreement on the outcome, the maxima over ranks) is made by the MAIN thread, in the same order on
    every rank whatever happened inside the thread."""
    import threading
    from isonclust2_amd.digest import fnv1a_reads
    world = dist.get_world_size() if dist is not None else 1
    ident = idist.native_unique_id(ctx, dist, torch)
    loc = {}

    def work():
        try:
            torch.cuda.set_device(ctx.device)   # (a fresh thread starts on device 0: HIP's current device is per thread)
            p = api.default_params(k, w, mode)
            view = sb.view if mode != "fast" else {kk: vv for kk, vv in sb.view.items() if kk not in ("raw_seq", "raw_off")}
            cb2 = pipeline.cluster_single(ctx, p, pipeline.SortedBatch(view=view, read_ids=sb.read_ids, batch_nr=sb.batch_nr,
                                                                       batch_start=sb.batch_start, batch_end=sb.batch_end))
            idist.native_init(ctx, dist, torch, ident=ident)
            tm = {}
            t0 = time.perf_counter()
            mg = idist.merge_all_native(ctx, p, cb2, torch, timing=tm)
            wall = (time.perf_counter() - t0) * 1e3
            loc["main"] = dict(clusters_out=mg.n_clusters, reads_assigned=int(len(mg.member_read)), fnv1a=fnv1a_reads(mg),
                               wall_ms=wall, exchange_lists_ms=tm["exchange_lists_ms"], merge_ms=tm["merge_ms"],
                               bytes_lists_this_rank=tm["bytes_lists"], bytes_records_this_rank=tm["bytes_records"],
                               sharded=bool(tm.get("sharded")), exchanges=tm.get("exchanges"))
            if world > 1:
                # fast mode once replicated (every rank scores and decides every representative) and once sharded (rank r
                # takes the representatives j with j % world == r; `valid` all-reduced over RCCL after every sweep)
                pf = api.default_params(k, w, "fast")
                vf = {kk: vv for kk, vv in sb.view.items() if kk not in ("raw_seq", "raw_off")}
                legs = {}
                for tag, env in (("replicated", "0"), ("sharded", "1")):
                    os.environ["IOC_DIST_SHARD"] = env      # (read by ioc_dist_merge at every call; this thread is the only one inside the library)
                    best = None
                    for _ in range(3):
                        # re-clustered before EVERY repetition: a merge replaces the context's queries, and a second merge of the
                        # same ClusteredBatch would silently take the host-array path (H2D) instead of the device gather
                        cbf = pipeline.cluster_single(ctx, pf, pipeline.SortedBatch(view=vf, read_ids=sb.read_ids, batch_nr=sb.batch_nr,
                                                                                    batch_start=sb.batch_start, batch_end=sb.batch_end))
                        tf = {}
                        mf = idist.merge_all_native(ctx, pf, cbf, torch, timing=tf)
                        t = ctx.timings()
                        rec = dict(merge_ms=tf["merge_ms"], score_ms=t["ms_score"], resolve_ms=t["ms_resolve"], sweeps=t["resolve_iters"],
                                   exchanges=tf["exchanges"], sharded=bool(tf["sharded"]), clusters_out=mf.n_clusters, fnv1a=fnv1a_reads(mf),
                                   representatives=int(sum(tf["clusters_in"])), lists_path=tf.get("lists_path"))
                        if best is None or rec["merge_ms"] < best["merge_ms"]:
                            best = rec
                    legs[tag] = best
                os.environ.pop("IOC_DIST_SHARD", None)
                loc["fast"] = legs
        except Exception as e:  # noqa: BLE001
            loc["error"] = f"{type(e).__name__}: {e}"[:400]

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    status = 2 if th.is_alive() else (1 if "error" in loc else 0)
    worst = int(idist.max_over_ranks(status, dist))
    if worst == 2:
        return {"error": f"no result within {timeout_s} s on " + ("this rank" if status == 2 else "another rank") +
                         " (thread left running; every rank exits without joining it)", "timed_out": True}
    if worst == 1:
        return {"error": loc.get("error", "another rank failed")}
    mx = lambda v: idist.max_over_ranks(v, dist)  # noqa: E731
    m = loc["main"]
    out = dict(m, wall_ms=mx(m["wall_ms"]), exchange_lists_ms=mx(m["exchange_lists_ms"]), merge_ms=mx(m["merge_ms"]),
               equals_torch_path=bool(m["fnv1a"] == torch_merge.get("fnv1a") and m["clusters_out"] == torch_merge.get("clusters_out")),
               binding="C++ over RCCL inside libisonclust2_hip.so (ioc_dist_init / ioc_dist_merge); lists HBM to HBM, "
                       "ragged all-gather = one ncclBroadcast per rank in one group")
    if "fast" in loc:
        legs = {}
        for tag in ("replicated", "sharded"):
            r = dict(loc["fast"][tag])
            for key in ("merge_ms", "score_ms", "resolve_ms"):
                r[key] = mx(r[key])
            legs[tag] = r
        legs["same_clustering"] = legs["replicated"]["fnv1a"] == legs["sharded"]["fnv1a"]
        legs["note"] = ("fast-mode merge of all ranks' representatives through ioc_dist_merge: IOC_DIST_SHARD=0 (every rank scores and "
                        "decides every representative) against the default (rank r owns j % world == r, `valid` all-reduced per sweep); "
                        "min of 3, maxima over ranks; score_ms / resolve_ms are the device phases inside merge_ms")
        out["fast_mode_sharded_resolve"] = legs
    return out


def visible_gpus():
    """Number of GPUs this process would see, WITHOUT touching HIP or torch: the kfd topology in sysfs (a node with
    simd_count > 0 is a GPU), cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.
    None = unknown (no kfd in sysfs, unreadable): the caller then skips its check and lets the ranks find out."""
    import glob
    n = 0
    try:
        nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
        if not nodes:
            return None
        for f in nodes:
            for line in open(f):
                if line.startswith("simd_count"):
                    n += int(line.split()[1]) > 0
                    break
    except Exception:  # noqa: BLE001
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(a):
    """`python bench.py --gpus N` on its own: start N ranks (one process per GPU, torch.distributed.run, RCCL) as CHILD
    processes and pass their one JSON line through.  This parent imports neither torch nor the library and makes no HIP
    call (devices are counted in sysfs), so no process that has initialised a GPU is ever replaced or forked from."""
    import socket
    ndev = visible_gpus()
    if ndev is not None and ndev < a.gpus and a.backend == "nccl":
        print(json.dumps({"error": f"--gpus {a.gpus} but {ndev} GPU(s) are visible: RCCL needs one device per rank "
                                   "(--backend gloo lets ranks share a device: plumbing rehearsal only)", "n_gpus": a.gpus}), flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--mode", default="both", choices=["both", "fast", "sahlin"],
                    help="both (default): headline = sahlin mode (BASELINE.json's metric, configs[2]) with the fast-mode "
                         "result (configs[1]) of the same batch beside it; fast / sahlin = that mode only")
    ap.add_argument("--cpu-sample", type=int, default=64, help="sahlin: reads in the CPU-baseline sample (a read that reaches the fallback costs ~0.5 s of scalar alignment)")
    ap.add_argument("--aligner", default="lean", choices=["fat", "lean"],
                    help="sahlin headline: fat = IOC_ALIGN_ARENA=fat (fine checkpoints, 56 GB arena kept by the resident process), "
                         "lean = the library default (coarse checkpoints, 7 GB); the other one is reported beside it")
    ap.add_argument("--cpu-runs", type=int, default=3, help="CPU baseline: at most this many runs (min is reported)")
    ap.add_argument("--cpu-budget", type=float, default=75.0, help="CPU baseline: seconds per leg after which no further run starts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-core", action="store_true", help="skip the core region (host arrays -> host results)")
    ap.add_argument("--no-cli", action="store_true", help="skip the cli region (whole `cluster` process on a .cer file)")
    ap.add_argument("--same-seed", action="store_true", help="every rank clusters its own copy of the seed-1 batch (default: rank r the batch of seed 1 + r)")
    ap.add_argument("--rank-seeds", action="store_true", help="(default since round 2; kept for old command lines)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-merge", action="store_true", help="N > 1: skip the merge of the ranks' batches after the timed region")
    ap.add_argument("--merge", action="store_true", help="(default since round 2; kept for old command lines)")
    ap.add_argument("--no-native-merge", action="store_true", help="N > 1: skip the second merge leg through the library's own C++ / RCCL binding (ioc_dist_merge)")
    ap.add_argument("--strict", action="store_true", help="exit code 3 when a leg of the merge (after the timed region) never came back; default: the line reports it, exit 0")
    ap.add_argument("--no-cpu-node", action="store_true", help="skip the node-level CPU leg (P concurrent single-batch oracle processes on P cores)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    k, w = 11, 15

    import torch
    from isonclust2_amd import api, pipeline, synth

    dist = None
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev          # one process per GPU; (gloo smoke runs may share a device)
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        tmo = datetime.timedelta(seconds=600)   # (a rank lost in the merge must not hold the others for the default half hour)
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend=a.backend, timeout=tmo)
    dev = torch.device("cuda", dev_index) if (dist is None or a.backend == "nccl") else torch.device("cpu")

    ctx = api.Context(dev_index)
    want_fast = a.mode in ("both", "fast")
    want_sahlin = a.mode in ("both", "sahlin")
    seed = 1 if a.same_seed else 1 + rank
    rs, sb, order = prepare(ctx, api, pipeline, synth, a.config, seed, k, w, rank)
    n_min = int(sb.view["off_rev"][-1])
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
    sah_lean = None
    if want_sahlin:
        ctx.set_params(api.default_params(k, w, "sahlin"))
        # The library's default aligner is the LEAN one (version 2: two pairs per wave, coarse checkpoints, 7 GB of arena for this
        # batch): what a one-shot `cluster` process gets.  A long-lived process that keeps the batch resident can spend 56 GB of
        # HBM on version 1's fine checkpoints (IOC_ALIGN_ARENA=fat) and halve the traceback.  Both are timed; the headline is the
        # mode named in `config.aligner`.
        def sahlin_run(mode):
            if mode == "fat":
                os.environ["IOC_ALIGN_ARENA"] = "fat"
            else:
                os.environ.pop("IOC_ALIGN_ARENA", None)
            r = timed_steps(ctx, torch, dist, dev, a.steps, a.warmup)
            os.environ.pop("IOC_ALIGN_ARENA", None)
            return r
        other = "fat" if a.aligner == "lean" else "lean"
        cls, strand, st, tm, elapsed, acc = sahlin_run(other)          # the mode that is NOT the headline, reported beside it
        sah_lean = {"aligner": other, "ms_per_step": elapsed / a.steps * 1e3, "value": total_reads * a.steps / elapsed, "align_fwd_ms": acc["ms_align_fwd"],
                    "align_trace_ms": acc["ms_align_trace"], "arena_bytes": tm.get("align_arena_bytes"), "aligner_version": tm.get("align_version"),
                    "clusters": st["n_clusters"], "_res": (cls, strand, st)}
        cls, strand, st, tm, elapsed, acc = sahlin_run(a.aligner)
        sah = {"value": total_reads * a.steps / elapsed, "unit": "reads/s", "ms_per_step": elapsed / a.steps * 1e3,
               "phase_ms": {"index_build": acc["ms_build"], "score": acc["ms_score"], "resolve_last": acc["ms_resolve"],
                            "align_fwd": acc["ms_align_fwd"], "align_trace": acc["ms_align_trace"]},
               "clusters": st["n_clusters"],
               "alignment": {"reads_aligned": st["n_aln_invoked"], "pairs": st["n_aln_pairs"],
                             "rounds": st["aln_rounds"], "order_dependent": st["n_aln_order_dep"],
                             "cells": tm["n_align_cells"],
                             "refused_by_packed_kernel": tm.get("n_align_refused", 0)}}
        sah_res = (cls, strand, st, tm, acc)
        sah["alignment"].update(arena_bytes=tm.get("align_arena_bytes"), aligner_version=tm.get("align_version"))
        os.environ.pop("IOC_ALIGN_ARENA", None)     # core / cli / merge below: the library default

    # ---- every rank's own result against the committed oracle digest of ITS batch (tests/golden, tools/gen_golden.py) ----
    golden = {}
    try:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_assignments.json")))
    except Exception:
        pass
    from isonclust2_amd.digest import fnv1a
    rank_parity = {}
    lean_res = sah_lean.pop("_res") if sah_lean else None
    for mode_name, res in (("fast", fast_res if want_fast else None), ("sahlin", sah_res if want_sahlin else None),
                           ("sahlin_other_aligner", lean_res)):
        if res is None or a.config != "config2":
            continue
        g = golden.get(f"config2:{seed}" + ("" if mode_name == "fast" else ":sahlin"))  # (the other aligner: the same golden)
        ok = -1 if g is None else int(f"{fnv1a(res[0], res[1]):016x}" == g["fnv1a"] and res[2]["n_clusters"] == g["clusters"])
        if dist is not None:
            t = torch.tensor([ok], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        rank_parity[mode_name] = {1: "every rank's assignments equal the oracle's digest of its batch", 0: "MISMATCH on at least one rank",
                                  -1: "no golden for at least one rank's batch"}[ok]
        if mode_name != "fast" and ok == 1:
            # (the alignment verdicts behind these digests come from the oracle's own sg_trace; parasail is absent
            # from the reference tree, so its traceback tie-breaking is unpinned: parity with the oracle, not with parasail)
            rank_parity[mode_name] += " (oracle's aligner: tie rules of the traceback are this build's, parasail's are unpinned)"

    # ---- roofline inputs that need the resident fast-mode clustering (rank 0) ----
    roof = roof_aln = None
    fcls = fstrand = None
    if rank == 0 and fast is not None:
        ctx.set_params(api.default_params(k, w, "fast"))
        fcls, fstrand, fst = ctx.cluster_resident()     # untimed: leaves the fast-mode clustering on the device
        M = fast_res[3]["n_minimizers"]
        # H = postings the reference's GetMinimizerHits traverses on this batch, counted on the device
        # from the final clustering (instrumentation launch, untimed); C_s = survivor candidates
        H = ctx.count_reference_postings()
        Cs = fast_res[3]["n_mapped_evals"]
        roof_counts = (M, H, Cs, fst)

    # ---- core region (every rank: its ClusteredBatch is also what the merge exchanges) ----
    core = {}
    cb_merge = None
    if not a.no_core or (world > 1 and not a.no_merge):
        if want_fast:
            core["fast"], cb = core_region(ctx, api, pipeline, sb, k, w, "fast", 1 if a.no_core else 3)
            cb_merge = cb
        if want_sahlin:
            core["sahlin"], cb = core_region(ctx, api, pipeline, sb, k, w, "sahlin", 1 if a.no_core else 3)
            cb_merge = cb
        from isonclust2_amd import dist as idist_
        for m in core:
            core[m]["reads_per_s"] = rs.n / (core[m]["ms_min"] * 1e-3)
            # the job: every rank's batch, the slowest rank's time (ranks run their core regions side by side)
            core[m]["ms_min_max_over_ranks"] = idist_.max_over_ranks(core[m]["ms_min"], dist)
            core[m]["reads_per_s_job"] = total_reads / (core[m]["ms_min_max_over_ranks"] * 1e-3)

    # ---- merge of the ranks' batches (configs[3]): the one exchange step of the path ----
    merge = None
    if world > 1 and not a.no_merge:
        from isonclust2_amd import dist as idist
        mode = "sahlin" if want_sahlin else "fast"
        try:
            merge = idist.timed_merge(ctx, api.default_params(k, w, mode), cb_merge, dist, torch, dev)
        except Exception as e:  # noqa: BLE001  (the merge runs after the timed region: the bench line still goes out)
            merge = {"error": f"{type(e).__name__}: {e}"[:400]}
        merge["mode"] = mode
        # (IOC_BENCH_NATIVE_ANY_BACKEND=1: rehearsal of this leg's control flow with ranks that share a card — RCCL then refuses
        # the communicator and the leg must report that on every rank and let the line go out)
        if "error" not in merge and not a.no_native_merge and (a.backend == "nccl" or os.environ.get("IOC_BENCH_NATIVE_ANY_BACKEND") == "1"):
            merge["native_rccl"] = native_merge_leg(ctx, api, pipeline, idist, dist, torch, sb, cb_merge, k, w, mode, merge)
        g4 = golden.get(f"config4:{mode}")
        if "error" not in merge and g4 is not None and not a.same_seed and a.config == "config2" and 2 <= world <= len(g4["seeds"]):
            # the left fold of the first `world` batches is a prefix of the 8-batch fold: cluster counts after every step
            # are in the golden; the digest only for all 8
            merge["golden"] = {"clusters_out_expected": g4["steps"][world - 2]["clusters_after"],
                               "clusters_match": merge["clusters_out"] == g4["steps"][world - 2]["clusters_after"]}
            if world == len(g4["seeds"]):
                merge["golden"]["digest_match"] = merge["fnv1a"] == g4["fnv1a"]

    if rank == 0:
        # ---- fast mode (configs[1]): HBM roofline of the scoring kernels + full-batch CPU baseline / parity ----
        if fast is not None:
            M, H, Cs, fst = roof_counts
            h_source = "device count (ioc_count_reference_postings)"
            if single:
                tmin, times, ost, mism, core_id = cpu_baseline_fast(rs, order, fcls, fstrand, k, w, a.cpu_runs, a.cpu_budget)
                fast["cpu_baseline"] = {"value": rs.n / tmin, "unit": "reads/s", "cores": 1, "kind": "port",
                                        "cpu_model": cpu_model(), "pinned_core": core_id, "runs_s": [round(t, 2) for t in times],
                                        "sample": f"the full {a.config} batch ({rs.n} reads), fast mode, ClusterSortedReads "
                                                  f"region, oracle -O3 -DNDEBUG -msse3, min of {len(times)} run(s), 1 pinned core"}
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
                # `traffic` = HBM bytes per launch from the PMC passes of tools/collect_profile.sh (FETCH_SIZE / WRITE_SIZE need their own
                # rocprofv3 runs: they cannot be read inside this process); the file says which build it was measured on
                tfile = os.path.join(ROOT, "profiles", "k_score_traffic.json")
                if os.path.exists(tfile):
                    try:
                        from isonclust2_amd.digest import source_stamp
                        tj = json.load(open(tfile))
                        roof["traffic"] = tj.get("hbm_bytes_per_launch")
                        st_ = (tj.get("stamp") or {}).get("sources_sha12")
                        roof["traffic_source"] = {"file": "profiles/k_score_traffic.json", "measured_on_sources": st_, "these_sources": source_stamp(),
                                                  "same_build": st_ == source_stamp()}
                    except Exception:
                        pass
        # ---- sahlin mode (configs[2]): the step is dominated by the alignment fallback's forward DP, an
        # integer-VALU kernel (no HBM traffic to speak of, no MFMA): its issue-rate roofline beside the HBM one ----
        if sah is not None:
            # (the cells the forward pass COMPUTED: version 2 leaves out the tiles outside its certified corridor — the matrix cells of
            # the batch, the reference's figure, are in `alignment.cells`)
            cells = sah_res[3].get("n_align_cells_computed") or sah_res[3]["n_align_cells"]
            cells_matrix = sah_res[3]["n_align_cells"]
            ms_fwd = sah_res[4]["ms_align_fwd"]
            if cells and ms_fwd > 0:
                peak, src = valu_peak()
                v2 = sah_res[3].get("align_version") == 2
                # useful lane-operations per cell: version 1 — add with byte select, max3, sub, max, max; version 2 — 7 per
                # column PAIR (perm, add, 4 packed max, sub), two cells each
                per_cell = 3.5 if v2 else ALIGN_VALU_PER_CELL
                ach = cells * per_cell / (ms_fwd * 1e-3) / 1e12
                rates, _ = valu_rates()
                mixr = next((v for kk, v in rates.items() if kk.startswith("aligner v2" if v2 else "aligner cell mix")), None)
                roof_aln = {"bound": "valu-int", "kernel": "k_fwd2 (two pairs per wave, 16-bit halves)" if v2 else "k_align_fwd", "achieved": ach, "peak": peak,
                            "peak_source": src, "unit": "T lane-op/s", "frac": ach / peak, "kernel_ms": ms_fwd,
                            "frac_of_guide_2cycle_issue": ach / (1024 * 32 * 2.4e9 / 1e12),
                            "frac_of_own_instruction_mix": (ach / mixr) if mixr else None, "instruction_mix_rate": mixr,
                            "cells": cells, "cells_of_the_matrices": cells_matrix, "valu_per_cell": per_cell,
                            "gcells_per_s": cells / (ms_fwd * 1e-3) / 1e9, "matrix_gcells_per_s": cells_matrix / (ms_fwd * 1e-3) / 1e9,
                            "note": "achieved / frac count the cells the kernel computed; the corridor's skipped tiles are work avoided, not work done"}
            if single:
                tmin, times, ost, mism, ns, core_id = cpu_baseline_sahlin(lambda: api.Context(dev_index), api, pipeline, rs, order, k, w,
                                                                         a.cpu_sample, a.cpu_runs, a.cpu_budget)
                sah["cpu_baseline"] = {"value": ns / tmin, "unit": "reads/s", "cores": 1, "kind": "port",
                                       "cpu_model": cpu_model(), "pinned_core": core_id, "runs_s": [round(t, 2) for t in times],
                                       "aligner": "the oracle's own SCALAR semi-global aligner with full traceback (oracle.cpp sg_trace) — "
                                                  "not parasail's striped SIMD scan (absent from the reference tree)",
                                       "sample": f"first {ns} reads of the sorted {a.config} batch, sahlin mode, oracle -O3 -DNDEBUG "
                                                 f"-msse3, min of {len(times)} run(s), 1 pinned core; {ost['aln_invoked']} reads reach the fallback"}
                sah["parity"] = {"entries": ns, "mismatches": mism, "oracle_aln_invoked": ost["aln_invoked"]}
                # the same batch if its alignments cost a SIMD score pass each instead of the scalar aligner: a LOWER bound of the CPU's
                # time (no traceback table, no 32-bit second pass), hence an upper bound of what one core could reach
                try:
                    sp = cpu_simd_score_pass(rs, order)
                    fast_s = rs.n / fast["cpu_baseline"]["value"] if fast and fast.get("cpu_baseline") else None
                    aln_s = sah["alignment"]["cells"] / (sp["gcups"] * 1e9)
                    sah["cpu_baseline_simd_bound"] = {
                        "value": rs.n / (aln_s + (fast_s or 0.0)), "unit": "reads/s", "cores": 1, "kind": "port",
                        "score_pass_gcups": sp["gcups"], "pairs_timed": sp["pairs"], "seconds_timed": sp["seconds"],
                        "sample": f"upper bound for one core: the batch's {sah['alignment']['pairs']} alignments ({sah['alignment']['cells']:.3g} cells) at the rate of a "
                                  "16-bit SSE2 striped SCORE pass (oracle/sg_striped.cpp; no traceback table, no 32-bit re-run of saturated pairs: "
                                  "parasail's sg_trace_scan_16 does both)" + (f" + the fast-mode oracle time of the batch ({fast_s:.1f} s)" if fast_s else "")}
                except Exception as e:  # noqa: BLE001
                    sah["cpu_baseline_simd_bound"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        cpu_node = None
        if single and not a.no_cpu_node:
            cpu_node = cpu_baseline_node(a.config, k, w)
        cli = None
        if world == 1 and not a.no_cli:
            cli = cli_region(rs, [m for m, on in (("sahlin", want_sahlin), ("fast", want_fast)) if on])
        head, head_mode = (sah, "sahlin") if sah is not None else (fast, "fast")
        out = {
            "metric": f"reads/s clustered ({head_mode} mode, k=11 w=15)",
            "value": head["value"], "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "value_region": "resident: one pass of the hot path (ioc_cluster_resident: index build, gap-bound table, scoring, resolve, "
                            "alignment rounds) over a sorted batch already in HBM; nothing computed is kept from one step to the next — "
                            "the aligner's corridor MODEL is (six sums per gap-open class over the pairs the context has aligned before: it "
                            "plans which tiles a step computes, the certificate decides what stands; the first step of a context plans from an error-model prior); "
                            "`core` = host arrays -> host results incl. PCIe, `cli` = whole cluster process incl. .cer I/O",
            "config": {"workload": f"{a.config} = BASELINE.json configs[{2 if head_mode == 'sahlin' else 1}]: {rs.tag}; one sorted "
                                   "3000-read / 50 Mb batch per GPU, minimizer SoA and sequences resident in HBM",
                       "mode": head_mode, "k": k, "w": w, "reads_per_gpu": rs.n, "minimizers_per_gpu": int(n_min),
                       "parallelism": f"batch-shard x{world}, no data-path collective in the timed region",
                       "rank_batches": "every rank its own copy of the seed-1 batch" if a.same_seed else "rank r: the batch of seed 1 + r (different batches)",
                       "identical_batches": bool(a.same_seed) and world > 1},
            "phase_ms": head["phase_ms"],
            "roofline": roof if roof is not None else roof_aln,
            "cpu_baseline": head.get("cpu_baseline"), "cpu_baseline_simd_bound": head.get("cpu_baseline_simd_bound"), "parity": head.get("parity"),
            "core": core or None, "cli": cli, "merge": merge, "golden_parity": rank_parity or None,
            "cpu_baseline_node": cpu_node,
        }
        if core.get(head_mode):
            # SURVEY §8(d)'s *core* region beside the resident `value` (bench contract: `value` has its inputs resident in HBM, a
            # PCIe-inclusive rate is never `value`): host arrays -> assignments + MinDB in host arrays, every rank its batch
            out["value_core"] = core[head_mode]["reads_per_s_job"]
            out["value_core_region"] = ("core: ioc_cluster_merge + ioc_index_export from host arrays, H2D of the minimizer SoA (+ sequences) and "
                                        "every D2H inside; min of 3 runs per rank, all ranks' reads / the slowest rank's time")
        if head_mode == "sahlin":
            out["alignment"] = sah["alignment"]
            out["config"]["aligner"] = ("IOC_ALIGN_ARENA=fat: version 1, fine checkpoints, arena kept resident" if a.aligner == "fat"
                                        else "library default: version 2, two pairs per wave, coarse checkpoints")
            out["alignment_other_aligner"] = sah_lean
            out["roofline_align"] = roof_aln
            if fast is not None:
                out["fast_mode"] = fast       # BASELINE.json configs[1] on the same batch
        print(json.dumps(out), flush=True)
    # a rank whose native merge leg ran out of time is still inside RCCL in the abandoned thread: no orderly shutdown to be had,
    # on ANY rank (the others would wait in the barrier below for a peer that has left) — the ranks agree on it first
    stuck = 1.0 if (merge and isinstance(merge.get("native_rccl"), dict) and merge["native_rccl"].get("timed_out")) else 0.0
    if dist is not None:
        try:
            from isonclust2_amd import dist as idist_x
            stuck = idist_x.max_over_ranks(stuck, dist)
        except Exception:  # noqa: BLE001
            stuck = 1.0
    if stuck:
        sys.stdout.flush()
        # (the line is out, with `merge.native_rccl.timed_out` in it; the abandoned thread sits in a collective, so no orderly way
        # out on any rank.  --strict: the exit code says so too — 3 —; without it a leg that runs AFTER the timed region does not
        # take the measured line down with it)
        os._exit(3 if a.strict else 0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
