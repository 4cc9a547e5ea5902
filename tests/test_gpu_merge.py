"""GPU parity of the merge path (`cluster -l L -r R`, src/cluster.cpp:67-322 with two real batches)
and of the exported MinDB (AddMinimizers, src/minimizer.cpp:31-42) against the oracle."""
import numpy as np
import pytest

from isonclust2_amd import api, pipeline, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _batches(rs, nb, k=11, w=15):
    """Globally score-sorted reads cut into nb consecutive batches (src/main.cpp:149-199)."""
    R = po.ReadSet.from_flat(rs.seq, rs.qual, rs.offs)
    R.score_sort(k, w)
    p = po.default_params(k, w)
    cuts = np.linspace(0, rs.n, nb + 1).astype(int)
    obs, sbs = [], []
    for b in range(nb):
        B = po.Batch(R, int(cuts[b]), int(cuts[b + 1]) - 1, p, batch_nr=b)
        info, off_f, off_r, mn, ps = B.minimizer_soa()
        view = dict(off_fwd=off_f, off_rev=off_r, min_val=mn, min_pos=ps, raw_len=info["raw_len"],
                    hpc_len=info["hpc_len"], score=info["score"], raw_err=info["raw_err"],
                    hpc_err=info["hpc_err"], state=info["state"].astype(np.uint8), min_qual=p.min_qual)
        obs.append(B)
        sbs.append(pipeline.SortedBatch(view=view, read_ids=info["orig"].astype(np.int64), batch_nr=b,
                                        batch_start=int(cuts[b]), batch_end=int(cuts[b + 1]) - 1))
    return obs, sbs


def _same_index(cb, B):
    keys, offs, post = B.index()
    assert np.array_equal(cb.mindb[0], keys)
    assert np.array_equal(cb.mindb[1], offs)
    assert np.array_equal(cb.mindb[2], post)


@pytest.mark.parametrize("cfg,seed", [("config1", 1), ("short_dup", 2)])
def test_index_export_equals_oracle_mindb(ctx, cfg, seed):
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, 1)
    obs[0].cluster(mode="fast")
    cb = pipeline.cluster_single(ctx, api.default_params(11, 15, "fast"), sbs[0])
    _same_index(cb, obs[0])
    ocl, ost = obs[0].assignments(rs.n)
    cls, strand = cb.assignments(rs.n)
    assert np.array_equal(cls, ocl) and np.array_equal(strand, ost)


def test_distinct_by_radix_sort_equals_the_bitonic_network(ctx, monkeypatch):
    """k_distinct_radix (reads of up to 8192 forward minimizers) and k_distinct (longer ones; IOC_DISTINCT_BITONIC=1 forces it)
    feed the same index: same assignments, same exported MinDB."""
    rs = synth.generate_config("config1", seed=4)
    _, sbs = _batches(rs, 1)
    p = api.default_params(11, 15, "fast")
    a = pipeline.cluster_single(ctx, p, sbs[0])
    monkeypatch.setenv("IOC_DISTINCT_BITONIC", "1")
    b = pipeline.cluster_single(ctx, p, sbs[0])
    assert a.n_clusters == b.n_clusters
    for x, y in zip(a.mindb, b.mindb):
        assert np.array_equal(x, y)
    ca, sa = a.assignments(rs.n)
    cb, sb_ = b.assignments(rs.n)
    assert np.array_equal(ca, cb) and np.array_equal(sa, sb_)


def test_index_export_ordered_on_the_device_equals_the_host_ordered_one(ctx, monkeypatch):
    """ioc_index_export orders the keys on the device (ioc_sort.hip); IOC_EXPORT_HOST_ORDER=1 is the host's std::sort
    of round 2's first version: same CSR, also for a merge (left lists + right lists in one index)."""
    rs = synth.generate_config("config1", seed=3)
    _, sbs = _batches(rs, 2)
    p = api.default_params(11, 15, "fast")

    def run():
        left = pipeline.cluster_single(ctx, p, sbs[0])
        both = pipeline.cluster_merge(ctx, p, left, pipeline.cluster_single(ctx, p, sbs[1]))
        return left.mindb, both.mindb

    dev = run()
    monkeypatch.setenv("IOC_EXPORT_HOST_ORDER", "1")
    host = run()
    for a, b in zip(dev, host):
        assert len(a[0]) > 1000
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y)


@pytest.mark.parametrize("cfg,seed,nb", [("config1", 1, 2), ("config1", 2, 4), ("short_dup", 1, 3)])
def test_left_fold_merge_equals_oracle(ctx, cfg, seed, nb):
    """((b0 + b1) + b2) + ... as in the reference README's example fold."""
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, nb)
    p = api.default_params(11, 15, "fast")
    for B in obs:
        B.cluster(mode="fast")
    cbs = [pipeline.cluster_single(ctx, p, sb) for sb in sbs]
    for cb, B in zip(cbs, obs):
        _same_index(cb, B)
    left_o, left = obs[0], cbs[0]
    for b in range(1, nb):
        left_o.cluster(right=obs[b], mode="fast")
        left = pipeline.cluster_merge(ctx, p, left, cbs[b])
        ocl, ost = left_o.assignments(rs.n)
        cls, strand = left.assignments(rs.n)
        bad = np.nonzero((cls != ocl) | (strand != ost))[0]
        assert len(bad) == 0, (b, len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
        assert left.n_clusters == left_o.n_clusters()
        _same_index(left, left_o)


@pytest.mark.parametrize("cfg,seed,nb,mode", [("config1", 1, 4, "fast"), ("short_dup", 1, 3, "fast"), ("config1", 2, 3, "sahlin"),
                                              ("short_dup", 2, 4, "sahlin")])
def test_one_pass_merge_of_all_batches_equals_the_oracle_fold(ctx, cfg, seed, nb, mode):
    """dist.merge_all: ((b0 + b1) + b2) ... computed as ONE pass of the merge path over the concatenated right
    representatives — against the oracle folding the batches one merge at a time (cluster.cpp:67-322), on batches
    that share transcripts (joins across batches, strand flips, ties; sahlin: representatives aligned against
    representatives)."""
    from isonclust2_amd import dist as idist
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, nb)
    for sb, B in zip(sbs, obs):   # raw sequences for the alignment fallback
        info = B.entry_info()
        seqs = [rs.read(int(i))[0] for i in info["orig"]]
        off = np.zeros(len(seqs) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in seqs])
        sb.view.update(raw_seq=b"".join(seqs), raw_off=off)
    p = api.default_params(11, 15, mode)
    for B in obs:
        B.cluster(mode=mode)
    cbs = [pipeline.cluster_single(ctx, p, sb) for sb in sbs]
    left_o = obs[0]
    aln = 0
    for b in range(1, nb):
        st = left_o.cluster(right=obs[b], mode=mode)
        aln += st["aln_invoked"]
    merged = idist.merge_all(ctx, p, cbs)
    ocl, ost = left_o.assignments(rs.n)
    cls, strand = merged.assignments(rs.n)
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, (len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
    assert merged.n_clusters == left_o.n_clusters()
    _same_index(merged, left_o)
    assert merged.stats["n_aln_invoked"] == aln
    # and the packed record round-trips (what the all-gather ships)
    g = idist.unpack_clustered(idist.pack_clustered(cbs[1]))
    assert np.array_equal(g.rep_view["min_val"], cbs[1].rep_view["min_val"]) and g.rep_seq == cbs[1].rep_seq
    assert np.array_equal(g.rep_off, cbs[1].rep_off) and np.array_equal(g.member_read, cbs[1].member_read)


@pytest.mark.parametrize("cfg,seed,nb,mode", [("config1", 3, 4, "fast"), ("short_dup", 3, 3, "sahlin")])
def test_device_resident_records_merge_equals_the_oracle_fold(ctx, cfg, seed, nb, mode):
    """The merge as the ranks of a node run it (dist.merge_all_device), minus the collective: every batch's
    representative records are gathered ON THE DEVICE out of the batch's query arrays (ioc_gather_records_device),
    the buffers are laid side by side as an all-gather would leave them, and the one-pass merge uses them in place
    (ioc_batch_view::minimizers_on_device, ::is_cluster) — against the oracle folding the batches one merge at a time."""
    import torch
    from isonclust2_amd import dist as idist
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, nb)
    for sb, B in zip(sbs, obs):
        info = B.entry_info()
        seqs = [rs.read(int(i))[0] for i in info["orig"]]
        off = np.zeros(len(seqs) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in seqs])
        sb.view.update(raw_seq=b"".join(seqs), raw_off=off)
    p = api.default_params(11, 15, mode)
    dev = torch.device("cuda", 0)
    parts, metas = [], []
    for sb, B in zip(sbs, obs):
        B.cluster(mode=mode)
        cb = pipeline.cluster_single(ctx, p, sb)
        mins, poss = idist.gather_local(ctx, cb, torch, dev)
        # the gathered lists are the host records' lists, word for word
        assert np.array_equal(mins.cpu().numpy().view(np.uint32), cb.rep_view["min_val"])
        assert np.array_equal(poss.cpu().numpy().view(np.uint32), cb.rep_view["min_pos"])
        parts.append((mins, poss))
        metas.append(idist.unpack_clustered(idist.pack_clustered(cb, with_minimizers=False)))
    first = pipeline.cluster_single(ctx, p, sbs[0])
    pipeline.cluster_single(ctx, p, sbs[1])
    with pytest.raises(RuntimeError):       # the context has moved on: an older batch can no longer be gathered
        idist.gather_local(ctx, first, torch, dev)
    cap = max(int(m.numel()) for m, _ in parts) + 5
    pad = lambda t: torch.cat([t, torch.full((cap - int(t.numel()),), 0x7FFFFFFF, dtype=torch.int32, device=dev)])
    recv_min = torch.cat([pad(m) for m, _ in parts])
    recv_pos = torch.cat([pad(q) for _, q in parts])
    merged = idist.merge_gathered(ctx, p, metas, recv_min, recv_pos, cap, export_mindb=True)
    left_o = obs[0]
    for b in range(1, nb):
        left_o.cluster(right=obs[b], mode=mode)
    ocl, ost = left_o.assignments(rs.n)
    cls, strand = merged.assignments(rs.n)
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, (len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
    assert merged.n_clusters == left_o.n_clusters()
    _same_index(merged, left_o)


@pytest.mark.parametrize("cfg,seed,mode,chunk", [("config1", 5, "fast", "37"), ("short_dup", 5, "fast", "200"), ("config1", 6, "sahlin", "61"),
                                                 ("short_dup", 6, "furious", "150")])
def test_a_batch_beyond_one_device_pass_runs_in_chunks(ctx, monkeypatch, cfg, seed, mode, chunk):
    """More entries than one device pass takes (131 072; IOC_MERGE_CHUNK forces it small here): ioc_cluster_merge runs the right
    batch chunk by chunk, each chunk against the clusters the chunks before it left — the reference's one loop (cluster.cpp:115).
    Single batches and a left fold whose right batches are chunked too: assignments and the exported MinDB equal the oracle's."""
    rs = synth.generate_config(cfg, seed=seed)
    obs, sbs = _batches(rs, 2)
    for sb, B in zip(sbs, obs):
        info = B.entry_info()
        seqs = [rs.read(int(i))[0] for i in info["orig"]]
        off = np.zeros(len(seqs) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in seqs])
        sb.view.update(raw_seq=b"".join(seqs), raw_off=off)
    p = api.default_params(11, 15, mode)
    for B in obs:
        B.cluster(mode=mode)
    monkeypatch.setenv("IOC_MERGE_CHUNK", chunk)
    cbs = [pipeline.cluster_single(ctx, p, sb) for sb in sbs]
    for cb, B in zip(cbs, obs):
        assert cb.n_clusters == B.n_clusters()
        _same_index(cb, B)
    obs[0].cluster(right=obs[1], mode=mode)
    assert obs[1].n_clusters() > int(chunk) // 4
    monkeypatch.setenv("IOC_MERGE_CHUNK", str(max(7, int(chunk) // 4)))
    merged = pipeline.cluster_merge(ctx, p, cbs[0], cbs[1])
    ocl, ost = obs[0].assignments(rs.n)
    cls, strand = merged.assignments(rs.n)
    bad = np.nonzero((cls != ocl) | (strand != ost))[0]
    assert len(bad) == 0, (len(bad), bad[:5], cls[bad[:5]], ocl[bad[:5]])
    assert merged.n_clusters == obs[0].n_clusters()
    _same_index(merged, obs[0])
