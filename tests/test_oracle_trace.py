"""The oracle's intermediate tables (SURVEY §8(c) golden item 3): candidate tables at chosen loop indices and every
getMappedRatio call — internal consistency here (CPU); tests/test_gpu_candidates.py compares the device's tables
with them."""
import numpy as np

from isonclust2_amd import synth
from oracle import pyoracle as po
from tests.helpers import oracle_entry_assignments, oracle_sorted_batch


def test_trace_rows_are_the_hit_map_of_the_loop(kat):
    rs = synth.generate_config("config1", seed=1)
    B, view = oracle_sorted_batch(rs)
    entries = [3, 57, 120, 250, 333, 499]
    po.trace_set(entries, mapped_calls=True)
    try:
        cls, strand, st = oracle_entry_assignments(B, view)
        rows, calls = po.trace_rows(), po.trace_mapped_calls()
    finally:
        po.trace_set(())
    assert len(calls["entry"]) == st["mapped_calls"]
    assert set(rows["entry"].tolist()) <= set(entries) and len(rows["entry"]) > 0
    for e in entries:
        m = rows["entry"] == e
        if not m.any():
            continue
        # SortMinimizerHits: Size descending in order_pos, every (cls, strand) once, clusters all older than the entry
        o = np.argsort(rows["order_pos"][m])
        sz = rows["size"][m][o]
        assert np.all(sz[:-1] >= sz[1:])
        keys = set(zip(rows["cls"][m].tolist(), rows["strand"][m].tolist()))
        assert len(keys) == int(m.sum())
        assert set(rows["strand"][m].tolist()) <= {1, -1}
        # a walked row is one of the logged getMappedRatio calls with the same totalMapped
        for c, s, t, w in zip(rows["cls"][m], rows["strand"][m], rows["total_mapped"][m], rows["walked"][m]):
            cm = (calls["entry"] == e) & (calls["cls"] == c) & (calls["strand"] == s)
            assert bool(cm.any()) == bool(w)
            if w:
                assert calls["total"][cm][0] == t
    # every call: ratio = total / length in the reference's double arithmetic
    assert np.array_equal(calls["ratio"], calls["total"].astype(np.float64) / calls["hpc_len"].astype(np.float64))
    # the decision of a traced entry is the first walked row passing the threshold
    for e in entries:
        cm = calls["entry"] == e
        passing = [(c, s) for c, s, r in zip(calls["cls"][cm], calls["strand"][cm], calls["ratio"][cm]) if np.float32(r) >= 0.65]
        if passing:
            assert (cls[e], strand[e]) == (passing[0][0], passing[0][1])
