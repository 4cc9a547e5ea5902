"""Plain-Python statements of the sequence-to-graph recurrence and of a path's score, shared by the POA tests (the oracle's
scalar POA on the CPU, the product's engine on the GPU) and by tools/fuzz_poa.py."""
SC = dict(m=4, n=-8, g=-8, e=-4, q=-20, c=-1)   # src/main.cpp:285-290
NEG = -10 ** 9


def _ref_score(bases, rank, ef, et, seq):
    """Local sequence-to-graph alignment, convex gap = best of two affine pieces (plain loops)."""
    n, L = len(bases), len(seq)
    preds = {v: [] for v in range(n)}
    for a, b in zip(ef, et):
        preds[int(b)].append(int(a))
    H = {-1: [0] * (L + 1)}
    F1 = {-1: [NEG] * (L + 1)}
    F2 = {-1: [NEG] * (L + 1)}
    best = 0
    for v in [int(x) for x in rank]:
        ps = preds[v] or [-1]
        h, f1, f2 = [0] * (L + 1), [NEG] * (L + 1), [NEG] * (L + 1)
        e1 = e2 = NEG
        for j in range(L + 1):
            f1[j] = max(max(H[p][j] + SC["g"], F1[p][j] + SC["e"]) for p in ps)
            f2[j] = max(max(H[p][j] + SC["q"], F2[p][j] + SC["c"]) for p in ps)
            if j == 0:
                h[j] = 0
                continue
            e1 = max(h[j - 1] + SC["g"], e1 + SC["e"])
            e2 = max(h[j - 1] + SC["q"], e2 + SC["c"])
            s = SC["m"] if bases[v] == seq[j - 1] else SC["n"]
            h[j] = max(0, max(H[p][j - 1] for p in ps) + s, f1[j], f2[j], e1, e2)
            best = max(best, h[j])
        H[v], F1[v], F2[v] = h, f1, f2
    return best


def _path_score(bases, ef, et, seq, nodes, pos):
    """Score of an alignment path, checking that it IS a walk through the graph and the read."""
    edges = set(zip(ef.tolist(), et.tolist()))
    total, run_kind, run_len = 0, None, 0
    last_node, last_pos = None, None

    def close():
        nonlocal total, run_kind, run_len
        if run_len:
            total += max(SC["g"] + (run_len - 1) * SC["e"], SC["q"] + (run_len - 1) * SC["c"])
        run_kind, run_len = None, 0

    for v, p in zip(nodes.tolist(), pos.tolist()):
        if v >= 0:
            assert last_node is None or (last_node, v) in edges, "consecutive path nodes must be joined by an edge"
            last_node = v
        if p >= 0:
            assert last_pos is None or p == last_pos + 1, "read positions must be consecutive"
            last_pos = p
        kind = "diag" if v >= 0 and p >= 0 else ("vert" if v >= 0 else "horz")
        if kind == "diag":
            close()
            total += SC["m"] if bases[v] == seq[p] else SC["n"]
        else:
            if kind != run_kind:
                close()
            run_kind, run_len = kind, run_len + 1
    close()
    return total


def mutate(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            out.append(rng.choice(b"ACGT"))
        elif x < 2 * rate / 3:
            continue
        elif x < rate:
            out += bytes([ch, rng.choice(b"ACGT")])
        else:
            out.append(ch)
    return bytes(out)


def random_addition(rng, truth, t):
    """the read shapes tools/fuzz_poa.py feeds a graph: noisy copy, fragment, long deletion / insertion, unrelated head or tail"""
    r = mutate(rng, truth, rng.choice([0.02, 0.1, 0.25]))
    kind = rng.randint(0, 6)
    if kind == 0 and len(r) > 30:      # fragment
        a = rng.randint(0, len(r) // 2)
        r = r[a:a + rng.randint(10, len(r) - a)]
    elif kind == 1:                    # long deletion (edges spanning many rows)
        a = rng.randint(0, max(1, len(r) - 60))
        r = r[:a] + r[a + rng.randint(17, 60):]
    elif kind == 2:                    # long insertion
        a = rng.randint(0, len(r))
        r = r[:a] + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(17, 70))) + r[a:]
    elif kind == 3:                    # unrelated head / tail
        r = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(5, 40))) + r + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 40)))
    return r
