/*
 * poa_oracle.cpp — scalar partial-order alignment behind the five graph operations of the consensus
 * (TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/ and tools/ load it; see oracle.h).
 *
 * What it restates.  The reference keeps one spoa::Graph per cluster (src/serialize.h:21,37) and performs, through
 * spoa 4.0's API: AlignmentEngine::Create(kSW, m 4, n -8, g -8, e -4, q -20, c -1) (src/main.cpp:285-324),
 * engine->Align(seq, graph) + graph->AddAlignment(alignment, seq, weight) (src/consensus.cpp:15-32),
 * graph->sequences().size() (:41, :51, :87), graph->GenerateConsensus() (:91), graph->Clear() + a new graph fed with the
 * representative (ConsPurge, :128-137); a graph seeded with the representative at weight 1 for every new cluster
 * (src/cluster.cpp:200-204).
 *
 * spoa itself is a third-party submodule that is ABSENT from /root/reference (.gitmodules), so this file restates the
 * PUBLISHED algorithm (Lee, Grasso & Sharlow 2002, "Multiple sequence alignment using partial order graphs"; Lee 2003 for the
 * heaviest-bundle consensus; Vaser, Sovic, Nagarajan & Sikic 2017 for spoa's formulation) in the shape spoa 4.0's scalar
 * (SISD) engine and graph give it:
 *   - nodes in creation order; an edge (tail, head) is created once and its weight grows with every sequence that takes it;
 *     consecutive bases contribute weight[i - 1] + weight[i]; in- and out-edge lists keep insertion order;
 *   - mismatching aligned bases become "aligned nodes" of each other (a clique per column);
 *   - topological order by depth-first search from every node in id order, in-edges first, a column's aligned nodes placed
 *     next to each other;
 *   - local (Smith-Waterman) sequence-to-graph DP over the rows in that order; the gap cost is the better of two affine
 *     pieces (g, e) and (q, c): matrices H, F / O (gap that consumes graph nodes), E / Q (gap that consumes read bases); the
 *     best cell is the FIRST maximum in (row, column) order;
 *   - traceback: diagonal move first (predecessors in in-edge order), then the vertical gap (extension of F, opening from H,
 *     extension of O, opening from H; predecessors in in-edge order), then the horizontal gap; a gap that was extended is
 *     followed, inside its own piece, to the cell that opened it: the path's score always equals the best cell's;
 *   - AddAlignment: unaligned prefix and suffix of the read as fresh chains (the suffix BEFORE the aligned part), then the
 *     aligned part base by base;
 *   - consensus: heaviest bundle with branch completion.
 * PARITY UNPINNED: no vector of the reference's tests touches a graph, and the source this was written after is not in the
 * tree.  What this oracle pins is the product's POA engine (isonclust2_amd/csrc/ioc_poa.hip) against an INDEPENDENT scalar
 * implementation of the same published algorithm, tie rules included — "parity with the oracle's POA; spoa unpinned".
 * Written without consulting ioc_poa.hip's tie choices.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace {

constexpr int32_t kNegInf = INT32_MIN / 2;

struct Edge {
    int tail, head;
    int64_t weight;
};

struct Node {
    uint8_t code;            // the letter itself
    std::vector<int> in, out;  // edge ids, insertion order
    std::vector<int> aligned;  // node ids, insertion order
};

struct Graph {
    std::vector<Node> nodes;
    std::vector<Edge> edges;
    std::vector<int> rank_to_node;
    int num_sequences = 0;

    int add_node(uint8_t code)
    {
        nodes.push_back(Node{code, {}, {}, {}});
        return int(nodes.size()) - 1;
    }
    void add_edge(int tail, int head, int64_t w)
    {
        for (int e : nodes[size_t(tail)].out)
            if (edges[size_t(e)].head == head) {
                edges[size_t(e)].weight += w;
                return;
            }
        edges.push_back(Edge{tail, head, w});
        nodes[size_t(tail)].out.push_back(int(edges.size()) - 1);
        nodes[size_t(head)].in.push_back(int(edges.size()) - 1);
    }
    // a fresh chain for seq[begin, end): first node id, -1 if empty
    int add_sequence(const std::string& s, const std::vector<uint32_t>& w, uint32_t begin, uint32_t end)
    {
        if (begin == end) return -1;
        int prev = -1, first = -1;
        for (uint32_t i = begin; i < end; ++i) {
            const int cur = add_node(uint8_t(s[i]));
            if (first < 0) first = cur;
            if (prev >= 0) add_edge(prev, cur, int64_t(w[i - 1]) + int64_t(w[i]));  // both bases contribute
            prev = cur;
        }
        return first;
    }
    void topological_sort()
    {
        rank_to_node.clear();
        const size_t N = nodes.size();
        std::vector<uint8_t> marks(N, 0);
        std::vector<uint8_t> ignored(N, 0);
        std::vector<int> stack;
        for (size_t start = 0; start < N; ++start) {
            if (marks[start] != 0) continue;
            stack.push_back(int(start));
            while (!stack.empty()) {
                const int cur = stack.back();
                bool valid = true;
                if (marks[size_t(cur)] != 2) {
                    for (int e : nodes[size_t(cur)].in) {
                        const int t = edges[size_t(e)].tail;
                        if (marks[size_t(t)] != 2) {
                            stack.push_back(t);
                            valid = false;
                        }
                    }
                    if (!ignored[size_t(cur)]) {
                        for (int a : nodes[size_t(cur)].aligned)
                            if (marks[size_t(a)] != 2) {
                                stack.push_back(a);
                                ignored[size_t(a)] = 1;
                                valid = false;
                            }
                    }
                    if (valid) {
                        marks[size_t(cur)] = 2;
                        if (!ignored[size_t(cur)]) {
                            rank_to_node.push_back(cur);
                            for (int a : nodes[size_t(cur)].aligned) rank_to_node.push_back(a);
                        }
                    } else {
                        marks[size_t(cur)] = 1;
                    }
                }
                if (valid) stack.pop_back();
            }
        }
    }
    // alignment: (node id or -1, read position or -1), in read order
    void add_alignment(const std::vector<std::pair<int, int>>& aln, const std::string& s, const std::vector<uint32_t>& w)
    {
        const uint32_t len = uint32_t(s.size());
        if (len == 0) return;
        if (aln.empty()) {
            add_sequence(s, w, 0, len);
            ++num_sequences;
            topological_sort();
            return;
        }
        std::vector<uint32_t> valid;
        for (const auto& a : aln)
            if (a.second != -1) valid.push_back(uint32_t(a.second));
        const size_t before = nodes.size();
        add_sequence(s, w, 0, valid.front());
        int prev = before == nodes.size() ? -1 : int(nodes.size()) - 1;
        const int last = add_sequence(s, w, valid.back() + 1, len);
        for (const auto& a : aln) {
            if (a.second == -1) continue;
            const uint8_t code = uint8_t(s[size_t(a.second)]);
            int cur = -1;
            if (a.first == -1) {
                cur = add_node(code);
            } else {
                const int j = a.first;
                if (nodes[size_t(j)].code == code) {
                    cur = j;
                } else {
                    for (int k : nodes[size_t(j)].aligned)
                        if (nodes[size_t(k)].code == code) {
                            cur = k;
                            break;
                        }
                    if (cur < 0) {
                        cur = add_node(code);
                        const std::vector<int> col = nodes[size_t(j)].aligned;
                        for (int k : col) {
                            nodes[size_t(k)].aligned.push_back(cur);
                            nodes[size_t(cur)].aligned.push_back(k);
                        }
                        nodes[size_t(j)].aligned.push_back(cur);
                        nodes[size_t(cur)].aligned.push_back(j);
                    }
                }
            }
            if (prev >= 0) add_edge(prev, cur, int64_t(w[size_t(a.second) - 1]) + int64_t(w[size_t(a.second)]));
            prev = cur;
        }
        if (last >= 0) add_edge(prev, last, int64_t(w[valid.back()]) + int64_t(w[valid.back() + 1]));
        ++num_sequences;
        topological_sort();
    }
    int branch_completion(uint32_t rank, std::vector<int64_t>& scores, std::vector<int>& pred) const
    {
        const int start = rank_to_node[rank];
        for (int e : nodes[size_t(start)].out)
            for (int f : nodes[size_t(edges[size_t(e)].head)].in)
                if (edges[size_t(f)].tail != start) scores[size_t(edges[size_t(f)].tail)] = -1;
        int best = -1;
        for (uint32_t i = rank + 1; i < rank_to_node.size(); ++i) {
            const int it = rank_to_node[i];
            scores[size_t(it)] = -1;
            pred[size_t(it)] = -1;
            for (int e : nodes[size_t(it)].in) {
                const Edge& ed = edges[size_t(e)];
                if (scores[size_t(ed.tail)] == -1) continue;
                if (scores[size_t(it)] < ed.weight ||
                    (scores[size_t(it)] == ed.weight && scores[size_t(pred[size_t(it)])] <= scores[size_t(ed.tail)])) {
                    scores[size_t(it)] = ed.weight;
                    pred[size_t(it)] = ed.tail;
                }
            }
            if (pred[size_t(it)] >= 0) scores[size_t(it)] += scores[size_t(pred[size_t(it)])];
            if (best < 0 || scores[size_t(best)] < scores[size_t(it)]) best = it;
        }
        return best;
    }
    std::string consensus() const
    {
        if (rank_to_node.empty()) return std::string();
        std::vector<int64_t> scores(nodes.size(), -1);
        std::vector<int> pred(nodes.size(), -1);
        int best = -1;
        for (int it : rank_to_node) {
            for (int e : nodes[size_t(it)].in) {
                const Edge& ed = edges[size_t(e)];
                if (scores[size_t(it)] < ed.weight ||
                    (scores[size_t(it)] == ed.weight && scores[size_t(pred[size_t(it)])] <= scores[size_t(ed.tail)])) {
                    scores[size_t(it)] = ed.weight;
                    pred[size_t(it)] = ed.tail;
                }
            }
            if (pred[size_t(it)] >= 0) scores[size_t(it)] += scores[size_t(pred[size_t(it)])];
            if (best < 0 || scores[size_t(best)] < scores[size_t(it)]) best = it;
        }
        if (!nodes[size_t(best)].out.empty()) {
            std::vector<uint32_t> node_rank(nodes.size(), 0);
            for (uint32_t i = 0; i < rank_to_node.size(); ++i) node_rank[size_t(rank_to_node[i])] = i;
            while (!nodes[size_t(best)].out.empty()) best = branch_completion(node_rank[size_t(best)], scores, pred);
        }
        std::string out;
        while (pred[size_t(best)] >= 0) {
            out.push_back(char(nodes[size_t(best)].code));
            best = pred[size_t(best)];
        }
        out.push_back(char(nodes[size_t(best)].code));
        std::reverse(out.begin(), out.end());
        return out;
    }
};

struct Engine {
    int32_t m, n, g, e, q, c;
    // local alignment of s against G: (node id | -1, read position | -1) in read order; *score = the best cell
    std::vector<std::pair<int, int>> align(const std::string& s, const Graph& G, int32_t* score) const
    {
        std::vector<std::pair<int, int>> aln;
        *score = 0;
        const size_t L = s.size(), W = L + 1, R = G.rank_to_node.size();
        if (R == 0 || L == 0) return aln;
        std::vector<uint32_t> node_rank(G.nodes.size(), 0);
        for (uint32_t i = 0; i < R; ++i) node_rank[size_t(G.rank_to_node[i])] = i;
        std::vector<int32_t> H((R + 1) * W, 0), F((R + 1) * W, kNegInf), E((R + 1) * W, kNegInf), O((R + 1) * W, kNegInf),
            Q((R + 1) * W, kNegInf);
        auto prof = [&](const Node& nd, size_t j) -> int32_t { return nd.code == uint8_t(s[j - 1]) ? m : n; };
        auto pred_rows = [&](const Node& nd) {
            std::vector<size_t> ps;
            for (int ed : nd.in) ps.push_back(size_t(node_rank[size_t(G.edges[size_t(ed)].tail)]) + 1);
            if (ps.empty()) ps.push_back(0);
            return ps;
        };
        int32_t best = 0;
        size_t bi = 0, bj = 0;
        for (size_t r = 0; r < R; ++r) {
            const Node& nd = G.nodes[size_t(G.rank_to_node[r])];
            const size_t i = r + 1;
            const std::vector<size_t> ps = pred_rows(nd);
            int32_t *Hr = &H[i * W], *Fr = &F[i * W], *Er = &E[i * W], *Or = &O[i * W], *Qr = &Q[i * W];
            for (size_t x = 0; x < ps.size(); ++x) {
                const int32_t *Hp = &H[ps[x] * W], *Fp = &F[ps[x] * W], *Op = &O[ps[x] * W];
                for (size_t j = 1; j < W; ++j) {
                    const int32_t f = std::max(Hp[j] + g, Fp[j] + e), o = std::max(Hp[j] + q, Op[j] + c), h = Hp[j - 1] + prof(nd, j);
                    if (x == 0) {
                        Fr[j] = f;
                        Or[j] = o;
                        Hr[j] = h;
                    } else {
                        Fr[j] = std::max(Fr[j], f);
                        Or[j] = std::max(Or[j], o);
                        Hr[j] = std::max(Hr[j], h);
                    }
                }
            }
            for (size_t j = 1; j < W; ++j) {
                Er[j] = std::max(Hr[j - 1] + g, Er[j - 1] + e);
                Qr[j] = std::max(Hr[j - 1] + q, Qr[j - 1] + c);
                Hr[j] = std::max(Hr[j], std::max(std::max(Fr[j], Er[j]), std::max(Or[j], Qr[j])));
                Hr[j] = std::max(Hr[j], 0);
                if (best < Hr[j]) {
                    best = Hr[j];
                    bi = i;
                    bj = j;
                }
            }
        }
        *score = best;
        // ---- traceback ----
        // Order of the tests as in spoa's scalar engine (diagonal, then the vertical gap, then the horizontal one; inside a gap
        // move: extension of the first piece, its opening, extension of the second piece, its opening; predecessors in in-edge
        // order).  An EXTENDED gap is followed back inside the piece it is in until the cell that opened it (extension before
        // opening on ties), so the path's own score always equals the cell's.  (spoa 4.0's scalar engine is recalled to follow an
        // extended gap while EITHER piece looks extended, which can run past the opening; unverifiable here — the source is
        // absent — and not restated.)
        size_t i = bi, j = bj;
        while (H[i * W + j] != 0) {
            const int32_t Hij = H[i * W + j];
            bool found = false;
            int ext_left = 0, ext_up = 0;  // 1: first piece (E / F), 2: second piece (Q / O)
            size_t pi = i, pj = j;
            if (i != 0 && j != 0) {
                const Node& nd = G.nodes[size_t(G.rank_to_node[i - 1])];
                const int32_t mc = prof(nd, j);
                for (size_t p : pred_rows(nd))
                    if (Hij == H[p * W + j - 1] + mc) {
                        pi = p;
                        pj = j - 1;
                        found = true;
                        break;
                    }
            }
            if (!found && i != 0) {
                const Node& nd = G.nodes[size_t(G.rank_to_node[i - 1])];
                for (size_t p : pred_rows(nd)) {
                    int up = 0;
                    if (Hij == F[p * W + j] + e) up = 1;
                    else if (Hij == H[p * W + j] + g) up = 0;
                    else if (Hij == O[p * W + j] + c) up = 2;
                    else if (Hij == H[p * W + j] + q) up = 0;
                    else continue;
                    ext_up = up;
                    pi = p;
                    pj = j;
                    found = true;
                    break;
                }
            }
            if (!found && j != 0) {
                bool hit = true;
                if (Hij == E[i * W + j - 1] + e) ext_left = 1;
                else if (Hij == H[i * W + j - 1] + g) ext_left = 0;
                else if (Hij == Q[i * W + j - 1] + c) ext_left = 2;
                else if (Hij == H[i * W + j - 1] + q) ext_left = 0;
                else hit = false;
                if (hit) {
                    pi = i;
                    pj = j - 1;
                    found = true;
                }
            }
            if (!found) break;  // (cannot happen: every positive cell has a source)
            aln.emplace_back(i == pi ? -1 : G.rank_to_node[i - 1], j == pj ? -1 : int(j) - 1);
            i = pi;
            j = pj;
            if (ext_left) {  // (i, j) is inside a horizontal gap of that piece: back to the cell that opened it
                const std::vector<int32_t>& X = ext_left == 1 ? E : Q;
                const int32_t ext = ext_left == 1 ? e : c;
                for (;;) {
                    aln.emplace_back(-1, int(j) - 1);
                    --j;
                    if (j == 0 || X[i * W + j] + ext != X[i * W + j + 1]) break;
                }
            } else if (ext_up) {  // ... inside a vertical gap
                const std::vector<int32_t>& X = ext_up == 1 ? F : O;
                const int32_t open = ext_up == 1 ? g : q, ext = ext_up == 1 ? e : c;
                for (;;) {
                    bool stop = true;
                    size_t ni = 0;
                    const Node& nd = G.nodes[size_t(G.rank_to_node[i - 1])];
                    for (size_t p : pred_rows(nd)) {
                        if (X[i * W + j] == H[p * W + j] + open) {
                            stop = true;
                            ni = p;
                            break;
                        }
                        if (X[i * W + j] == X[p * W + j] + ext) {
                            stop = false;
                            ni = p;
                            break;
                        }
                    }
                    aln.emplace_back(G.rank_to_node[i - 1], -1);
                    i = ni;
                    if (stop || i == 0) break;
                }
            }
        }
        std::reverse(aln.begin(), aln.end());
        return aln;
    }
};

struct Store {
    Engine eng;
    std::map<int, std::unique_ptr<Graph>> g[2];
    std::vector<std::pair<int, int>> last_aln;
    int32_t last_score = 0;

    void add_seq(Graph& G, const std::string& s, uint32_t weight)
    {
        last_aln = eng.align(s, G, &last_score);
        G.add_alignment(last_aln, s, std::vector<uint32_t>(s.size(), weight));
    }
};

int cb_create(void* u, int side, int idx, const char* seq, int len)
{
    Store* S = static_cast<Store*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    std::unique_ptr<Graph> G(new Graph);
    S->add_seq(*G, std::string(seq, size_t(len)), 1);  // src/cluster.cpp:200-204
    S->g[side][idx] = std::move(G);
    return 0;
}
int cb_size(void* u, int side, int idx)
{
    Store* S = static_cast<Store*>(u);
    if (side < 0 || side > 1) return -1;
    auto it = S->g[side].find(idx);
    return it == S->g[side].end() ? -1 : it->second->num_sequences;
}
int cb_add(void* u, int side, int idx, const char* seq, int len, unsigned weight)
{
    Store* S = static_cast<Store*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    auto it = S->g[side].find(idx);
    if (it == S->g[side].end()) return -1;
    S->add_seq(*it->second, std::string(seq, size_t(len)), weight);  // src/consensus.cpp:15-23
    return 0;
}
int cb_consensus(void* u, int side, int idx, char* out, int cap)
{
    Store* S = static_cast<Store*>(u);
    if (side < 0 || side > 1) return -1;
    auto it = S->g[side].find(idx);
    if (it == S->g[side].end()) return -1;
    const std::string c = it->second->consensus();  // src/consensus.cpp:91
    if (int(c.size()) > cap) return -1;
    std::memcpy(out, c.data(), c.size());
    return int(c.size());
}
int cb_purge(void* u, int side, int idx, const char* seq, int len, unsigned weight)
{
    Store* S = static_cast<Store*>(u);
    if (side < 0 || side > 1 || len < 0) return -1;
    std::unique_ptr<Graph> G(new Graph);
    S->add_seq(*G, std::string(seq, size_t(len)), weight);  // ConsPurge, src/consensus.cpp:128-137
    S->g[side][idx] = std::move(G);
    return 0;
}

}  // namespace

extern "C" {

// the layout of orc_cons_ops (oracle.h) = the first six members of ioc_consensus_ops (include/isonclust2_hip.h)
struct orp_ops {
    void* user;
    int (*create)(void*, int, int, const char*, int);
    int (*size)(void*, int, int);
    int (*add)(void*, int, int, const char*, int, unsigned);
    int (*consensus)(void*, int, int, char*, int);
    int (*purge)(void*, int, int, const char*, int, unsigned);
};

void* orp_create(int m, int n, int g, int e, int q, int c)
{
    Store* S = new Store;
    S->eng = Engine{m, n, g, e, q, c};
    return S;
}
void orp_destroy(void* s) { delete static_cast<Store*>(s); }
void orp_bind(void* s, orp_ops* ops)
{
    ops->user = s;
    ops->create = cb_create;
    ops->size = cb_size;
    ops->add = cb_add;
    ops->consensus = cb_consensus;
    ops->purge = cb_purge;
}
// nodes (letter, topological order), edges (tail, head, weight; creation order) and, per node, its aligned nodes as
// [aligned_off[v], aligned_off[v + 1]) of aligned[]; pass NULLs to size.  Returns 0, -1 if there is no such graph.
int orp_graph_export(void* s, int side, int idx, int32_t* n_nodes, int32_t* n_edges, int32_t* n_aligned, char* bases, int32_t* rank,
                     int32_t* ef, int32_t* et, int64_t* ew, int32_t* aligned_off, int32_t* aligned)
{
    Store* S = static_cast<Store*>(s);
    if (side < 0 || side > 1) return -1;
    auto it = S->g[side].find(idx);
    if (it == S->g[side].end()) return -1;
    const Graph& G = *it->second;
    size_t na = 0;
    for (const Node& nd : G.nodes) na += nd.aligned.size();
    if (n_nodes) *n_nodes = int32_t(G.nodes.size());
    if (n_edges) *n_edges = int32_t(G.edges.size());
    if (n_aligned) *n_aligned = int32_t(na);
    if (bases)
        for (size_t v = 0; v < G.nodes.size(); ++v) bases[v] = char(G.nodes[v].code);
    if (rank)
        for (size_t r = 0; r < G.rank_to_node.size(); ++r) rank[r] = G.rank_to_node[r];
    if (ef && et && ew)
        for (size_t x = 0; x < G.edges.size(); ++x) {
            ef[x] = G.edges[x].tail;
            et[x] = G.edges[x].head;
            ew[x] = G.edges[x].weight;
        }
    if (aligned_off && aligned) {
        int32_t o = 0;
        for (size_t v = 0; v < G.nodes.size(); ++v) {
            aligned_off[v] = o;
            for (int a : G.nodes[v].aligned) aligned[o++] = a;
        }
        aligned_off[G.nodes.size()] = o;
    }
    return 0;
}
// a copy of graph (side, idx) of store `from` becomes graph (to_side, to_idx) of store `to` (the merge of two clustered batches
// takes the left batch's graphs as side 0 and the right batch's as side 1, src/cluster.cpp:273-278); 0, -1 if there is none
int orp_graph_copy(void* from, int side, int idx, void* to, int to_side, int to_idx)
{
    Store* A = static_cast<Store*>(from);
    Store* B = static_cast<Store*>(to);
    if (side < 0 || side > 1 || to_side < 0 || to_side > 1) return -1;
    auto it = A->g[side].find(idx);
    if (it == A->g[side].end()) return -1;
    B->g[to_side][to_idx] = std::unique_ptr<Graph>(new Graph(*it->second));
    return 0;
}
// the alignment of the last create / add / purge: pairs in read order; returns their number
int orp_last_alignment(void* s, int32_t cap, int32_t* nodes, int32_t* pos, int32_t* score)
{
    Store* S = static_cast<Store*>(s);
    if (score) *score = S->last_score;
    const int n = int(S->last_aln.size());
    if (nodes && pos) {
        if (cap < n) return -1;
        for (int i = 0; i < n; ++i) {
            nodes[i] = S->last_aln[size_t(i)].first;
            pos[i] = S->last_aln[size_t(i)].second;
        }
    }
    return n;
}

}  // extern "C"
