"""Shape of a captured hipGraph from its DOT dump (tools/graph_dot_dump.py): nodes, edges, roots / leaves, the widest
topological level, and the WIDTH of the DAG (largest set of mutually independent nodes = the number of chains a scheduler
needs to run everything that may run concurrently; Dilworth: nodes - maximum matching of the transitive closure).
    python tools/graph_dot_width.py a.dot [b.dot ...]"""
import collections
import re
import sys


def load(path):
    txt = open(path, errors="replace").read()
    nodes, edges = {}, []
    for m in re.finditer(r'"([^"]+)"\s*\[(.*?)\];', txt, flags=re.S):
        lab = re.search(r'label\s*=\s*"([^"]*)"', m.group(2), flags=re.S)
        nodes[m.group(1)] = lab.group(1) if lab else ""
    for m in re.finditer(r'"([^"]+)"\s*->\s*"([^"]+)"', txt):
        edges.append((m.group(1), m.group(2)))
    for a, b in edges:
        nodes.setdefault(a, "")
        nodes.setdefault(b, "")
    return nodes, edges


def analyse(path):
    nodes, edges = load(path)
    ids = {n: i for i, n in enumerate(nodes)}
    n = len(ids)
    succ = [[] for _ in range(n)]
    indeg = [0] * n
    for a, b in set(edges):
        succ[ids[a]].append(ids[b])
        indeg[ids[b]] += 1
    order, q = [], collections.deque(i for i in range(n) if indeg[i] == 0)
    roots = len(q)
    level = [0] * n
    deg = indeg[:]
    while q:
        u = q.popleft()
        order.append(u)
        for v in succ[u]:
            level[v] = max(level[v], level[u] + 1)
            deg[v] -= 1
            if deg[v] == 0:
                q.append(v)
    assert len(order) == n, "cycle?"
    lv = collections.Counter(level)
    # transitive closure as bitsets, reverse topological order
    reach = [0] * n
    for u in reversed(order):
        r = 0
        for v in succ[u]:
            r |= reach[v] | (1 << v)
        reach[u] = r
    # maximum bipartite matching (Hopcroft-Karp would be faster; n is a few thousand at most) on u -> v in reach[u]
    match_r = [-1] * n
    adj = [[v for v in range(n) if (reach[u] >> v) & 1] for u in range(n)]
    sys.setrecursionlimit(10000)

    def try_kuhn(u, seen):
        for v in adj[u]:
            if not seen[v]:
                seen[v] = True
                if match_r[v] < 0 or try_kuhn(match_r[v], seen):
                    match_r[v] = u
                    return True
        return False
    matched = 0
    for u in order:
        if try_kuhn(u, [False] * n):
            matched += 1
    width = n - matched
    fan = collections.Counter(len(s) for s in succ)
    sid = collections.Counter(m.group(1) for lab in nodes.values() for m in [re.search(r"StreamId:(\d+)", lab)] if m)
    if sid:
        print("   runtime stream ids (nodes per id):", dict(sorted(sid.items(), key=lambda kv: int(kv[0]))))
    print("%s: %d nodes, %d edges, %d roots, %d leaves, %d levels, widest level %d, DAG width %d; out-degree histogram %s" % (
        path, n, len(set(edges)), roots, sum(1 for s in succ if not s), len(lv), max(lv.values()), width, dict(sorted(fan.items()))))
    return nodes, succ, ids


if __name__ == "__main__":
    for p in sys.argv[1:]:
        analyse(p)
