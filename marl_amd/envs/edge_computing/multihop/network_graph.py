"""Host-side, one-time construction of the static tables the batched environment reads.

Reproduces, draw for draw, what the reference builds in
  generate_graph            envs/edge_computing/multihop/network_graph.py:219-264
  Link                      envs/edge_computing/multihop/links.py:7-9
  EdgeNode.__init__         envs/edge_computing/multihop/edge_node.py:22-64
so that a batched env and the reference env constructed from the same namespace
have identical topology, link parameters and node parameters.  The random
sources are the same library generators the reference uses (stdlib `random`
seeded 2023 at network_graph.py:13, `numpy.random.RandomState(link_seed)` and
`RandomState(edge_node_seed)`): their streams ARE the specification.
"""
import random
from dataclasses import dataclass, field

import numpy as np

GIGA = 1024 * 1024 * 1024
TOPOLOGY_PY_SEED = 2023  # network_graph.py:13


@dataclass
class StaticTables:
    edges: list                 # undirected, in insertion order: ring first, then chords
    neighbours: list            # per node, in adjacency-list insertion order (edge_node.py:54)
    cores: np.ndarray           # [N] int32
    p: np.ndarray               # [N] task_probability
    efr: np.ndarray             # [N] execution_failure_rate
    cap: np.ndarray             # [N] cpu_capacity (cycles/s)
    qlen_max: np.ndarray        # [N] execution_queue_len
    rate: np.ndarray            # [N,N] bit/s, 0 where not adjacent
    tfr: np.ndarray             # [N,N]
    adj: np.ndarray             # [N,N+1] uint8: node.actions membership (neighbours + self)
    extra: dict = field(default_factory=dict)

    @property
    def n_nodes(self):
        return len(self.cores)

    def unique_edges(self):
        """Graph.get_all_edges_3 (network_graph.py:170-190): (min,max) pairs in adjacency-dict order."""
        seen, out = set(), []
        order = []  # defaultdict key insertion order == first time a node appears in add_edge_bandwidth
        for u, v in self.edges:
            for x in (v, u):  # adj_list[v] is touched first (network_graph.py:47-50)
                if x not in order:
                    order.append(x)
        for u in order:
            for v in self.neighbours[u]:
                e = (min(u, v), max(u, v))
                if e not in seen:
                    seen.add(e)
                    out.append(e)
        return out


def synthetic_edges(n, rng=None):
    """Ring + a sparse random matching of chords (network_graph.py:235-249)."""
    if n < 2:
        raise AssertionError("Number of nodes must be greater than or equal to 2")
    rng = rng if rng is not None else random.Random(TOPOLOGY_PY_SEED)
    edges = [(i, (i + 1) % n) for i in range(n - 1)] + [(n - 1, 0)]
    present = set(edges)
    pending = set(range(n))
    while len(pending) > 1:
        a = rng.choice(list(pending))
        b = rng.choice(list(pending - {a}))
        if (a, b) not in present and (b, a) not in present:
            edges.append((a, b))
            present.add((a, b))
        pending.discard(a)
        pending.discard(b)
    return edges


def graphml_edges(path):
    """Edge list of a real-world topology file as generate_graph2 takes it (network_graph.py:267-307):
    `[(int(u), int(v)) for u, v in nx.read_graphml(path).edges()]`, without networkx.  Returns (n_nodes, edges).

    networkx adds the file's nodes in document order, then its edges in document order, and `Graph.edges()` walks the
    nodes in that order and each node's neighbours in insertion order, reporting an edge at the first endpoint it
    reaches: the edge ORDER decides which link draws (RandomState(link_seed), two per edge and pass) an edge gets, so
    it is reproduced here.  A file with parallel edges is a MultiGraph to networkx, whose edges() repeats them."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    ns = root.tag[:root.tag.index("}") + 1] if root.tag.startswith("{") else ""
    graph = root.find(ns + "graph")
    if graph is None:
        raise ValueError("%s: no <graph> element" % path)
    if graph.get("edgedefault", "undirected") == "directed":
        raise ValueError("%s: directed topologies are not what generate_graph2 reads (it int()s an undirected edge list)" % path)
    adj = {}  # node id -> {neighbour id: multiplicity}, both in insertion order
    for node in graph.findall(ns + "node"):
        adj.setdefault(node.get("id"), {})
    for e in graph.findall(ns + "edge"):
        u, v = e.get("source"), e.get("target")
        adj.setdefault(u, {})
        adj.setdefault(v, {})
        adj[u][v] = adj[u].get(v, 0) + 1
        if u != v:
            adj[v][u] = adj[v].get(u, 0) + 1
    edges, seen = [], set()
    for u, nbrs in adj.items():
        for v, mult in nbrs.items():
            if v not in seen:
                edges.extend([(int(u), int(v))] * mult)
        seen.add(u)
    return len(adj), edges


def link_parameters(n_edges, args):
    """Two passes over RandomState(link_seed), each constructing a full Link (two uniforms) per edge and
    keeping one of its fields (network_graph.py:252-254, links.py:7-9)."""
    rs = np.random.RandomState(args.link_seed)

    def one_link():
        rate = rs.uniform(args.transmission_rate_min, args.transmission_rate_max) * 1000000 * 8
        fail = rs.uniform(args.transmission_failure_rate_min, args.transmission_failure_rate_max)
        return rate, fail

    rates = [one_link()[0] for _ in range(n_edges)]
    fails = [one_link()[1] for _ in range(n_edges)]
    return rates, fails


def build_static_tables(args, edges=None):
    n = int(args.edge_node_num)
    topo_file = getattr(args, "topology_file", "") or ""
    if edges is None and topo_file:
        # generate_graph2: a GraphML topology (the reference ships none; the path is absolute or relative to the cwd).
        # As the reference notes (:293), edge_node_num must already be the file's node count.
        n_file, edges = graphml_edges(topo_file)
        if n_file != n:
            raise ValueError("topology file %s has %d nodes, edge_node_num is %d" % (topo_file, n_file, n))
        bad = [e for e in edges if not (0 <= e[0] < n and 0 <= e[1] < n)]
        if bad:
            raise ValueError("topology file %s: node ids must be 0..%d, got edge %s" % (topo_file, n - 1, bad[0]))
    edges = list(edges) if edges is not None else synthetic_edges(n)
    rates, fails = link_parameters(len(edges), args)

    # adjacency lists in the reference's insertion order (network_graph.py:37-50)
    nbr = [[] for _ in range(n)]
    rate = np.zeros((n, n), dtype=np.float64)
    tfr = np.zeros((n, n), dtype=np.float64)
    for (u, v), r, f in zip(edges, rates, fails):
        if u not in nbr[v]:
            nbr[v].append(u)
            rate[v, u], tfr[v, u] = r, f
        if v not in nbr[u]:
            nbr[u].append(v)
            rate[u, v], tfr[u, v] = r, f

    # node parameters: ONE shared RandomState, three draws per node in this order (edge_node.py:31,34,57)
    rs = np.random.RandomState(args.edge_node_seed)
    cores = np.zeros(n, dtype=np.int32)
    p = np.zeros(n)
    efr = np.zeros(n)
    for i in range(n):
        cores[i] = rs.choice(args.cpu_core_list)
        p[i] = rs.uniform(args.task_probability_min, args.task_probability_max)
        efr[i] = rs.uniform(args.execution_failure_rate_min, args.execution_failure_rate_max)
    cap = np.array([int(c) * GIGA * args.single_core_cpu_capacity * args.beta for c in cores], dtype=np.float64)
    qlen_max = np.array([int(c) / args.cpu_core_list[0] for c in cores], dtype=np.float64)

    adj = np.zeros((n, n + 1), dtype=np.uint8)
    for i in range(n):
        for j in nbr[i]:
            adj[i, j] = 1
        adj[i, i] = 1
        tfr[i, i] = 0.0  # edge_node.py:60-61
    return StaticTables(edges=edges, neighbours=nbr, cores=cores, p=p, efr=efr, cap=cap, qlen_max=qlen_max,
                        rate=rate, tfr=tfr, adj=adj)


_CACHE = {}


def get_static_tables(args):
    """Per-process cache, like the reference's class-level `EdgeNode.graph` (edge_node.py:19-20,87-91): the
    topology is generated once.  Keyed by what determines it, so that a second configuration in the same
    process gets what a fresh reference process would have built."""
    key = (getattr(args, "topology_file", "") or "", int(args.edge_node_num), int(args.link_seed), int(args.edge_node_seed),
           float(args.transmission_rate_min), float(args.transmission_rate_max),
           float(args.transmission_failure_rate_min), float(args.transmission_failure_rate_max),
           float(args.task_probability_min), float(args.task_probability_max),
           float(args.execution_failure_rate_min), float(args.execution_failure_rate_max),
           tuple(args.cpu_core_list), float(args.single_core_cpu_capacity), float(args.beta))
    if key not in _CACHE:
        _CACHE[key] = build_static_tables(args)
    return _CACHE[key]
