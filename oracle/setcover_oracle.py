"""CPU restatement of the reference's `setcover` tool.  TEST INFRASTRUCTURE ONLY.

  read_clusters   tools/Parsers.cpp:23-84    (cluster-end-0 lines only)
  set_cover       tools/setcover.cpp:30-110  (greedy maximum coverage)
  write_clusters  tools/Parsers.cpp:86-170

Tie rule: the reference keeps (cluster, size) in a boost::bimap whose right view is
a multiset ordered by size; `right.rbegin()` is the LAST element among those of maximal size, and
both insertion and `replace_data` place an element after all elements of equal key.  So among the
clusters of maximal current size the one that arrived at that size most recently wins, the initial
arrival order being ascending cluster index.  Parity status: UNPINNED — the reference has no test
or golden vector for this tool and Boost is not available here to run it; the rule is restated from
Boost.MultiIndex's documented ordered_non_unique behaviour.
"""
import heapq


def read_clusters(path):
    clusters = []
    with open(path) as f:
        for n, line in enumerate(f, 1):
            line = line.rstrip("\n")
            if not line:
                raise SystemExit("Error: Empty clusters line %d of %s" % (n, path))
            fields = line.split("\t")
            if len(fields) < 3:
                raise SystemExit("Error: Format error for clusters line %d of %s" % (n, path))
            cid, cend, frag = int(fields[0]), int(fields[1]), int(fields[2])
            if cend != 0:
                continue
            if cid < 0:
                raise SystemExit("Error: Invalid cluster ID for line %d of %s" % (n, path))
            while len(clusters) <= cid:
                clusters.append([])
            clusters[cid].append(frag)
    return clusters


def set_cover(clusters):
    """Returns solution[cluster] = list of assigned elements."""
    solution = [[] for _ in clusters]
    max_el = max((e for c in clusters for e in c), default=-1)
    el2cl = [[] for _ in range(max_el + 1)]
    sizes = []
    seq = []
    counter = 0
    heap = []           # max-heap on (size, seq) with lazy deletion
    for ci, c in enumerate(clusters):
        sizes.append(len(c))
        for e in c:
            if e < 0:
                raise SystemExit("Error: negative elements not permitted")
            el2cl[e].append(ci)
        counter += 1
        seq.append(counter)
        heapq.heappush(heap, (-len(c), -counter, ci))
    assigned = [False] * (max_el + 1)
    while heap:
        s, q, ci = heap[0]
        if -s != sizes[ci] or -q != seq[ci]:
            heapq.heappop(heap)
            continue
        if sizes[ci] == 0:
            break
        for e in clusters[ci]:
            if not assigned[e]:
                solution[ci].append(e)
                assigned[e] = True
                for cj in el2cl[e]:
                    sizes[cj] -= 1
                    counter += 1
                    seq[cj] = counter
                    heapq.heappush(heap, (-sizes[cj], -counter, cj))
    return solution


def write_clusters(in_path, solution, min_size):
    keep = [set(s) if len(s) >= min_size else set() for s in solution]
    out = []
    with open(in_path) as f:
        for line in f:
            line = line.rstrip("\n")
            fields = line.split("\t")
            cid, frag = int(fields[0]), int(fields[2])
            if cid < len(keep) and frag in keep[cid]:
                out.append(line + "\n")
    return "".join(out)


def setcover(in_path, min_size):
    return write_clusters(in_path, set_cover(read_clusters(in_path)), min_size)
