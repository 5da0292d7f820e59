"""TEST INFRASTRUCTURE ONLY — CPU restatement of HierarchicalClusterer::DoClustering
(tools/HierarchicalClusterer.cpp:46-140).  Never imported by the product path.

Parity unpinned: the reference holds no test or golden output for this class (it is compiled into
clustermatepairs but never called) and it needs Boost.Bimap, absent here.  The order of equal keys in
`multiset_of<double>` is restated from Boost.MultiIndex's documented ordered_non_unique behaviour (an
insert goes to the upper bound of its key, so `left.begin()` is the earliest entry among equal minima).
"""


def do_clustering(distances, threshold):
    """distances: n×n table (only [i][j], j > i, is read, :63-66).  Returns GetClusters(): list of lists."""
    n = len(distances)
    if n < 1:                                                    # :50-53
        return []
    clusters = []
    sorted_distances = {}          # SortedPair -> (distance, entry stamp); the bimap's two views in one
    cluster_indices = []
    stamp = 0
    for i in range(n):                                           # :59-67
        clusters.append([i])
        cluster_indices.append(i)
        for j in range(i + 1, n):
            sorted_distances[(i, j)] = (float(distances[i][j]), stamp)
            stamp += 1

    def spair(a, b):
        return (a, b) if a < b else (b, a)

    while sorted_distances:
        to_merge, (dmin, _) = min(sorted_distances.items(), key=lambda kv: kv[1])    # left.begin(), :69-71
        if not dmin < threshold:
            break
        first, second = to_merge
        size_first = float(len(clusters[first]))
        size_second = float(len(clusters[second]))
        size_merged = size_first + size_second
        merged = len(clusters)                                   # :78-81
        clusters.append(clusters[first] + clusters[second])
        kept = []
        for c in cluster_indices:                                # :84-117
            if c == first or c == second:
                continue
            d1, _ = sorted_distances.pop(spair(first, c))
            d2, _ = sorted_distances.pop(spair(second, c))
            d = (size_first * d1 + size_second * d2) / size_merged                   # :107
            sorted_distances[spair(merged, c)] = (d, stamp)
            stamp += 1
            kept.append(c)
        cluster_indices = kept + [merged]                        # :120
        del sorted_distances[to_merge]                           # :123
    return [clusters[c] for c in cluster_indices if len(clusters[c]) >= 1]            # :127-139
