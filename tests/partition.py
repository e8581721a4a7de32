"""The contiguous block partition of the condensed schedule, restated for the tests (no torch: the GPU tests import it too)."""


def contiguous_owners(costs, world):
    """dna_adjust::ComputeBlockOwners(condensed): contiguous runs, the largest run's cost as small as possible"""
    B = len(costs)

    def parts(cap):
        n, load = 1, 0.0
        for c in costs:
            if load + c > cap and load > 0.0:
                n, load = n + 1, 0.0
            load += c
        return n
    lo, hi = max(costs), sum(costs)
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        if parts(mid) <= world:
            hi = mid
        else:
            lo = mid
    cap = hi * (1.0 + 1e-12)
    owner, r, load = [], 0, 0.0
    for k, c in enumerate(costs):
        if load + c > cap and load > 0.0 and r + 1 < world:
            r, load = r + 1, 0.0
        if B - k <= world - 1 - r and load > 0.0 and r + 1 < world:
            r, load = r + 1, 0.0
        owner.append(r)
        load += c
    return owner
