"""Randomised segmentations of a GNSS network, cut the way dnasegment cuts (test infrastructure).

dnasegment (dnasegment.cpp:235-348, 528-700) grows a block station by station: a station that becomes INNER takes all its remaining
measurements into the block, the stations at their other ends become JUNCTION stations of the block, and the junction stations left when
the block is full are carried to the next block, where they "become inner or remain junction" (dnasegment.cpp:529-531).  What the
adjustment relies on (dnaadjust.cpp:998-1128, 10449-10499; seg_file.cpp:305-392):
  * a station is inner in exactly one block and junction in the (consecutive) blocks before it, from the first block that holds a
    measurement to it:  JSL(k) is a subset of ISL(k+1) + JSL(k+1);
  * a measurement belongs to exactly one block, the first one in which one of its stations is inner; all its stations are in that
    block's ISL + JSL;
  * the lists of a block are ascending in the global indices.
The synthetic generator's strips (host/synth.cpp) only ever produce JSL(k) inside ISL(k+1).  Here the stations are put in a noisy sweep
order and cut into blocks of random size, so that junction stations persist over several blocks, the junction sets are uneven, and
blocks without a measurement of their own occur (every measurement of their inner stations went to an earlier block)."""
import numpy as np

from tests import dnaformats as F


def baselines(bms):
    """(first record index, station1, station2) of every G baseline of a .bms image"""
    first = np.nonzero((bms["measStart"] == 0) & (bms["measType"] == b"G"))[0]
    if first.size != np.count_nonzero(bms["measStart"] == 0):
        raise ValueError("segfuzz cuts single-baseline (G) networks only")
    return first.astype(np.int64), bms["station1"][first].astype(np.int64), bms["station2"][first].astype(np.int64)


def random_cut(n_stations, first, s1, s2, rng, mean_block=12, noise=0.35, lone_last=True):
    """ISL, JSL, CML of a random dnasegment-like cut.  Stations are ordered by their index (the generator numbers a grid row by row)
    plus noise * mean_block * N(0, 1) and cut into blocks of 1 ... 2 * mean_block inner stations; lone_last: the last station of the
    order is a block of its own (it has no measurement of its own: all of them went to earlier blocks)."""
    order = np.argsort(np.arange(n_stations) + noise * mean_block * rng.standard_normal(n_stations), kind="stable")
    cuts = [0]
    while cuts[-1] < n_stations:
        cuts.append(min(n_stations, cuts[-1] + int(rng.integers(1, 2 * mean_block + 1))))
    if lone_last and cuts[-1] - cuts[-2] > 1:
        cuts.insert(-1, n_stations - 1)
    B = len(cuts) - 1
    blk = np.empty(n_stations, dtype=np.int64)
    for k in range(B):
        blk[order[cuts[k]:cuts[k + 1]]] = k
    ISL = [np.sort(np.nonzero(blk == k)[0]).astype(np.uint32) for k in range(B)]
    mb = np.minimum(blk[s1], blk[s2])                 # the block of a measurement: where its first station turns inner
    CML = [np.sort(first[mb == k]).astype(np.uint32) for k in range(B)]
    # a station is junction from the first block that holds a measurement to it up to the block before its own
    start = blk.copy()
    for a, b in ((s1, s2), (s2, s1)):
        np.minimum.at(start, a, blk[b])
    JSL = [np.sort(np.nonzero((start <= k) & (blk > k))[0]).astype(np.uint32) for k in range(B)]
    return ISL, JSL, CML


def connected_runs(JSL):
    """net id per block: a block whose junction list is empty ends a contiguous network (dnaadjust.cpp:10449-10474)"""
    nets, net = [], 0
    for k in range(len(JSL)):
        nets.append(net)
        if len(JSL[k]) == 0:
            net += 1
    return nets


def write_cut(base, rng, mean_block=12, noise=0.35, lone_last=True):
    """re-segments <base>.{bst,bms} (written as ONE block by the synthetic generator) in place; returns a summary"""
    bst, bms = F.read_bst(base + ".bst"), F.read_bms(base + ".bms")
    first, s1, s2 = baselines(bms)
    ISL, JSL, CML = random_cut(len(bst), first, s1, s2, rng, mean_block, noise, lone_last)
    nets = connected_runs(JSL)
    F.write_seg(base + ".seg", ISL, JSL, CML, nets, bms)
    persist = 0          # longest run of blocks over which one station stays junction
    if len(JSL) > 1:
        life = np.zeros(len(bst), dtype=np.int64)
        for j in JSL:
            life[j] += 1
        persist = int(life.max())
    return {"blocks": len(ISL), "nets": nets[-1] + 1, "max_junction_life": persist, "junction_sizes": [len(j) for j in JSL],
            "blocks_without_measurements": int(sum(1 for c in CML if len(c) == 0)), "inner_sizes": [len(i) for i in ISL]}
