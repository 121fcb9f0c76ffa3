"""Pair sharding for the multi-GPU path: directed pairs are split across ranks so that
(i, j) and (j, i) stay on one rank and constraint counts balance (greedy bin packing); per-frame
regularisers are owned by rank (frame % world); frames are OWNED in contiguous chunks of ceil(F / world).
The exchange steps run inside the library over RCCL (cvd_comm.hip):
  per Jacobian evaluation   all-reduce g and the per-frame costs, reduce-scatter H_ff to the frames' owners,
                            all-gather diag(H) and the owners' f32 block inverses
  per PCG iteration         reduce-scatter q to the owners + all-reduce [Z^T q | p.q] after the product, all-gather z / c /
                            the r^T z shares after the owners' update -- two grouped collectives per iteration
(dense mode: the pixel walk of a Jacobian evaluation is per pair and divides by the ranks; its per-pair records are folded on
the rank that walked the pair, the sums above follow).

Pure index bookkeeping: no optimizer arithmetic here.
"""
import numpy as np


def shard_pairs(pairs, offsets, world):
    """Returns a list (len = world) of index arrays into `pairs`. Deterministic; union = all, disjoint."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    counts = np.diff(np.asarray(offsets))
    groups = {}
    for i, (a, b) in enumerate(pairs):
        groups.setdefault((min(a, b), max(a, b)), []).append(i)
    order = sorted(groups.items(), key=lambda kv: (-int(counts[kv[1]].sum()), kv[0]))
    load = np.zeros(world, dtype=np.int64)
    out = [[] for _ in range(world)]
    for _, idxs in order:
        r = int(np.argmin(load))
        out[r].extend(idxs)
        load[r] += int(counts[idxs].sum())
    return [np.array(sorted(o), dtype=np.int64) for o in out]


def take_pairs(pairs, offsets, loc, is_static, idx):
    """Sub-collection (pairs, offsets, loc, is_static) of the pairs in idx (kept in map order)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    offsets = np.asarray(offsets)
    new_off = [0]
    locs, stat = [], []
    for i in idx:
        locs.append(loc[offsets[i]:offsets[i + 1]])
        stat.append(is_static[offsets[i]:offsets[i + 1]])
        new_off.append(new_off[-1] + int(offsets[i + 1] - offsets[i]))
    loc_o = np.concatenate(locs) if locs else np.zeros((0, loc.shape[1]), loc.dtype)
    st_o = np.concatenate(stat) if stat else np.zeros((0,), np.uint8)
    return pairs[idx], np.asarray(new_off, dtype=np.int64), loc_o, st_o


def take_pair_flows(pairs, flow, mask, idx):
    """Dense mode: the flow / mask images of the pairs in idx (kept in map order).  A pair's pixel walk is independent of every other
    pair's (reference: one residual block per masked pixel of a pair, lib/FlowConstraints.cpp:381-395, lib/PoseOptimizer.cpp:1185-1232),
    so the images shard with the pairs; every pair carries width x height candidate constraints, hence shard_pairs with uniform
    offsets balances them."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    return pairs[idx], np.ascontiguousarray(flow[idx]), np.ascontiguousarray(mask[idx])


def uniform_offsets(num_pairs, per_pair):
    """offsets of a collection whose pairs all hold per_pair constraints (dense mode: width x height pixel slots)."""
    return np.arange(num_pairs + 1, dtype=np.int64) * int(per_pair)
