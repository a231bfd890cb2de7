"""Utterance sharding across the GPUs of a node.

The hot path has no exchange step: every utterance's beam search touches only
its own emissions and hypothesis set, the trie / LM tables are read-only
replicas.  So N GPUs = N independent shards, one process
per GPU; torch.distributed is only used to agree on timing (bench.py) or to
collect results on one rank (gather_results)."""


def shard_bounds(n_utt, rank, world):
    """Contiguous, balanced [lo, hi) of utterance indices for `rank`."""
    base, rem = divmod(n_utt, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def length_balanced_shards(lengths, world):
    """Round-robin over utterances sorted by length (longest first): the
    per-rank frame totals stay within one utterance of each other when T
    varies.  Returns a list of index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    load = [0] * world
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def gather_results(local, dist=None):
    """All ranks pass {utterance index: result}; rank 0 gets the merged dict."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    merged = {}
    for part in out:
        merged.update(part)
    return merged
