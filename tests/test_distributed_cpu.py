"""CPU, world_size 2 over gloo: the multi-GPU path shards utterances across
ranks with no data-path collective.  Each rank decodes its shard (here with the
host-thread emulation of the kernels -- there is no GPU in this CI) and rank 0
checks that the union equals the oracle's decode of the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers
from text_amd import sharding


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 256, 8192):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_length_balanced_shards():
    rng = np.random.RandomState(0)
    lengths = rng.randint(500, 1500, size=64)
    shards = sharding.length_balanced_shards(lengths, 8)
    assert sorted(i for s in shards for i in s) == list(range(64))
    loads = [int(lengths[s].sum()) for s in shards]
    assert max(loads) - min(loads) <= int(lengths.max())


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    import torch.distributed as dist
    import cases
    from oracle import orclib
    from text_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = dict(cases.BY_NAME["lf_ctc_t20_k4"])
        Ts = [20, 5, 13, 0, 17, 9]
        lo, hi = sharding.shard_bounds(len(Ts), rank, world)
        sess = helpers.FltxSession(helpers.EMU_LIB)
        embs = [synth.emissions("ctc", 40 + i, Ts[i], c["N"]) for i in range(lo, hi)]
        flat = np.concatenate([e.reshape(-1) for e in embs]) if embs else np.zeros(0, np.float32)
        d = sess.decoder(c, dict(tr=None), threads=64)
        d.decode_batch(flat, Ts[lo:hi], c["N"])
        local = {lo + b: [(h.score, h.tokens.tolist()) for h in d.results(b)] for b in range(hi - lo)}
        merged = sharding.gather_results(local, dist)
        if rank == 0:
            orc = orclib.load("oracle")
            ok = True
            for i, T in enumerate(Ts):
                want = helpers.run_checker(orc, dict(c, T=T),
                                           dict(e=synth.emissions("ctc", 40 + i, T, c["N"]), tr=None, lex=None))
                got = merged[i]
                ok &= len(want) == len(got) and all(
                    w.score == g[0] and w.tokens.tolist() == g[1] for w, g in zip(want, got))
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_and_agree_with_oracle(emu_session):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
