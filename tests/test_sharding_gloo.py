"""The N > 1 path on CPU: two processes, gloo backend.  Checks that the shard plans
partition the output stream, that the halo exchange (the same function bench.py
runs over RCCL) delivers exactly the right neighbour's head, and that every rank
ends up holding exactly the samples its planned outputs read."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shard_len, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sdr_amd.lib as L
        import signals as S
        from sdr_amd import sharding
        chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
        plan = sharding.ShardPlan(chain, rank, world, shard_len)
        # the global stream, identical on every rank; each rank keeps only its shard
        stream = S.iq_u8(world * shard_len + plan.halo_cap)
        buf = torch.zeros(2 * plan.n_in, dtype=torch.uint8)
        buf[: 2 * shard_len] = torch.from_numpy(stream[2 * plan.s0: 2 * plan.s1].copy())
        sharding.halo_exchange(buf, plan, dist)
        got = buf[2 * shard_len:].numpy()
        right0 = ((rank + 1) % world) * shard_len
        exp = stream[2 * right0: 2 * (right0 + plan.halo_cap)]
        ok_halo = bool(np.array_equal(got, exp))
        # the same for K super-blocks in one message pair (sdrhip_fm_chain_halo_exchange_batch's twin): row k = the shard of super-block
        # k, marked by adding k to every byte, so a halo that lands in the wrong row (or the wrong rank's) is seen
        for K in (2, 5):
            rows = torch.zeros(K, 2 * plan.n_in, dtype=torch.uint8)
            for k in range(K):
                rows[k, : 2 * shard_len] = torch.from_numpy((stream[2 * plan.s0: 2 * plan.s1] + np.uint8(17 * k)).copy())
            sharding.halo_exchange_batch(rows, plan, dist)
            for k in range(K):
                ok_halo = ok_halo and bool(np.array_equal(rows[k, 2 * shard_len:].numpy(), exp + np.uint8(17 * k)))
                ok_halo = ok_halo and bool(np.array_equal(rows[k, : 2 * shard_len].numpy(), stream[2 * plan.s0: 2 * plan.s1] + np.uint8(17 * k)))
        # planned outputs read only [s0, s1 + halo) and halo <= halo_cap
        ok_range = plan.halo <= plan.halo_cap and plan.q1 >= plan.q0
        gathered = [None] * world
        dist.all_gather_object(gathered, (plan.q0, plan.q1, plan.halo))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok_halo, ok_range, gathered))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, False, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_and_plans(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    shard_len = 16 * 8192
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok_halo, ok_range, gathered in results:
        assert ok_halo, f"rank {rank}: halo mismatch ({gathered})"
        assert ok_range
    plans = sorted(results)[0][3]
    assert plans[0][0] == 0
    for (a0, a1, _), (b0, b1, _) in zip(plans[:-1], plans[1:]):
        assert a1 == b0, "owned output ranges must tile the stream"
    assert all(h > 0 for _, _, h in plans)
