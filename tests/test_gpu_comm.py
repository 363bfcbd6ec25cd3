"""The native halo exchange of the library (sdr_amd/csrc/comm.cpp): RCCL point-to-point and the
single-process peer-copy transport, through the C ABI.  On a one-GPU box the ring closes on itself (a rank's right
neighbour is the rank itself); with two or more devices the same tests run a real ring."""
import os
import subprocess

import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "bin", "halo_ring")


def _chain(L):
    return L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=8192)


def _pattern(rank, n):
    i = np.arange(n, dtype=np.int64)
    return ((rank * 53 + i * 7 + (i >> 8)) & 0xFF).astype(np.uint8)


def test_halo_size_is_the_composed_overlap(hip):
    ch = _chain(hip)
    h = ch.halo_samples()
    assert h % 8 == 0 and ch.max_halo() <= h < ch.max_halo() + 8
    assert 3000 < h < 6000          # ~ (127 + 8*(1 + 64 + 127*10/3)) input samples


def test_rccl_ring_of_one_process_per_gpu(hip):
    """sdrhip_comm_init_rank + sdrhip_fm_chain_halo_exchange, one communicator of size 1: send and receive to self."""
    import torch
    L = hip
    ch = _chain(L)
    shard, halo = 1 << 16, ch.halo_samples()
    comm = L.Comm(1, 0, L.comm_unique_id())
    assert comm.transport == "rccl" and comm.size == 1 and comm.rank == 0
    buf = torch.zeros(2 * (shard + halo), dtype=torch.uint8, device="cuda")
    buf[: 2 * shard] = torch.from_numpy(_pattern(0, 2 * shard)).cuda()
    buf[2 * shard:] = 0xEE
    st = torch.cuda.current_stream()
    for _ in range(3):
        comm.chain_halo_exchange(ch, buf.data_ptr(), shard, stream=st.cuda_stream)
    torch.cuda.synchronize()
    got = buf[2 * shard:].cpu().numpy()
    assert np.array_equal(got, _pattern(0, 2 * halo))
    assert np.array_equal(buf[: 2 * shard].cpu().numpy(), _pattern(0, 2 * shard))      # the shard itself is untouched
    comm.close()


@pytest.mark.parametrize("count", [1, 2, 16])
def test_rccl_halos_of_several_super_blocks_in_one_message(hip, count):
    """sdrhip_fm_chain_halo_exchange_batch (round 5): `count` rows, each a shard and its halo region; one gather, one send / recv pair,
    one scatter.  Ring of one: every row's halo region must end up holding THAT row's head; guard bytes behind every row stay."""
    import torch
    L = hip
    ch = _chain(L)
    shard, halo = 1 << 16, ch.halo_samples()
    comm = L.Comm(1, 0, L.comm_unique_id())
    row_bytes = 2 * (shard + halo) + 64                       # 64 guard bytes behind every row
    rows = torch.full((count, row_bytes), 0xEE, dtype=torch.uint8, device="cuda")
    for k in range(count):
        rows[k, : 2 * shard] = torch.from_numpy(_pattern(k + 1, 2 * shard)).cuda()
    staging = torch.empty(max(1, ch.halo_staging_bytes(count)), dtype=torch.uint8, device="cuda")
    assert ch.halo_staging_bytes(count) == 2 * count * 2 * halo
    st = torch.cuda.current_stream()
    for _ in range(2):
        comm.chain_halo_exchange_batch(ch, rows.data_ptr(), shard, row_bytes, count, staging.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    got = rows.cpu().numpy()
    for k in range(count):
        assert np.array_equal(got[k, 2 * shard: 2 * (shard + halo)], _pattern(k + 1, 2 * halo)), f"row {k}: halo"
        assert np.array_equal(got[k, : 2 * shard], _pattern(k + 1, 2 * shard)), f"row {k}: the shard itself changed"
        assert (got[k, 2 * (shard + halo):] == 0xEE).all(), f"row {k}: wrote past the halo region"
    comm.close()


@pytest.mark.parametrize("transport", ["rccl", "peer-copy"])
def test_single_process_ring_over_all_devices(hip, transport):
    """sdrhip_comm_init_local + sdrhip_halo_exchange_all over every visible device (1 on the test box, 8 on a node)."""
    import torch
    L = hip
    ndev = min(L.device_count(), 8)
    t = L.TRANSPORT_RCCL if transport == "rccl" else L.TRANSPORT_PEER_COPY
    comms = L.Comm.local(list(range(ndev)), t)
    assert [c.rank for c in comms] == list(range(ndev)) and all(c.transport == transport for c in comms)
    ch = _chain(L)
    shard, halo = 1 << 16, ch.halo_samples()
    bufs, streams = [], []
    for r in range(ndev):
        with torch.cuda.device(r):
            b = torch.empty(2 * (shard + halo), dtype=torch.uint8, device=f"cuda:{r}")
            b[: 2 * shard] = torch.from_numpy(_pattern(r, 2 * shard)).to(f"cuda:{r}")
            b[2 * shard:] = 0xEE
            bufs.append(b)
            streams.append(torch.cuda.current_stream(r))
    for r in range(ndev):
        torch.cuda.synchronize(r)
    for _ in range(2):
        L.halo_exchange_all(comms, [s.cuda_stream for s in streams], [b.data_ptr() for b in bufs],
                            [b.data_ptr() + 2 * shard for b in bufs], 2 * halo)
    for r in range(ndev):
        torch.cuda.synchronize(r)
    for r in range(ndev):
        got = bufs[r][2 * shard:].cpu().numpy()
        assert np.array_equal(got, _pattern((r + 1) % ndev, 2 * halo)), f"rank {r}"
    for c in comms:
        c.close()


def test_peer_copy_comm_refuses_the_per_rank_call(hip):
    L = hip
    comms = L.Comm.local([0], L.TRANSPORT_PEER_COPY)
    with pytest.raises(L.SdrHipError):
        comms[0].halo_exchange(1, 2, 16)
    comms[0].close()


def test_c_level_ring_one_process_per_rank(hip, tmp_path):
    """examples/halo_ring.c: plain C over the C ABI, one PROCESS per GPU (two ranks when the box has two devices)."""
    if not os.path.exists(EXE):
        from sdr_amd import build as B
        B.build()
    assert os.path.exists(EXE)
    nranks = min(hip.device_count(), 2)
    idf = str(tmp_path / "rccl.id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([EXE, str(nranks), str(r), idf, str(r), str(1 << 18), "3"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(nranks)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, f"rank {r}: {err}"
        assert f"halo_ring rank {r}/{nranks}: OK (rccl" in out


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_peer_copy_ring_of_several_ranks_on_one_device(hip, nranks):
    """The single-process transport does not care that its ranks share a device: a ring of 2 / 3 / 5 ranks, each with its own
    buffer and stream on device 0, exercises the neighbour indexing and the event ordering of sdrhip_halo_exchange_all on a
    one-GPU box -- and then carries the whole sharded FM chain: every rank runs its shard with the halo it received, and the
    concatenated audio equals the single-stream result (the last rank's halo is the ring's wrap-around, so it runs against
    the stream's end instead)."""
    import torch
    L = hip
    comms = L.Comm.local([0] * nranks, L.TRANSPORT_PEER_COPY)
    ch = _chain(L)
    halo = ch.halo_samples()
    shard = 24 * 8192
    total = nranks * shard
    rng = np.random.default_rng(77 + nranks)
    u8 = rng.integers(0, 256, 2 * total, dtype=np.uint8)
    streams = [torch.cuda.Stream() for _ in range(nranks)]
    bufs = []
    for r in range(nranks):
        b = torch.full((2 * (shard + halo),), 0xEE, dtype=torch.uint8, device="cuda")
        b[: 2 * shard] = torch.from_numpy(u8[2 * r * shard: 2 * (r + 1) * shard]).cuda()
        bufs.append(b)
    torch.cuda.synchronize()
    for _ in range(3):                      # repeated exchanges: the pull of one round is ordered before the next round's
        L.halo_exchange_all(comms, [s.cuda_stream for s in streams], [b.data_ptr() for b in bufs],
                            [b.data_ptr() + 2 * shard for b in bufs], 2 * halo)
    torch.cuda.synchronize()
    for r in range(nranks):
        nxt = (r + 1) % nranks
        assert np.array_equal(bufs[r][2 * shard:].cpu().numpy(), u8[2 * nxt * shard: 2 * nxt * shard + 2 * halo]), f"rank {r}"
    # the sharded chain on those buffers
    full_dev = torch.from_numpy(u8).cuda()
    _, q_all, _ = ch.plan(0, total, total)
    ws = torch.empty(ch.workspace_bytes(total + halo), dtype=torch.uint8, device="cuda")
    ref = torch.zeros(q_all, dtype=torch.float32, device="cuda")
    ch.run(full_dev.data_ptr(), 0, total, ref.data_ptr(), 0, q_all, ws.data_ptr(), ws.numel())
    torch.cuda.synchronize()
    pieces = []
    for r in range(nranks):
        s0, s1 = r * shard, (r + 1) * shard
        q0, q1, _ = ch.plan(s0, s1, total)
        n_in = shard + (halo if r + 1 < nranks else 0)      # the last shard ends with the stream
        out = torch.zeros(q1 - q0, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(streams[r]):
            ch.run(bufs[r].data_ptr(), s0, n_in, out.data_ptr(), q0, q1, ws.data_ptr(), ws.numel(), stream=streams[r].cuda_stream)
        streams[r].synchronize()
        pieces.append(out)
    got = torch.cat(pieces)
    assert got.numel() == q_all
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), "sharded chain over the exchanged halos vs the single stream"
    for c in comms:
        c.close()


def test_bench_gpus_2_runs_two_ranks_on_this_box(hip):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (sharing this box's device when it has one), exchanges
    the halo (host transport: RCCL refuses two ranks on one device) and reports n_gpus 2; the audio of every rank is the same
    whether the exchange is overlapped with the halo-free outputs or not."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(BENCH_TRANSPORT="host", BENCH_CHECKSUM="1")
    crcs = []
    for no_overlap in ("0", "1"):
        env["BENCH_NO_OVERLAP"] = no_overlap
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "512",
                              "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        r = json.loads(lines[0])
        assert r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
        assert "host memory" in r["config"]["halo_transport"]
        crcs.append(r["audio_crc32_per_rank"])
    assert crcs[0] == crcs[1] and len(crcs[0]) == 2
