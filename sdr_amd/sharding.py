"""Sharding of the sample stream across GPUs.

The stream shards naturally by sample block: rank r of R owns input samples
[r*S, (r+1)*S) of every super-block.  Every stage of the FM chain looks only
FORWARD from an output's first sample (FIR windows, the resampler's polyphase
window) except fmDemod, which looks one decimator output back -- and ownership is
defined so that this never crosses a shard start: an audio output belongs to the
shard in which its whole receptive field STARTS (sdrhip_fm_chain_plan).  So the only
data a rank needs from elsewhere is a RIGHT halo: the first `halo` samples of its
right neighbour's shard (the ntaps-1 overlaps of the four stages composed, ~4k
samples = 8 KB of u8 IQ).  One neighbour send/recv per step, no other collective:
the message is latency-bound, never bandwidth-bound, so xGMI link bandwidth is
irrelevant here and a ring/all-reduce would be the wrong tool.

The exchange is written against torch.distributed's P2P API, so the same code runs
over RCCL ("nccl" backend, GPU tensors) and over gloo (CPU tensors, used by the
world_size-2 tests).
"""


class ShardPlan:
    """What rank `rank` of `world` processes when every rank owns `shard_len` samples."""

    def __init__(self, chain, rank, world, shard_len):
        if shard_len % 8 != 0:
            raise ValueError("shard_len must be a multiple of 8 samples (16-byte aligned u8 IQ tiles)")
        self.rank, self.world, self.shard_len = rank, world, shard_len
        self.s0 = rank * shard_len
        self.s1 = self.s0 + shard_len
        # unbounded stream: the last rank's halo is the head of the next super-block
        self.q0, self.q1, self.halo = chain.plan(self.s0, self.s1, -1)
        cap = chain.max_halo()
        self.halo_cap = (cap + 7) // 8 * 8          # same on every rank: fixed-size messages
        self.n_in = shard_len + self.halo_cap        # samples resident per rank (shard + halo)
        self.k2_samples = shard_len + self.halo      # input samples the decimate kernel actually consumes
        # outputs [q0, q_mid) lie entirely inside the rank's own samples: they can be computed while the
        # halo is still in flight; only [q_mid, q1) wait for it
        self.q_mid = max(self.q0, min(self.q1, chain.ready(self.s1)))
        self.left = (rank - 1) % world
        self.right = (rank + 1) % world


def halo_exchange_start(buf_u8, plan, dist):
    """Device-memory variant, asynchronous: returns the requests; `for r in reqs: r.wait()` makes the CURRENT
    stream wait for the received halo (RCCL runs the transfer on its own stream)."""
    nb = 2 * plan.halo_cap
    head = buf_u8[:nb]
    tail = buf_u8[2 * plan.shard_len: 2 * plan.shard_len + nb]
    ops = [dist.P2POp(dist.isend, head, plan.left), dist.P2POp(dist.irecv, tail, plan.right)]
    return dist.batch_isend_irecv(ops)


def halo_exchange(buf_u8, plan, dist, via_host=False):
    """buf_u8: 1-D uint8 tensor of 2*(shard_len + halo_cap) bytes (interleaved IQ).
    Sends this rank's first halo_cap samples to the LEFT neighbour and receives the
    RIGHT neighbour's head into the halo region.  (For the last rank the right
    neighbour is rank 0, standing for the head of the next super-block.)"""
    nb = 2 * plan.halo_cap
    head = buf_u8[:nb]
    tail = buf_u8[2 * plan.shard_len: 2 * plan.shard_len + nb]
    if plan.world == 1:
        return
    if via_host:
        # plumbing check for backends without device-memory P2P (gloo + GPU buffers)
        h_head = head.cpu()
        h_tail = h_head.new_empty(h_head.shape)
        ops = [dist.P2POp(dist.isend, h_head, plan.left), dist.P2POp(dist.irecv, h_tail, plan.right)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        tail.copy_(h_tail)
        return
    ops = [dist.P2POp(dist.isend, head, plan.left), dist.P2POp(dist.irecv, tail, plan.right)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def halo_exchange_batch(rows_u8, plan, dist, via_host=False):
    """The halos of K consecutive super-blocks in ONE message pair (the twin of sdrhip_fm_chain_halo_exchange_batch):
    rows_u8 is a 2-D uint8 tensor [K, 2*(shard_len + halo_cap)], row k = this rank's shard of super-block k followed by its halo
    region.  Sends the K heads to the LEFT neighbour as one message and receives the RIGHT neighbour's K heads into the K halo
    regions.  For the last rank the right neighbour is rank 0, whose row k then stands for the head of super-block k + 1's first
    shard (a stream cut into super-blocks hands that over as carried state; the synthetic benchmark re-reads the same rows)."""
    if plan.world == 1:
        return
    nb = 2 * plan.halo_cap
    heads = rows_u8[:, :nb].contiguous()                       # gather (the library: hipMemcpy2DAsync)
    if via_host or heads.is_cuda and dist.get_backend() == "gloo":
        h = heads.cpu()
        r = h.new_empty(h.shape)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, h, plan.left), dist.P2POp(dist.irecv, r, plan.right)]):
            req.wait()
        rows_u8[:, 2 * plan.shard_len: 2 * plan.shard_len + nb] = r.to(rows_u8.device)
        return
    recv = heads.new_empty(heads.shape)
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, heads, plan.left), dist.P2POp(dist.irecv, recv, plan.right)]):
        req.wait()
    rows_u8[:, 2 * plan.shard_len: 2 * plan.shard_len + nb] = recv      # scatter
