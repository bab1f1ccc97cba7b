"""Batch sharding over torch.distributed, exercised on CPU with the gloo backend (world_size 2
and 3).  The compute function is a stand-in (the CPU oracle as checker-side stub) -- the product's
per-rank compute is the HIP library; what is tested here is the sharding / scatter / gather logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rbdreference_amd.dist import shard_bounds, shard_sizes


def test_shard_bounds_cover_the_batch():
    for B in (0, 1, 7, 8, 9, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            s = shard_sizes(B, w)
            assert sum(s) == B and max(s) - min(s) <= 1
            prev = 0
            for r in range(w):
                a, b = shard_bounds(B, w, r)
                assert a == prev and b - a == s[r]
                prev = b
            assert prev == B
    assert shard_bounds(1 << 20, 8, 3) == (3 * 131072, 4 * 131072)      # BASELINE configs[3]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, q_all, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rbd_oracle as orc
        from rbdreference_amd import iiwa_like
        from rbdreference_amd.dist import ShardedRBD, all_gather_rows, scatter_rows
        om = orc.model_from_robot(iiwa_like())

        def grad(q, qd, qdd, return_c=False, **kw):
            if return_c:
                c, dc = orc.rnea_grad(om, q.numpy(), qd.numpy(), qdd.numpy(), return_c=True)
                return torch.from_numpy(c), torch.from_numpy(dc)
            return torch.from_numpy(orc.rnea_grad(om, q.numpy(), qd.numpy(), qdd.numpy()))

        def minv(q, **kw):
            return torch.from_numpy(orc.minv(om, q.numpy()))
        sh = ShardedRBD(None, compute_rnea_grad=grad, compute_minv=minv, model_hash="abc")
        q, qd, qdd = q_all
        local = sh.rnea_grad(q, qd, qdd)                      # stays sharded
        a, b = shard_bounds(B, world, rank)
        assert local.shape[0] == b - a
        full = sh.rnea_grad(q, qd, qdd, gather=True)
        Mi = sh.minv(q, gather=True)
        # scatter from rank 0, compute, gather: same thing
        qs = scatter_rows(q if rank == 0 else None, B, (7,), q.dtype, "cpu")
        assert torch.equal(qs, q[a:b])
        again = all_gather_rows(grad(qs, qd[a:b], qdd[a:b]), B)
        assert torch.equal(again, full)
        # return_c=True makes the compute return (c, dc_du): both are gathered
        c_full, dc_full = sh.rnea_grad(q, qd, qdd, gather=True, return_c=True)
        assert torch.equal(dc_full, full) and c_full.shape == (B, 7)
        # a sub-group whose local rank 0 is NOT global rank 0: scatter_rows takes group-local ranks
        # and must translate them for the point-to-point calls
        if world >= 3:
            sub = dist.new_group(ranks=list(range(1, world)))          # every rank calls new_group
            if rank >= 1:
                sw, sr = dist.get_world_size(sub), dist.get_rank(sub)
                sa, sb = shard_bounds(B, sw, sr)
                qs2 = scatter_rows(q if sr == 0 else None, B, (7,), q.dtype, "cpu", src=0, group=sub)
                assert torch.equal(qs2, q[sa:sb])
                g2 = all_gather_rows(grad(qs2, qd[sa:sb], qdd[sa:sb]), B, group=sub)
                assert torch.equal(g2, full)
                sh2 = ShardedRBD(None, group=sub, compute_rnea_grad=grad, compute_minv=minv, model_hash="abc")
                assert torch.equal(sh2.minv(q, gather=True), Mi)
        if rank == 0:
            ret["full"] = full.numpy(); ret["minv"] = Mi.numpy()
        # mismatching robots must be detected
        try:
            ShardedRBD(None, compute_rnea_grad=grad, compute_minv=minv, model_hash=f"h{rank}")
            ok = False
        except RuntimeError:
            ok = True
        assert ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 64), (2, 37), (3, 10)])
def test_sharded_equals_unsharded_gloo(world, B):
    from oracle import rbd_oracle as orc
    from rbdreference_amd import iiwa_like
    rng = np.random.default_rng(B)
    q = torch.from_numpy(rng.uniform(-3, 3, (B, 7))); qd = torch.from_numpy(rng.uniform(-1, 1, (B, 7)))
    qdd = torch.from_numpy(rng.uniform(-1, 1, (B, 7)))
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, (q, qd, qdd), ret), nprocs=world, join=True)
    om = orc.model_from_robot(iiwa_like())
    assert np.array_equal(ret["full"], orc.rnea_grad(om, q.numpy(), qd.numpy(), qdd.numpy()))   # bit-exact
    assert np.array_equal(ret["minv"], orc.minv(om, q.numpy()))
