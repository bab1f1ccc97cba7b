"""Worker of tests/test_gpu_parity.py::test_sharded_rbd_over_nccl: one rank of a nccl (RCCL) process group
that evaluates its shard with the HIP kernels and checks gather / scatter against the unsharded call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from rbdreference_amd import RBDReference, iiwa_like
        from rbdreference_amd.dist import ShardedRBD, all_gather_rows, scatter_rows, shard_bounds
        rbd = RBDReference(iiwa_like(), build=False)
        sh = ShardedRBD(rbd)
        for B in (1000, 64 * 7 + 5):
            rng = np.random.default_rng(B)                      # the same global batch on every rank
            q, qd, qdd = (torch.tensor(x, device=dev, dtype=torch.float32) for x in
                          (rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))))
            want_c, want = rbd.rnea_grad(q, qd, qdd, return_c=True)   # unsharded, this device
            want_M = rbd.minv(q)
            a, b = shard_bounds(B, world, rank)
            local_out = sh.rnea_grad(q, qd, qdd)
            assert local_out.shape[0] == b - a
            full = sh.rnea_grad(q, qd, qdd, gather=True)
            assert torch.equal(full, want), "gathered dc_du differs from the unsharded call"
            c_full, dc_full = sh.rnea_grad(q, qd, qdd, gather=True, return_c=True)
            assert torch.equal(c_full, want_c) and torch.equal(dc_full, want)
            assert torch.equal(sh.minv(q, gather=True), want_M)
            qs = scatter_rows(q if rank == 0 else None, B, (7,), q.dtype, dev)
            assert torch.equal(qs, q[a:b])
            again = all_gather_rows(rbd.rnea_grad(qs, qd[a:b], qdd[a:b]), B)
            assert torch.equal(again, want)
        torch.cuda.synchronize()
        print(f"rank {rank}/{world}: sharded ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
