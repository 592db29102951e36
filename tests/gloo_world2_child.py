"""Child process of tests/test_gpu_dist.py::test_two_ranks_on_one_gpu_* (never imported by pytest: it initialises a process group).

    python tests/gloo_world2_child.py RANK WORLD PORT

Rank RANK of WORLD processes that SHARE cuda:0: the real HIP `LWSNet` on every rank, `lwsnet_amd.dist.sharded_forward` over the
gloo backend (RCCL refuses two ranks on one device; the gather carries the device maps through host memory), and on rank 0
the gathered stage-4 maps compared BIT FOR BIT with the unsharded `model(left, right)[3]` of the same process -- the assertion
SURVEY.md section 8(e) "Test without 8 GPUs" asks for, with two processes on the HIP path (/root/reference/models/models.py:106-164
is per-sample, so pure batch sharding cannot change a bit).  Cases: B = 4 and ragged B = 5 at 64x256, and 2 x (8 x 256x512) =
BASELINE config 4's per-rank shape.  Then the staged gather bench.py uses, through the same backend.  Prints "OK ..." on rank 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    from lwsnet_amd import dist as ldist          # (exports GPU_MAX_HW_QUEUES / HSA_ENABLE_IPC_MODE_LEGACY before HIP is up)
    import torch
    import torch.distributed as dist
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.synth import make_batch
    from lwsnet_amd.weights import default_args, make_state_dict
    assert os.environ["GPU_MAX_HW_QUEUES"] and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    r, local_rank, w = ldist.init_from_env("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    dev = ldist.local_device(local_rank, share_one_gpu=True)
    assert dev == torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
    for B, H, W in ((4, 64, 256), (5, 64, 256), (8 * world, 256, 512)):
        left, right = make_batch(B, H, W, 300 + B)
        left, right = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
        preds, gathered = ldist.sharded_forward(model, left, right)
        lo, hi = ldist.shard_range(B, rank, world)
        assert preds[3].shape[0] == hi - lo and preds[3].is_cuda
        if rank == 0:
            want = model(left, right)
            assert gathered.is_cuda and gathered.shape == want[3].shape
            assert torch.equal(gathered, want[3]), f"B={B} {H}x{W}: gathered stage-4 maps differ from the unsharded forward"
            assert all(torch.equal(p, q[lo:hi]) for p, q in zip(preds, want))
        else:
            assert gathered is None
        dist.barrier()
    # the staged gather of bench.py (1 pair per rank per step, 3 steps per gather, 7 steps -> two full buffers + a tail of one)
    B, H, W = 1, 64, 256
    left, right = make_batch(world, H, W, 500)
    left, right = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
    sg = ldist.StagedGather(B, H, W, 3, dev)
    sg.warm()
    for k in range(7):
        model(left[rank:rank + 1] + 0.01 * k, right[rank:rank + 1], out=[None, None, None, sg.slot()])
        sg.commit()
    sg.flush()
    torch.cuda.synchronize()
    if rank == 0:
        assert sg.count == 3
        want = model(left + 0.01 * 6, right)[3]
        for q in range(world):
            got, nvalid = sg.gathered(q)
            assert nvalid == 1 and torch.equal(got[:1], want[q:q + 1]), f"staged gather: rank {q}'s tail slot differs"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"OK gloo world_size={world} on one GPU: sharded HIP forward == unsharded, bit for bit (B=4, ragged B=5, "
              f"{world} x 8 x 256x512); staged gather too")


if __name__ == "__main__":
    main()
