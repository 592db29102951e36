"""Child process of tests/test_gpu_dist.py (never imported by pytest itself: it initialises RCCL).

init_process_group("nccl", world_size=1) on cuda:0, then the real LWSNet through lwsnet_amd.dist.sharded_forward (the
gather runs through RCCL even at world size 1) and through the asynchronous per-step gather bench.py issues; the
gathered stage-4 maps must equal the plain forward bit for bit.  Prints "OK ..." on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                       # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from lwsnet_amd import dist as ldist
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.synth import make_batch
    from lwsnet_amd.weights import default_args, make_state_dict
    rank, local_rank, world = ldist.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    dev = torch.device("cuda", local_rank)
    model = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
    left, right = make_batch(3, 64, 256, 70)
    left, right = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
    plain = [p.clone() for p in model(left, right)]
    preds, gathered = ldist.sharded_forward(model, left, right)
    assert gathered.is_cuda and gathered is not preds[3]
    assert torch.equal(gathered, plain[3]), "gathered stage-4 maps differ from the plain forward"
    assert all(torch.equal(a, b) for a, b in zip(preds, plain))
    # the per-step asynchronous gather of bench.py, 10 steps back to back
    bufs = [torch.empty_like(plain[3])]
    work = None
    for _ in range(10):
        pred = model(left, right)
        work = ldist.gather_async(pred[3], bufs, dst=0)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(bufs[0], plain[3])
    # the staged gather bench.py uses for 1 pair per step: 7 steps, 3 per gather -> two full buffers + a tail of one
    sg = ldist.StagedGather(3, 64, 256, 3, dev)
    maps = []
    for k in range(7):
        lk = left + 0.01 * k
        maps.append(model(lk, right)[3].clone())
        model(lk, right, out=[None, None, None, sg.slot()])
        sg.commit()
    sg.flush()
    torch.cuda.synchronize()
    got, nvalid = sg.gathered(0)
    assert sg.count == 3 and nvalid == 1 and torch.equal(got[:3], maps[6]), "staged gather: tail slot differs"
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t.item()) == 1.5
    dist.destroy_process_group()
    print("OK nccl world_size=1: sharded_forward, async gather and staged gather bitwise equal to the plain forward")


if __name__ == "__main__":
    main()
