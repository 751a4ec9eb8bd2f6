"""can two processes share ONE GPU and exchange device tensors through the gloo backend? (RCCL refuses duplicate devices)"""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def work(rank, world, port, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    x = torch.full((1 << 20,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    y = torch.arange(8, device=dev, dtype=torch.float32) * (rank + 1)
    dist.broadcast(y, 0)
    torch.cuda.synchronize()
    print(backend, "rank", rank, "all_reduce ->", float(x[0]), float(x[-1]), "broadcast ->", y.tolist()[:3], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    for backend, port in (("gloo", 29611), ("nccl", 29612)):
        try:
            mp.spawn(work, args=(2, port, backend), nprocs=2, join=True)
        except Exception as e:
            print(backend, "FAILED:", str(e).splitlines()[-1][:300], flush=True)
