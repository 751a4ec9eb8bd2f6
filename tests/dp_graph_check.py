"""Run by tests/test_gpu_stepgraph.py in a child process (DVQ_FORCE_DP=1, one-rank nccl group): the data-parallel training
step -- RCCL all-reduces between hipGraph segments, one of them launched from inside the backward -- must match the eager
data-parallel step.  `python tests/dp_graph_check.py PORT [eager|graph|both]` prints stage markers (flushed) so that a
crash can be located."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def say(*a):
    print(*a, flush=True)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    port = int(sys.argv[1])
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    assert os.environ.get("DVQ_FORCE_DP") == "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    say("STAGE init ok")
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    say("STAGE plain all_reduce ok", float(t.sum()))
    import test_gpu_stepgraph as T
    dev = torch.device("cuda:0")
    steps = 7
    res = {}
    if mode in ("eager", "both"):
        res["e"] = T._run(dev, False, steps, "full")
        say("STAGE eager dp steps ok", res["e"][2][-1].tolist())
    if mode in ("graph", "both"):
        res["g"] = T._run(dev, True, steps, "full")
        m_g, tr_g, l_g = res["g"]
        sg = tr_g._graph["sg"]
        kinds = [k for k, _ in sg.items]
        say("STAGE graph dp steps ok", l_g[-1].tolist(), "segments", sg.n_segments(), "eager items", kinds.count("eager"))
        assert tr_g.graph_replays == steps - 2, tr_g.graph_replays
        # exchange points cut the step: VQ statistics (2 forwards), decoder-side + encoder-side gradient all-reduce + wait,
        # discriminator gradients + wait
        assert kinds.count("eager") >= 6 and sg.n_segments() == kinds.count("eager") + 1, kinds
    if mode == "both":
        m_e, tr_e, l_e = res["e"]
        # every bucket exactly once per step and optimizer, eager and replayed alike (the recording pass runs the exchange
        # callables once more: steps + 1)
        for bg, be in zip(tr_g.buckets, tr_e.buckets):
            assert be.launched > 0 and bg.launched * steps == be.launched * (steps + 1), (bg.launched, be.launched)
        np.testing.assert_allclose(l_g, l_e, rtol=1.5e-1, atol=1.5e-2)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    say("DP_GRAPH_OK")


if __name__ == "__main__":
    main()
