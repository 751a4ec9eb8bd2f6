"""Launch lists (csrc/cmdlist.hip, kernels.CmdList): a captured hipGraph re-issued as plain launches on two streams.
The training-step use is covered by tests/test_gpu_stepgraph.py (replays are launch lists by default); here the mechanism
itself: fork / join edges of a two-stream capture, memset nodes, argument blocks borrowed from the graph, refusal of
1-D memcpy nodes.  `pytest -m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture(fn, stream):
    g = torch.cuda.CUDAGraph(keep_graph=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        fn()
    torch.cuda.synchronize()
    return g


def test_cmdlist_replays_two_stream_capture(dev):
    from dynamicvectorquantization_amd import kernels as K
    main, side, cap = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    fork = torch.cuda.Stream(dev)
    x = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    y, z, acc = torch.zeros_like(x), torch.zeros_like(x), torch.zeros(1, device=dev)
    out = torch.zeros(1, device=dev)

    def body():
        cur = torch.cuda.current_stream()
        torch.mul(x, 2.0, out=y)                     # main chain
        fork.wait_stream(cur)
        with torch.cuda.stream(fork):                # side chain: depends on y's producer only through the fork point
            torch.add(x, 1.0, out=z)
            z.mul_(3.0)
        acc.zero_()                                  # a memset or fill node on the main chain
        acc.add_(y.sum())
        cur.wait_stream(fork)                        # join
        torch.add(acc, z.sum(), out=out)

    g = _capture(body, cap)
    cl = K.CmdList(g)
    assert cl.kernels >= 5 and cl.side_kernels >= 2 and cl.waits >= 1, (cl.kernels, cl.side_kernels, cl.waits)
    for rep in range(3):
        x.copy_(torch.arange(1 << 16, dtype=torch.float32, device=dev) * (rep + 1) * 1e-3)
        y.fill_(-1)
        z.fill_(-1)
        out.fill_(-1)
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            cl.replay(main, side)
        main.synchronize()
        ref = (x * 2.0).sum() + ((x + 1.0) * 3.0).sum()
        torch.testing.assert_close(out[0], ref, rtol=1e-5, atol=1e-2)
        assert torch.equal(y, x * 2.0) and torch.equal(z, (x + 1.0) * 3.0)
    del cl                                           # the list goes before the graph whose argument blocks it borrows
    del g


def test_cmdlist_single_stream_needs_no_side_stream(dev):
    from dynamicvectorquantization_amd import kernels as K
    cap, main = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a = torch.ones(1024, device=dev)
    b = torch.zeros(1024, device=dev)
    g = _capture(lambda: torch.add(a, 41.0, out=b), cap)
    cl = K.CmdList(g)
    assert cl.side_kernels == 0 and cl.waits == 0 and not cl.side_open
    b.zero_()
    torch.cuda.synchronize()
    cl.replay(main, main)                            # same stream twice is accepted when nothing forks
    main.synchronize()
    assert float(b.min()) == 42.0 == float(b.max())


def test_cmdlist_refuses_memcpy_nodes(dev):
    """hipMemcpyAsync under capture becomes a 1-D memcpy node whose parameters cannot be read back (ROCm 7.2): the list must say so
    instead of replaying garbage (runtime.StepGraph then falls back to hipGraphLaunch for that segment)"""
    from dynamicvectorquantization_amd import kernels as K
    from dynamicvectorquantization_amd._lib import DvqError
    cap = torch.cuda.Stream(dev)
    a = torch.ones(4096, device=dev)
    b = torch.zeros(4096, device=dev)
    g = _capture(lambda: b.copy_(a), cap)            # contiguous, same dtype: a memcpy, not a kernel
    with pytest.raises(DvqError, match="memcpy"):
        K.CmdList(g)
    c = torch.zeros(4096, device=dev)
    g2 = _capture(lambda: K.copy_kernel_(c, a), cap)  # the kernel copy the step uses instead
    cl = K.CmdList(g2)
    cl.replay(torch.cuda.current_stream(dev), torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    assert torch.equal(c, a)
