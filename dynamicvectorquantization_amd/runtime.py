"""Process-wide runtime state of the HIP path: compute dtype, packed-weight epoch, kernel selection."""
from __future__ import annotations

import contextlib
import os

import torch

# activations / packed weights are computed in this dtype ("fp32" = parity mode, "bf16" = perf mode)
_compute_dtype = torch.float32
# bumped whenever parameters change in place (optimizer step, load_state_dict): invalidates packed weights
_weights_epoch = 0
# 0 auto, 1 naive kernels, 2 generic MFMA kernels (LDS-DMA igemm), 3 register-staged igemm (A/B),
# 4 require the LDS-resident-halo 3x3 kernel, 9 require the pipelined implicit-GEMM conv kernel (tests cross-check 1 / 2 / 4 / 9 on the GPU)
_impl = 0


def compute_dtype() -> torch.dtype:
    return _compute_dtype


_fp32_split = None          # None: the library's default (environment DVQ_FP32_SPLIT); True / False: set explicitly


def fp32_split() -> bool:
    """fp32 compute with the matrix products on split-bf16 planes (three bf16 MFMA passes, ~2^-17 relative error per product): `fp32x3`"""
    from . import _lib
    return bool(_lib.load().dvq_fp32_split())


def set_fp32_split(on):
    global _fp32_split
    from . import _lib
    _fp32_split = None if on is None else bool(on)
    if on is not None:
        _lib.check(_lib.load().dvq_set_fp32_split(int(bool(on))), "dvq_set_fp32_split")


def set_compute_dtype(dtype):
    """torch.float32 / torch.bfloat16, or "bf16" | "fp32" (exact fp32 matrix instructions) | "fp32x3" (fp32 tensors, products as three
    bf16 MFMA passes on split operands: the fast tolerance-meeting mode).  A torch dtype leaves the fp32 product mode as it is."""
    global _compute_dtype
    if isinstance(dtype, str):
        if dtype in ("fp32x3", "float32x3"):
            set_fp32_split(True)
            dtype = torch.float32
        else:
            if dtype in ("fp32", "float32") and _fp32_split:
                set_fp32_split(False)
            dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[dtype]
    assert dtype in (torch.float32, torch.bfloat16)
    _compute_dtype = dtype


@contextlib.contextmanager
def compute_dtype_ctx(dtype):
    """compute dtype (and fp32 product mode) inside the block, both restored afterwards -- the library's flag to what it WAS (which may
    have come from the environment), the Python-side record likewise"""
    global _fp32_split
    old, old_flag = _compute_dtype, _fp32_split
    old_lib = fp32_split() if isinstance(dtype, str) and dtype.startswith(("fp32", "float32")) else None
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        set_compute_dtype(old)
        if old_lib is not None and fp32_split() != old_lib:
            set_fp32_split(old_lib)
        _fp32_split = old_flag


def weights_epoch() -> int:
    return _weights_epoch


def bump_weights_epoch():
    """every parameter of the process may have changed (load_state_dict, manual edits, storage moves)"""
    global _weights_epoch
    _weights_epoch += 1


# Parameters owned by one optimizer form a group (Parameter._dvq_group, set by HipAdam.flatten): its step only
# invalidates the packed copies of that group, so the frozen VGG16 of LPIPS and the other optimizer's convs are not
# re-packed after every step.
_group_epoch = {}


def bump_group_epoch(group: int):
    _group_epoch[group] = _group_epoch.get(group, 0) + 1


def param_epoch(param):
    g = getattr(param, "_dvq_group", 0)
    return (_weights_epoch, _group_epoch.get(g, 0))


# Codebooks are rewritten in place by the EMA update of every training forward (quantize2_mask.py:117-128).  An eagerly
# launched update bumps its own module's version; a REPLAYED training step rewrites the codebooks without running any
# Python, so the Trainer bumps this process-wide epoch after every replay -- the search-side copies (vq_prepare planes)
# of every VQEmbedding are then rebuilt by the next eager search.
_codebook_epoch = 0


def codebook_epoch() -> int:
    return _codebook_epoch


def bump_codebook_epoch():
    global _codebook_epoch
    _codebook_epoch += 1


# GroupNorm+swish applied inside the consuming conv kernel (no materialised activation: saves HBM traffic and
# ~1/3 of the saved-activation memory) -- measured SLOWER than the separate HBM-bound pass at B=64 (the in-LDS
# transform steals VALU issue slots from MFMA-feeding waves), so it is off by default; the statistics epilogue of
# the producing conv is always on.  DVQ_FUSE_GN=1 enables it (e.g. when memory-bound on activations).
_fuse_gn = os.environ.get("DVQ_FUSE_GN", "0") == "1"


def fuse_gn_prologue() -> bool:
    return _fuse_gn


_fuse_gn_inference = os.environ.get("DVQ_FUSE_GN_INFERENCE", "1") == "1"


def fuse_gn_inference() -> bool:
    """GroupNorm+swish applied inside the consuming 3x3 conv on forwards that record no tape"""
    return _fuse_gn_inference


def set_fuse_gn_prologue(v: bool):
    global _fuse_gn
    _fuse_gn = bool(v)


def impl() -> int:
    return _impl


def set_impl(v: int):
    global _impl
    assert v in (0, 1, 2, 3, 4, 9)
    _impl = v


@contextlib.contextmanager
def impl_ctx(v: int):
    old = _impl
    set_impl(v)
    try:
        yield
    finally:
        set_impl(old)


# ---------------------------------------------------------------------------------------------------------------
# Training-step capture (hipGraph).  The step's launch sequence is static (fixed shapes, no host-side data dependence), so
# the Trainer records it once and replays it: ~2400 ctypes launches per step become a handful of graph launches.  The
# captured sequence is cut at every point where something must run eagerly (RCCL collectives): `graph_break(fn)` ends the
# current capture segment, runs `fn` now and remembers it, then starts the next segment -- on replay the segments and
# the remembered callables run in recorded order.  Outside a capture `graph_break(fn)` is just `fn()`.
# ---------------------------------------------------------------------------------------------------------------
_capture = None
_GRAPH_DEBUG = os.environ.get("DVQ_GRAPH_DEBUG", "0") == "1"
_BREAK_EVERY = int(os.environ.get("DVQ_GRAPH_BREAK_EVERY", "0"))
_SEPARATE_POOLS = os.environ.get("DVQ_GRAPH_SEPARATE_POOLS", "0") == "1"     # experiment: one private pool per segment


def capturing() -> bool:
    return _capture is not None


def graph_break(fn):
    join_side()                # a segment may not end (and an exchange may not start) with weight gradients still on the side stream
    if _capture is None:
        return fn()
    return _capture.brk(fn)


# host-side dropout seeds (a hash of (seed, element index) decides every keep bit on the device: csrc/dvq_common.h): derived from torch's
# seed + a process-wide counter, so runs are reproducible under torch.manual_seed and no launch reads device RNG state
_seed_counter = [0]


def next_dropout_seed() -> int:
    _seed_counter[0] += 1
    return (torch.initial_seed() * 1000003 + _seed_counter[0]) & 0x7FFFFFFFFFFFFFFF


def step_replay_mode() -> str:
    """how a recorded segment is replayed: "list" = csrc/cmdlist.hip re-issues the captured launches on the current + side stream
    (default: the device sees an eager step's queues); "graph" = hipGraphLaunch of the instantiated capture (DVQ_STEP_REPLAY)"""
    return os.environ.get("DVQ_STEP_REPLAY", "list")


class StepGraph:
    """A recorded training step: hipGraph segments interleaved with eager callables, sharing one private memory pool
    (segments are replayed in capture order, never concurrently, so blocks freed in one segment may be reused by the next)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.items = []                       # ("graph", torch.cuda.CUDAGraph) | ("eager", callable)
        self.stream = torch.cuda.Stream(self.device)
        self.aux = torch.cuda.Stream(self.device)       # eager items of the recording pass (see _brk)
        self.pool = torch.cuda.graph_pool_handle()
        self._g = None
        self._tick = torch.zeros(1, dtype=torch.int32, device=self.device)   # first node of every segment: no empty graphs
        # RCCL's watchdog thread polls events while we capture: only this thread's unsafe calls may invalidate the capture
        self._mode = "thread_local"

    def _dbg(self, *a):
        if _GRAPH_DEBUG:
            print("[stepgraph]", *a, flush=True)

    def _on_launch(self, what):
        """debugging: library calls of the current segment; DVQ_GRAPH_BREAK_EVERY=N cuts a segment every N calls so that a
        faulting replay can be narrowed down to a handful of launches"""
        self._names.append(what)
        if _BREAK_EVERY and len(self._names) >= _BREAK_EVERY and not self._in_brk:
            self.brk(lambda: None)

    def _begin(self):
        self._dbg("begin segment", len(self.items))
        self._names = []
        # launch-list replay (default) reads the captured nodes itself and never instantiates an executable graph
        g = torch.cuda.CUDAGraph(keep_graph=True) if step_replay_mode() == "list" else torch.cuda.CUDAGraph()
        if _SEPARATE_POOLS:
            self.pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self.pool, capture_error_mode=self._mode)
        self._g = g
        self._tick.add_(1)

    def _end(self):
        self._g.capture_end()
        if _GRAPH_DEBUG:
            self._dbg("segment", len(self.items), "calls:", " ".join(getattr(self, "_names", [])))
        if step_replay_mode() == "list":
            from . import kernels as K
            from ._lib import DvqError
            try:
                self.items.append(("list", K.CmdList(self._g)))
            except DvqError as e:
                # a segment with nodes the list cannot re-issue (a torch memcpy): this segment replays through hipGraphLaunch
                import warnings
                warnings.warn(f"StepGraph: segment {len(self.items)} replays as a hipGraph ({e})")
                self.items.append(("graph", self._g))
        else:
            self.items.append(("graph", self._g))
        self._g = None

    def brk(self, fn):
        self._in_brk = True
        try:
            return self._brk(fn)
        finally:
            self._in_brk = False

    def _brk(self, fn):
        self._end()
        self._dbg("eager item", len(self.items), getattr(fn, "__qualname__", fn))
        # While recording, the eager callable runs on a stream that is NEVER captured: RCCL keeps events of its work (its
        # watchdog thread polls them), and HIP refuses to query an event whose last record was on a stream that is capturing
        # at query time (hipErrorCapturedEvent) -- which the recording stream is again a moment later.
        self.aux.wait_stream(self.stream)
        with torch.cuda.stream(self.aux):
            out = fn()
        self.stream.wait_stream(self.aux)
        if _GRAPH_DEBUG:
            torch.cuda.synchronize(self.device)
            self._dbg("eager item done")
        self.items.append(("eager", fn))
        self._begin()
        return out

    @contextlib.contextmanager
    def capture(self):
        """everything launched inside runs in capture mode on a side stream (kernels are recorded, not executed)"""
        global _capture
        assert _capture is None, "nested step capture"
        import gc
        torch.cuda.synchronize(self.device)
        gc.collect()
        torch.cuda.empty_cache()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        from . import _lib
        if _GRAPH_DEBUG:
            _lib._launch_hook = self._on_launch
        self._in_brk, self._names = False, []
        with torch.cuda.stream(self.stream):
            _capture = self
            ok = False
            try:
                self._begin()
                yield self
                ok = True
            finally:
                _capture = None
                _lib._launch_hook = None
                if self._g is not None:
                    try:
                        if ok:
                            join_side(self.device)
                        self._end()
                    except Exception:
                        if ok:
                            raise
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        torch.cuda.synchronize(self.device)

    def replay(self):
        for i, (kind, it) in enumerate(self.items):
            if kind == "graph":
                it.replay()
            elif kind == "list":
                it.replay(torch.cuda.current_stream(self.device), side_stream(self.device)["stream"])
            else:
                it()
            if _GRAPH_DEBUG:                     # localise a faulting segment: finish every item before the next one
                torch.cuda.synchronize(self.device)
                self._dbg("replayed item", i, kind)

    def n_segments(self):
        return sum(1 for k, _ in self.items if k != "eager")

    def launch_counts(self):
        """(kernel launches, of those on the side stream, cross-stream waits) of one replay; None for hipGraph segments"""
        ls = [it for k, it in self.items if k == "list"]
        return (sum(l.kernels for l in ls), sum(l.side_kernels for l in ls), sum(l.waits for l in ls)) if ls else None

    def __del__(self):
        # the recorded kernels hold the address of this stream's scratch slot (csrc/misc.hip: dvq_workspace_stream pins it while
        # capturing): with the recording gone the slot may serve another stream
        try:
            from . import kernels as K
            K.workspace_release(self.stream)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# Weight gradients on a side stream: wgrad(dy, x) of a convolution has no consumer before the optimizer step (or the gradient
# exchange of its bucket), while dgrad(dy) heads the chain dx -> GroupNorm backward -> ... of HBM- / VALU-bound kernels.  Issued on
# a second stream the MFMA-bound weight-gradient kernels fill the matrix pipes while that chain runs.  All weight gradients share
# ONE side stream, so the registered workspace (split partials) is used by one kernel at a time.
# ---------------------------------------------------------------------------------------------
_side = {}


_side_on = False


def side_wgrad_enabled():
    return _side_on


class side_wgrad:
    """context: inside, layers.Conv2d.bwd issues its weight gradient on the side stream; on exit the current stream has waited
    for it.  Only callers that read gradients through join_side() points may switch this on (trainer.Trainer does; code that
    calls bwd() and then reads .grad directly keeps the single-stream behaviour).  DVQ_SIDE_WGRAD=0 disables it."""

    def __enter__(self):
        global _side_on
        self.prev = _side_on
        # eagerly launched steps only: a hipGraph replay of the two-stream capture ran no faster than the single-stream one on
        # ROCm 7.2 (179.9 vs 179.1 ms, with 39 ms of host time per launch; DEBUG_HIP_FORCE_GRAPH_QUEUES / packet capture made no
        # difference), while eager steps gain 2.5 % (174.8 vs 179.3 ms); DVQ_SIDE_WGRAD=graph forces it inside captures too.
        # A capture that is replayed as a launch list keeps the fork: the list issues the side chain on the side stream.
        mode = os.environ.get("DVQ_SIDE_WGRAD", "1")
        _side_on = mode == "graph" or (mode == "1" and (not capturing() or step_replay_mode() == "list"))
        return self

    def __exit__(self, *exc):
        global _side_on
        _side_on = self.prev
        join_side()
        return False


def side_stream(device=None):
    dev = torch.device(device if device is not None else torch.cuda.current_device())
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _side:
        _side[key] = {"stream": torch.cuda.Stream(torch.device("cuda", key)), "dirty": False, "keep": []}
    return _side[key]


def run_on_side(fn, *tensors):
    """launch fn() on the side stream after everything issued so far on the current stream; `tensors` are what it reads"""
    st = side_stream(tensors[0].device if tensors else None)
    cur = torch.cuda.current_stream()
    if cur == st["stream"]:        # already inside work that was forked to the side stream (a head's backward issuing its weight gradients)
        fn()
        return
    st["stream"].wait_stream(cur)
    with torch.cuda.stream(st["stream"]):
        fn()
    # the operands stay referenced until the side stream is done with them: either the current stream has waited for the side stream
    # (join_side) -- freed after that point they are safe to re-use in stream order -- or, on eagerly launched steps, an event recorded
    # behind the launch has completed (then every stream may re-use them).  Without the second rule a step whose only join is the one
    # before the optimizer (the stage-2 step without data parallelism) held every block's gradient temporaries -- ~0.4 GB per
    # StackGPT block -- for the whole backward.  (Tensor.record_stream would defer the re-use of these -- large -- blocks
    # to the allocator's own event queries and make it grow the pool with synchronising hipMallocs for several steps.)
    keep = st["keep"]
    ev = None
    if not capturing():
        free = st.setdefault("events", [])
        while keep and keep[0][0] is not None and keep[0][0].query():
            free.append(keep.pop(0)[0])
        ev = free.pop() if free else torch.cuda.Event()
        ev.record(st["stream"])
    keep.append((ev, tuple(t for t in tensors if t is not None)))
    st["dirty"] = True


def join_side(device=None):
    """make the current stream wait for the side stream (before gradients are read: optimizer step, gradient exchange, the end
    of a recorded segment)"""
    for key, st in _side.items():
        if st["dirty"] and (device is None or torch.device(device).index in (None, key)):
            if torch.cuda.current_stream(torch.device("cuda", key)) == st["stream"]:
                continue           # called from work that itself runs on the side stream: nothing to wait for, the fork stays open
            torch.cuda.current_stream(torch.device("cuda", key)).wait_stream(st["stream"])
            st["dirty"] = False
            st.setdefault("events", []).extend(e for e, _ in st["keep"] if e is not None)
            del st["events"][64:]
            st["keep"].clear()
