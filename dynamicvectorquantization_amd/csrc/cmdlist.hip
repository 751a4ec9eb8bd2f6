// Launch lists: a recorded training-step segment re-issued as ordinary stream launches.
//
// hipGraphLaunch on ROCm 7.2 runs a 2400-kernel step 7 % slower than the same kernels issued eagerly (extra barrier packets between
// the nodes, and a two-stream capture gains nothing: DESIGN 3a), but the eager step needs the Python host code for every launch.
// A launch list keeps what stream capture is good at -- it has already recorded every (kernel, grid, block, arguments) tuple of
// the step, with the argument blocks deep-copied into the graph's nodes -- and drops the graph executor: the nodes are walked once
// into a flat array of operations, and a replay is a C loop of hipLaunchKernel calls on TWO ordinary streams, main and side
// (the capture's fork / join edges become event record / wait pairs), i.e. exactly the queue contents of an eager step at
// ~2 us of host time per launch.
//
// The list borrows the argument blocks of the hipGraph_t it was built from: the graph must outlive the list (runtime.StepGraph
// keeps the torch.cuda.CUDAGraph(keep_graph=True) object next to the list handle).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dvq_common.h"

namespace {

enum OpKind { OP_KERNEL = 0, OP_MODULE_KERNEL = 1, OP_MEMSET = 2, OP_MEMCPY = 3, OP_RECORD = 4, OP_WAIT = 5 };

struct Op {
    int kind;
    int stream;                 // 0 main, 1 side
    hipKernelNodeParams k;      // OP_KERNEL / OP_MODULE_KERNEL
    hipFunction_t fn;           // OP_KERNEL: the stub's device function, resolved once (hipLaunchKernel looks it up on every call)
    hipMemsetParams ms;         // OP_MEMSET
    hipMemcpy3DParms cp;        // OP_MEMCPY
    int event;                  // OP_RECORD / OP_WAIT
};

struct CmdList {
    std::vector<Op> ops;
    std::vector<hipEvent_t> events;
    hipEvent_t tail_event = nullptr;
    int n_kernels = 0, n_side = 0, n_sync = 0, n_other = 0;
    bool side_open = false;     // the side stream holds work the main stream has not waited for when the list ends
};

#define CL_HIP(call, what)                                                                \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            dvq_set_error("dvq_cmdlist: %s: %s", (what), hipGetErrorString(e__));         \
            return DVQ_ELAUNCH;                                                           \
        }                                                                                 \
    } while (0)

}  // namespace

extern "C" {

int dvq_cmdlist_create(void* graph, dvq_cmdlist_t* out) {
    DVQ_REQUIRE(graph != nullptr && out != nullptr, DVQ_EINVAL, "dvq_cmdlist_create: null pointer");
    hipGraph_t g = (hipGraph_t)graph;
    size_t n = 0, ne = 0;
    CL_HIP(hipGraphGetNodes(g, nullptr, &n), "hipGraphGetNodes");
    std::vector<hipGraphNode_t> nodes(n);
    if (n) CL_HIP(hipGraphGetNodes(g, nodes.data(), &n), "hipGraphGetNodes");
    CL_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne), "hipGraphGetEdges");
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if (ne) CL_HIP(hipGraphGetEdges(g, from.data(), to.data(), &ne), "hipGraphGetEdges");
    // node handle -> creation index
    std::vector<std::pair<hipGraphNode_t, int>> index(n);
    for (size_t i = 0; i < n; ++i) index[i] = {nodes[i], (int)i};
    std::sort(index.begin(), index.end());
    auto idx_of = [&](hipGraphNode_t h) {
        auto it = std::lower_bound(index.begin(), index.end(), std::make_pair(h, -1));
        return it != index.end() && it->first == h ? it->second : -1;
    };
    std::vector<std::vector<int>> deps(n), succ(n);
    std::vector<int> indeg(n, 0);
    for (size_t e = 0; e < ne; ++e) {
        const int a = idx_of(from[e]), b = idx_of(to[e]);
        DVQ_REQUIRE(a >= 0 && b >= 0, DVQ_EINVAL, "dvq_cmdlist_create: edge to a node outside the graph");
        deps[b].push_back(a);
        succ[a].push_back(b);
        ++indeg[b];
    }
    // issue order: creation order where it is a topological order (stream capture: it is), else Kahn with the smallest index first
    std::vector<int> order;
    order.reserve(n);
    {
        std::vector<int> ready;
        for (size_t i = 0; i < n; ++i)
            if (indeg[i] == 0) ready.push_back((int)i);
        std::make_heap(ready.begin(), ready.end(), std::greater<int>());
        while (!ready.empty()) {
            std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
            const int v = ready.back();
            ready.pop_back();
            order.push_back(v);
            for (int s : succ[v])
                if (--indeg[s] == 0) {
                    ready.push_back(s);
                    std::push_heap(ready.begin(), ready.end(), std::greater<int>());
                }
        }
        DVQ_REQUIRE(order.size() == n, DVQ_EINVAL, "dvq_cmdlist_create: the graph has a cycle");
    }
    // two chains: a node continues the stream whose last node it depends on (main preferred); a node that depends on neither tail
    // starts / continues the other chain.  Extra serialisation is always safe, a missing wait never happens: every dependency on a
    // node of the other stream becomes an event pair unless an equal-or-later event of that stream was already waited for.
    std::vector<int> strm(n, 0), pos(n, 0);
    int tail[2] = {-1, -1}, count[2] = {0, 0};
    for (int v : order) {
        int s = 0;
        bool on0 = false, on1 = false;
        for (int d : deps[v]) {
            on0 |= d == tail[0];
            on1 |= d == tail[1];
        }
        if (deps[v].empty() || on0) s = 0;
        else if (on1) s = 1;
        else s = tail[0] < 0 ? 0 : 1;
        strm[v] = s;
        pos[v] = count[s]++;
        tail[s] = v;
    }
    std::vector<int> event_of(n, -1);
    CmdList* L = new CmdList();
    int waited[2] = {-1, -1};          // waited[s]: position on the OTHER stream that stream s has synchronised with
    std::vector<std::vector<int>> waits(n);
    for (int v : order) {
        const int s = strm[v];
        int need = -1, need_node = -1;
        for (int d : deps[v])
            if (strm[d] != s && pos[d] > waited[s] && pos[d] > need) need = pos[d], need_node = d;
        if (need_node >= 0) {
            if (event_of[need_node] < 0) {
                event_of[need_node] = (int)L->events.size();
                L->events.push_back(nullptr);
            }
            waits[v].push_back(event_of[need_node]);
            waited[s] = need;
        }
    }
    for (auto& e : L->events)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            dvq_set_error("dvq_cmdlist_create: hipEventCreate failed");
            delete L;
            return DVQ_ELAUNCH;
        }
    auto fail = [&](const char* what, hipError_t e) {
        dvq_set_error("dvq_cmdlist_create: %s: %s", what, hipGetErrorString(e));
        for (auto ev : L->events)
            if (ev) (void)hipEventDestroy(ev);
        delete L;
        return DVQ_ELAUNCH;
    };
    int last_side = -1, last_join = -1;       // op indices: last side-stream op, last main-stream wait for the side stream
    for (int v : order) {
        const int s = strm[v];
        for (int ev : waits[v]) {
            Op w{};
            w.kind = OP_WAIT, w.stream = s, w.event = ev;
            L->ops.push_back(w);
            ++L->n_sync;
            if (s == 0) last_join = (int)L->ops.size() - 1;
        }
        hipGraphNodeType ty;
        hipError_t e = hipGraphNodeGetType(nodes[v], &ty);
        if (e != hipSuccess) return fail("hipGraphNodeGetType", e);
        Op op{};
        op.stream = s;
        bool emit = true;
        if (ty == hipGraphNodeTypeKernel) {
            e = hipGraphKernelNodeGetParams(nodes[v], &op.k);
            if (e != hipSuccess) return fail("hipGraphKernelNodeGetParams", e);
            hipFunction_t hf = nullptr;
            // runtime-API launches record the host stub; module launches (none in this library) record the hipFunction_t
            op.kind = hipGetFuncBySymbol(&hf, op.k.func) == hipSuccess ? OP_KERNEL : OP_MODULE_KERNEL;
            op.fn = op.kind == OP_KERNEL ? hf : (hipFunction_t)op.k.func;
            (void)hipGetLastError();
            ++L->n_kernels;
            L->n_side += s;
        } else if (ty == hipGraphNodeTypeMemset) {
            e = hipGraphMemsetNodeGetParams(nodes[v], &op.ms);
            if (e != hipSuccess) return fail("hipGraphMemsetNodeGetParams", e);
            if (!(op.ms.height <= 1 && (op.ms.elementSize == 1 || op.ms.elementSize == 2 || op.ms.elementSize == 4)))
                return fail("2-D memset node", hipErrorNotSupported);
            op.kind = OP_MEMSET;
            ++L->n_other;
        } else if (ty == hipGraphNodeTypeMemcpy) {
            // ROCm 7.2 keeps a captured hipMemcpyAsync as a 1-D memcpy node whose (dst, src, bytes) have no getter: only nodes that
            // carry complete 3-D parameters can be re-issued.  The callers keep copies out of recorded steps (kernel copies).
            e = hipGraphMemcpyNodeGetParams(nodes[v], &op.cp);
            if (e != hipSuccess) return fail("hipGraphMemcpyNodeGetParams", e);
            if (op.cp.extent.width == 0 || op.cp.extent.width >= (1ull << 40) || op.cp.srcPtr.ptr == nullptr || op.cp.dstPtr.ptr == nullptr ||
                op.cp.srcArray != nullptr || op.cp.dstArray != nullptr || op.cp.extent.height > (1u << 20) || op.cp.extent.depth > (1u << 20))
                return fail("a 1-D memcpy node (hipMemcpyAsync under capture) cannot be read back; use a kernel copy", hipErrorNotSupported);
            if (getenv("DVQ_CMDLIST_DEBUG"))
                fprintf(stderr, "[cmdlist] memcpy node %d: src %p pitch %zu xs %zu ys %zu pos %zu,%zu,%zu  dst %p pitch %zu xs %zu ys %zu pos %zu,%zu,%zu  extent %zu x %zu x %zu kind %d\n",
                        v, op.cp.srcPtr.ptr, op.cp.srcPtr.pitch, op.cp.srcPtr.xsize, op.cp.srcPtr.ysize, op.cp.srcPos.x, op.cp.srcPos.y,
                        op.cp.srcPos.z, op.cp.dstPtr.ptr, op.cp.dstPtr.pitch, op.cp.dstPtr.xsize, op.cp.dstPtr.ysize, op.cp.dstPos.x,
                        op.cp.dstPos.y, op.cp.dstPos.z, op.cp.extent.width, op.cp.extent.height, op.cp.extent.depth, (int)op.cp.kind);
            op.kind = OP_MEMCPY;
            ++L->n_other;
        } else if (ty == hipGraphNodeTypeEmpty) {
            emit = false;
        } else {
            dvq_set_error("dvq_cmdlist_create: node type %d is not supported (kernel / memset / memcpy / empty only)", (int)ty);
            for (auto ev : L->events) (void)hipEventDestroy(ev);
            delete L;
            return DVQ_EINVAL;
        }
        if (emit) {
            L->ops.push_back(op);
            if (s == 1) last_side = (int)L->ops.size() - 1;
        }
        if (event_of[v] >= 0) {
            Op r{};
            r.kind = OP_RECORD, r.stream = s, r.event = event_of[v];
            L->ops.push_back(r);
            if (s == 1) last_side = (int)L->ops.size() - 1;
        }
    }
    L->side_open = last_side > last_join;
    if (L->side_open && hipEventCreateWithFlags(&L->tail_event, hipEventDisableTiming) != hipSuccess)
        return fail("hipEventCreate", hipErrorOutOfMemory);
    *out = (dvq_cmdlist_t)L;
    return DVQ_OK;
}

int dvq_cmdlist_replay(dvq_cmdlist_t list, dvq_stream_t main_stream, dvq_stream_t side_stream) {
    DVQ_REQUIRE(list != nullptr, DVQ_EINVAL, "dvq_cmdlist_replay: null list");
    CmdList* L = (CmdList*)list;
    hipStream_t st[2] = {(hipStream_t)main_stream, (hipStream_t)side_stream};
    DVQ_REQUIRE(L->n_side == 0 || st[1] != st[0], DVQ_EINVAL, "dvq_cmdlist_replay: the list needs a side stream");
    static const bool g_module_launch = [] {
        // "module": hipModuleLaunchKernel on the function resolved at build time instead of hipLaunchKernel(host stub) -- no faster
        // on ROCm 7.2 (36.6 vs 35.3 ms of host time per 2081-launch step)
        const char* e = getenv("DVQ_CMDLIST_LAUNCH");
        return e != nullptr && strcmp(e, "module") == 0;
    }();
    for (const Op& op : L->ops) {
        hipStream_t s = st[op.stream];
        switch (op.kind) {
            case OP_KERNEL:
                if (!g_module_launch) {
                    CL_HIP(hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, s), "hipLaunchKernel");
                    break;
                }
                [[fallthrough]];
            case OP_MODULE_KERNEL:
                CL_HIP(hipModuleLaunchKernel(op.fn, op.k.gridDim.x, op.k.gridDim.y, op.k.gridDim.z, op.k.blockDim.x,
                                             op.k.blockDim.y, op.k.blockDim.z, op.k.sharedMemBytes, s, op.k.kernelParams, op.k.extra),
                       "hipModuleLaunchKernel");
                break;
            case OP_MEMSET: {
                const size_t count = op.ms.width;
                if (op.ms.elementSize == 1) CL_HIP(hipMemsetAsync(op.ms.dst, (int)op.ms.value, count, s), "hipMemsetAsync");
                else if (op.ms.elementSize == 2) CL_HIP(hipMemsetD16Async((hipDeviceptr_t)op.ms.dst, (unsigned short)op.ms.value, count, s), "hipMemsetD16Async");
                else CL_HIP(hipMemsetD32Async((hipDeviceptr_t)op.ms.dst, (int)op.ms.value, count, s), "hipMemsetD32Async");
                break;
            }
            case OP_MEMCPY:
                if (op.cp.extent.height <= 1 && op.cp.extent.depth <= 1 && op.cp.srcArray == nullptr && op.cp.dstArray == nullptr)
                    CL_HIP(hipMemcpyAsync((char*)op.cp.dstPtr.ptr + op.cp.dstPos.x, (const char*)op.cp.srcPtr.ptr + op.cp.srcPos.x,
                                          op.cp.extent.width, op.cp.kind, s), "hipMemcpyAsync");
                else
                    CL_HIP(hipMemcpy3DAsync(&op.cp, s), "hipMemcpy3DAsync");
                break;
            case OP_RECORD:
                CL_HIP(hipEventRecord(L->events[op.event], s), "hipEventRecord");
                break;
            case OP_WAIT:
                CL_HIP(hipStreamWaitEvent(s, L->events[op.event], 0), "hipStreamWaitEvent");
                break;
        }
    }
    if (L->side_open) {        // the capture ended with a join that produced no node: whatever follows on main sees the side chain
        CL_HIP(hipEventRecord(L->tail_event, st[1]), "hipEventRecord");
        CL_HIP(hipStreamWaitEvent(st[0], L->tail_event, 0), "hipStreamWaitEvent");
    }
    return DVQ_OK;
}

int dvq_cmdlist_info(dvq_cmdlist_t list, int64_t* info4) {
    DVQ_REQUIRE(list != nullptr && info4 != nullptr, DVQ_EINVAL, "dvq_cmdlist_info: null pointer");
    CmdList* L = (CmdList*)list;
    info4[0] = L->n_kernels, info4[1] = L->n_side, info4[2] = L->n_sync, info4[3] = L->n_other | ((int64_t)L->side_open << 32);
    return DVQ_OK;
}

int dvq_cmdlist_destroy(dvq_cmdlist_t list) {
    if (list == nullptr) return DVQ_OK;
    CmdList* L = (CmdList*)list;
    for (auto ev : L->events)
        if (ev) (void)hipEventDestroy(ev);
    if (L->tail_event) (void)hipEventDestroy(L->tail_event);
    delete L;
    return DVQ_OK;
}

}  // extern "C"
