"""How many kernels does a host call enqueue?  Counted without a profiler: the call is captured into a hipGraph
(torch.cuda.graph keeps torch's allocator informed) and the graph's nodes are listed through the HIP runtime itself
(hipGraphGetNodes / hipGraphNodeGetType) — every kernel launch, memset and memcpy the call makes is one node."""
import ctypes

import torch

HIP_GRAPH_NODE_TYPE_KERNEL = 0          # hipGraphNodeTypeKernel (hip_runtime_api.h)


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not loadable")


def graph_node_types(fn, device, warm=True):
    """Runs fn() once eagerly (code objects resident, allocator warm), then captures ONE fn() call and returns the list
    of node types of the captured graph."""
    hip = _hip()
    dev = torch.device(device)
    if warm:
        fn()
        torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            keep = fn()                  # (whatever it returns stays alive until the nodes are counted)
    torch.cuda.current_stream(dev).wait_stream(side)
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(raw, None, ctypes.byref(n))
    assert rc == 0, "hipGraphGetNodes -> %d" % rc
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    rc = hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n))
    assert rc == 0, "hipGraphGetNodes -> %d" % rc
    types = []
    for k in range(n.value):
        t = ctypes.c_int(-1)
        rc = hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[k]), ctypes.byref(t))
        assert rc == 0, "hipGraphNodeGetType -> %d" % rc
        types.append(t.value)
    del keep
    return types


def kernels_enqueued(fn, device="cuda", warm=True):
    """(kernel nodes, all nodes) of one fn() call."""
    types = graph_node_types(fn, device, warm)
    return sum(1 for t in types if t == HIP_GRAPH_NODE_TYPE_KERNEL), len(types)
