#!/usr/bin/env python3
"""Which edge was missing?  (diagnostic, GPU box, ONE run.)

Round 3 found that with rac_absmax_fwd resetting its scale word by hipMemsetD32Async the SECOND replay of the captured 140-node
step read a stale word; the isolated four-node probe of round 4 did not reproduce it.  This tool captures the whole
step once with that memset variant (a diagnostic build: tools/build_variant.sh absmemset "-DRAC_ABSMAX_MEMSET" conv3x3.hip), keeps
the hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()), and reads the graph itself: every node's type, the edge
list (hipGraphGetNodes / hipGraphGetEdges), the MEMSET node's direct predecessors and successors, and whether the kernel that the
stream order puts right behind it (absmax_kernel, which atomically maxes into the word) is reachable from it.  It also replays the
instantiated graph a few times and reports whether replays differ -- from the evidence of ONE run, no repetition of the failure.

usage:  RACFORMER_HIP_LIB=build/lib_absmemset.so python3 tools/graph_memset_edges.py [out.json]"""
import ctypes
import json
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from racformer_amd import synthetic as syn  # noqa: E402
from racformer_amd.fused import scratch_namespace  # noqa: E402

NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "graph_memset_edges.json")
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    cfg = syn.F8
    head = bench.build_head(cfg, dev)
    pyramid = [f.to(dev) for f in syn.make_pyramid(cfg, 0)]
    lss, radar = syn.make_bev(cfg, 0, 0).to(dev), syn.make_bev(cfg, 0, 1).to(dev)
    metas = syn.make_img_metas(cfg)
    head.transformer.decoder.stage_metas(metas, 1, dev)

    def run():
        preds = head(list(pyramid), lss, radar, metas)
        return head.get_detections_fixed(preds)

    with scratch_namespace(("memset_probe", 0)), torch.no_grad():
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                eager = run().clone()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            det = run()
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    types = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        types[nodes[i]] = t.value
    ne = ctypes.c_size_t(0)
    assert hip.hipGraphGetEdges(raw, None, None, ctypes.byref(ne)) == 0
    fr, to = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
    assert hip.hipGraphGetEdges(raw, fr, to, ctypes.byref(ne)) == 0
    succ, pred = defaultdict(list), defaultdict(list)
    for a, b in zip(fr, to):
        succ[a].append(b)
        pred[b].append(a)
    order = {nodes[i]: i for i in range(n.value)}          # hipGraphGetNodes returns the nodes in creation (= capture) order

    def kname(node):
        if types[node] != 0:
            return NODE_TYPES.get(types[node], str(types[node]))

        class KP(ctypes.Structure):
            _fields_ = [("blockDim", ctypes.c_uint * 3), ("extra", ctypes.c_void_p), ("func", ctypes.c_void_p), ("gridDim", ctypes.c_uint * 3),
                        ("kernelParams", ctypes.c_void_p), ("sharedMemBytes", ctypes.c_uint)]
        kp = KP()
        if hip.hipGraphKernelNodeGetParams(ctypes.c_void_p(node), ctypes.byref(kp)) != 0:
            return "kernel(?)"
        hip.hipKernelNameRefByPtr.restype = ctypes.c_char_p
        name = hip.hipKernelNameRefByPtr(ctypes.c_void_p(kp.func), None)
        return (name.decode(errors="replace") if name else "kernel")[:60] + f" grid={kp.gridDim[0]}"

    memsets = [nd for nd in nodes if types[nd] == 2]
    report = {"nodes": n.value, "edges": ne.value, "node_types": {NODE_TYPES.get(k, str(k)): sum(1 for v in types.values() if v == k) for k in set(types.values())},
              "memset_nodes": []}
    roots = [nd for nd in nodes if not pred[nd]]
    report["root_nodes"] = [f"#{order[r]} {kname(r)}" for r in roots]

    def reachable(a, b):
        seen, stack = set(), [a]
        while stack:
            x = stack.pop()
            if x == b:
                return True
            if x in seen:
                continue
            seen.add(x)
            stack += succ[x]
        return False

    for m in memsets:
        i = order[m]
        nxt = nodes[i + 1] if i + 1 < n.value else None
        prv = nodes[i - 1] if i > 0 else None
        report["memset_nodes"].append({
            "capture_index": i,
            "predecessors": [f"#{order[p]} {kname(p)}" for p in pred[m]],
            "successors": [f"#{order[s]} {kname(s)}" for s in succ[m]],
            "node_captured_just_before": f"#{order[prv]} {kname(prv)}" if prv is not None else None,
            "node_captured_just_after": f"#{order[nxt]} {kname(nxt)}" if nxt is not None else None,
            "before_reaches_memset": bool(prv is not None and reachable(prv, m)),
            "memset_reaches_after": bool(nxt is not None and reachable(m, nxt))})
    # linearity of the whole capture: consecutive nodes without a path between them
    gaps = [f"#{i} {kname(nodes[i])} -/-> #{i + 1} {kname(nodes[i + 1])}" for i in range(n.value - 1) if not reachable(nodes[i], nodes[i + 1])]
    report["consecutive_nodes_without_a_path"] = gaps
    dot = os.path.join(os.path.dirname(out_path), "graph_memset.dot")
    report["dot_rc"] = int(hip.hipGraphDebugDotPrint(raw, dot.encode(), ctypes.c_uint(0)))
    # instantiate and replay: do replays reproduce the eager detections?
    g.instantiate()
    same = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        same.append(bool(torch.equal(det, eager)))
    report["replays_equal_eager"] = same
    json.dump(report, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != "consecutive_nodes_without_a_path"}, indent=1))
    print("consecutive nodes without a path:", len(gaps), gaps[:6])


if __name__ == "__main__":
    main()
