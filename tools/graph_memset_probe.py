#!/usr/bin/env python3
"""Is a MEMSET node inside a captured HIP graph ordered with the kernel nodes around it?  (GPU box.)

usage: python3 tools/graph_memset_probe.py [out_prefix]

Round 3 found rac_absmax_fwd's scale word holding another tensor's bytes on the second replay of a plan when the word was
reset with hipMemsetD32Async inside the capture, and replaced the memset by a one-thread kernel.  This probe captures the
smallest graph with that shape on a stream --

    K1: kernel, writes a[:] = 3            MEMSET: word <- 0 (hipMemsetD32Async, a memset NODE)
    K2: kernel, word += a[0]               K3: kernel, out[i] = word   (i = replay counter slot)

-- replays it N times and (a) prints what every replay read (3.0 each time if the memset is ordered between K1/K2 of the
same replay and after K3 of the previous one; a growing or stale value otherwise), (b) writes the graph as DOT
(hipGraphDebugDotPrint through torch's CUDAGraph.debug_dump): the node list with their types and the dependency edges the
capture recorded."""
import ctypes
import os
import sys

import torch


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/graph_memset_probe"
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetD32Async.restype = ctypes.c_int
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    a = torch.zeros(1 << 20, device=dev)
    word = torch.full((1,), 123.0, device=dev)
    outs = torch.zeros(64, device=dev)
    slot = torch.zeros(1, dtype=torch.long, device=dev)

    def body():
        a.fill_(3.0)                                                                       # K1
        rc = hip.hipMemsetD32Async(ctypes.c_void_p(word.data_ptr()), 0, 1,
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))  # MEMSET node
        assert rc == 0, rc
        word.add_(a[:1])                                                                   # K2
        outs.index_copy_(0, slot, word)                                                    # K3
        slot.add_(1)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    slot.zero_(); outs.zero_(); word.fill_(123.0)
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g):
        body()
    dot = prefix + ".dot"
    try:
        g.debug_dump(dot)
    except Exception as e:  # noqa: BLE001
        print("debug_dump failed:", e)
    n = 16
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    vals = outs[:n].tolist()
    print("word read by K3 in replays 0..%d: %s" % (n - 1, vals))
    print("ordered (3.0 every replay):", all(v == 3.0 for v in vals))
    if os.path.exists(dot):
        txt = open(dot).read()
        print("---- %s (%d bytes)" % (dot, len(txt)))
        print(txt[:6000])


if __name__ == "__main__":
    main()
