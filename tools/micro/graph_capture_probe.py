#!/usr/bin/env python3
"""Why `KernelUpdateHead._forward_autograd` captures its chain graph BEFORE it touches the head's parameters (ROCm 7 / PyTorch 2.10):
`torch.cuda.make_graphed_callables(m, args)` faults inside hipStreamEndCapture when an EAGER autograd graph through m's parameters is
alive at capture time (their AccumulateGrad nodes are bound to the default stream; PyTorch warns about exactly this in backward).
    python tools/micro/graph_capture_probe.py ok      -> OK     (module never ran before the capture)
    python tools/micro/graph_capture_probe.py live    -> segfault (an output of an eager forward is still referenced)
    python tools/micro/graph_capture_probe.py freed   -> OK     (same eager forward, its output deleted first)"""
import sys
import torch
import torch.nn as nn

mode = sys.argv[1] if len(sys.argv) > 1 else 'ok'
dev = 'cuda:0'
m = nn.Sequential(nn.Linear(64, 2048), nn.ReLU(), nn.Linear(2048, 64)).to(dev)
args = (torch.randn(2, 20, 64, device=dev, requires_grad=True),)
if mode in ('live', 'freed'):
    o = m(*args)
    if mode == 'freed':
        del o
    torch.cuda.synchronize()
g = torch.cuda.make_graphed_callables(m, args, allow_unused_input=True)
g(*args).sum().backward()
torch.cuda.synchronize()
print('OK', mode)
