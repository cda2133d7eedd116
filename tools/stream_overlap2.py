#!/usr/bin/env python
"""Diagnostic: cross-stream concurrency characterisation (pure torch kernels)."""
import torch, time, os
dev = torch.device("cuda:0")
A = torch.cuda.Stream(device=dev); B = torch.cuda.Stream(device=dev)
x = torch.zeros(1 << 16, device=dev); y = torch.zeros(1 << 16, device=dev)
M = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize()
def chain(s, t, n):
    with torch.cuda.stream(s):
        for _ in range(n): t.add_(1.0)
def big(s):
    with torch.cuda.stream(s):
        for _ in range(4): (M @ M)
def run(name, fa, fb):
    for rep in range(3):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record(A); fa(); e[1].record(A)
        e[2].record(B); fb(); e[3].record(B)
        torch.cuda.synchronize()
        print(f"{name}: A {e[0].elapsed_time(e[1])*1e3:8.1f} us | B starts +{e[0].elapsed_time(e[2])*1e3:8.1f}, B {e[2].elapsed_time(e[3])*1e3:8.1f} us, B ends +{e[0].elapsed_time(e[3])*1e3:8.1f}")
run("A=chain150, B=chain20", lambda: chain(A, x, 150), lambda: chain(B, y, 20))
run("A=big matmul x4, B=chain20", lambda: big(A), lambda: chain(B, y, 20))
run("A=chain150, B=chain150", lambda: chain(A, x, 150), lambda: chain(B, y, 150))
# interleaved submission
def inter():
    for _ in range(150):
        with torch.cuda.stream(A): x.add_(1.0)
        with torch.cuda.stream(B): y.add_(1.0)
for rep in range(3):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(A); e[2].record(B); inter(); e[1].record(A); e[3].record(B)
    torch.cuda.synchronize()
    print(f"interleaved 150+150: A {e[0].elapsed_time(e[1])*1e3:8.1f} us, B {e[2].elapsed_time(e[3])*1e3:8.1f} us")
torch.cuda.synchronize(); t=time.perf_counter(); chain(A,x,150); torch.cuda.synchronize(); print("chain150 alone", (time.perf_counter()-t)*1e6)
