#!/usr/bin/env python3
"""Host-to-device and device-to-host rates of pinned 64 MB chunks, alone and with reader threads copying page-cache
data into the other staging buffer at the same time (what the reference ingest does).  usage: python tools/h2d_probe.py"""
import os, sys, threading, time
import torch
CH = 64 << 20
N = 48
dev = torch.empty(CH * 2, dtype=torch.uint8, device="cuda")
pin = [torch.empty(CH, dtype=torch.uint8).pin_memory() for _ in range(2)]
src = torch.randint(0, 255, (CH,), dtype=torch.uint8)

def run(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {N * CH / dt / 1e9:.1f} GB/s", flush=True)

def h2d():
    for i in range(N):
        dev[:CH].copy_(pin[i & 1], non_blocking=True)

def d2h():
    for i in range(N):
        pin[i & 1].copy_(dev[:CH], non_blocking=True)

def h2d_with_memcpy(threads):
    stop = [False]
    def worker(k):
        part = CH // threads
        while not stop[0]:
            pin[1][k * part:(k + 1) * part].copy_(src[k * part:(k + 1) * part])
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    for t in ths: t.start()
    for i in range(N):
        dev[:CH].copy_(pin[0], non_blocking=True)
    torch.cuda.synchronize()
    stop[0] = True
    for t in ths: t.join()

run("H2D alone", h2d); run("H2D alone", h2d); run("D2H alone", d2h)
for th in (4, 16):
    run(f"H2D while {th} threads fill the other pinned buffer", lambda: h2d_with_memcpy(th))
