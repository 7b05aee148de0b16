"""Do two torch streams overlap in this process at all?  Chains of torch.cuda._sleep kernels."""
import time
import torch
dev = torch.device("cuda", 0)
s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
print("stream handles", [hex(x.cuda_stream) for x in s])
def chain(st, n=100, cyc=400000):
    with torch.cuda.stream(st):
        for _ in range(n):
            torch.cuda._sleep(cyc)
for st in s:
    chain(st, 10)
torch.cuda.synchronize()
t0 = time.perf_counter(); chain(s[0]); torch.cuda.synchronize(); one = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(100):
    with torch.cuda.stream(s[0]):
        torch.cuda._sleep(400000)
    with torch.cuda.stream(s[1]):
        torch.cuda._sleep(400000)
torch.cuda.synchronize(); two = time.perf_counter() - t0
print(f"one chain {one*1e6/100:.2f} us/kernel; two chains {two*1e6/100:.2f} us/pair -> overlap x{2*one/two:.2f}")

# --- streams created directly through the HIP runtime (hipStreamCreateWithFlags, non-blocking), wrapped as ExternalStream
import ctypes as C
hip = C.CDLL("libamdhip64.so")
raw = []
for i in range(2):
    h = C.c_void_p()
    rc = hip.hipStreamCreateWithFlags(C.byref(h), 1)
    assert rc == 0
    raw.append(h.value)
s = [torch.cuda.ExternalStream(r, dev) for r in raw]
print("raw stream handles", [hex(r) for r in raw])
for st in s:
    chain(st, 10)
torch.cuda.synchronize()
t0 = time.perf_counter(); chain(s[0]); torch.cuda.synchronize(); one = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(100):
    with torch.cuda.stream(s[0]):
        torch.cuda._sleep(400000)
    with torch.cuda.stream(s[1]):
        torch.cuda._sleep(400000)
torch.cuda.synchronize(); two = time.perf_counter() - t0
print(f"raw HIP streams: one chain {one*1e6/100:.2f} us/kernel; two chains {two*1e6/100:.2f} us/pair -> overlap x{2*one/two:.2f}")
import os
print({k: v for k, v in os.environ.items() if k.startswith(("GPU_", "HIP_", "HSA_", "ROC", "AMD_"))})
