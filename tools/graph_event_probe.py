"""Probe: can a HIP event recorded as an EXTERNAL node inside a captured graph (hipEventRecordWithFlags(..., hipEventRecordExternal))
be used to time a kernel of a REPLAYED graph?  (torch refuses external events on ROCm; this goes to libamdhip64 directly.)"""
import ctypes as C
import torch
hip = C.CDLL("libamdhip64.so")
dev = 'cuda:0'
x = torch.randn(8192, 8192, device=dev)
s = torch.cuda.Stream()


def mk():
    e = C.c_void_p()
    rc = hip.hipEventCreate(C.byref(e))
    assert rc == 0, rc
    return e


e0, e1 = mk(), mk()
hip.hipEventRecordWithFlags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    y = x @ x
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        z = x * 2
        print('record rc', hip.hipEventRecordWithFlags(e0, C.c_void_p(s.cuda_stream), 1))
        y = x @ x
        print('record rc', hip.hipEventRecordWithFlags(e1, C.c_void_p(s.cuda_stream), 1))
        w = y + 1
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    ms = C.c_float(-1)
    rc = hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    print('replay', i, 'rc', rc, 'elapsed ms', ms.value)
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); y = x @ x; b.record(); torch.cuda.synchronize(); print('eager matmul ms', a.elapsed_time(b))
