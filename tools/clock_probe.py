#!/usr/bin/env python3
"""In-kernel clock of the 256x256 GEMM K loop (DIAGNOSTIC library, tools/diag_lib.py: GSTVD_GEMM_ST=3; with GSTVD_DIAG_ABLATE=1:
MFMA + LDS reads only, =2: LDS-DMA only).  Runs ~2 s of back-to-back launches on random data, then reads the per-workgroup stamps."""
import ctypes as C, os, sys, time
os.environ["GSTVD_GEMM_ST"] = "3"
os.environ.setdefault("GSTVD_GEMM256_NIU", "4")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import diag_lib
diag_lib.use()
import torch
from gst_visdial_amd import ops, _lib
M, N, K = 4096, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 3072
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        ops.gemm(A, B, Cc, M, N, K)
    torch.cuda.synchronize()
buf = (C.c_uint64 * (256 * 4))()
_lib.check("gstvd_debug_gemm_clock", _lib.load().gstvd_debug_gemm_clock(buf, 256 * 4))
ghz = sorted(buf[4 * i] / max(buf[4 * i + 1], 1) * 0.1 for i in range(256))
cyc = sorted(buf[4 * i] / max(buf[4 * i + 2], 1) for i in range(256))
us = sorted(buf[4 * i + 1] / 100.0 / max(buf[4 * i + 2], 1) for i in range(256))
print("ABLATE=%s K=%d: in-kernel clock median %.2f GHz (min %.2f max %.2f); K loop %.0f shader cycles/step, %.3f us/step (median over 256 workgroups)"
      % (os.environ.get("GSTVD_DIAG_ABLATE", "0"), K, ghz[128], ghz[0], ghz[-1], cyc[128], us[128]))
