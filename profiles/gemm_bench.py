"""Isolated launches of the GEMM kernels through the C ABI's debug entry point (casv_debug_gemm: random operands, optional
row gather, plain or fused-LSTM epilogue):   python profiles/gemm_bench.py [number of shapes] [iterations]
This is the program behind profiles/r01_gemm_pmc.txt (there still under its old path scratch/gemm_bench.py)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cor_asv_ann_amd.engine import HipEngine
from cor_asv_ann_amd import _native as nv
eng = HipEngine(1, 32, 8)
shapes = [(1, 8192, 2048, 1024, 1), (1, 8192, 2048, 1536, 1), (1, 8192, 2048, 1024, 0), (0, 8192, 2048, 1024, 0), (1, 1024, 2048, 1024, 0), (0, 8192, 512, 512, 0), (0, 8192, 256, 512, 0), (1, 16384, 2048, 1024, 0)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for lstm, M, N, K, gather in shapes:
    ms = ctypes.c_double()
    nv.check(eng.lib.casv_debug_gemm(eng.handle, lstm, M, N, K, gather, iters, ctypes.byref(ms)))
    print('lstm=%d M=%d N=%d K=%d gather=%d: %.1f us  %.1f TFLOP/s' % (lstm, M, N, K, gather, ms.value*1e3, 2.0*M*N*K/ms.value/1e9), flush=True)
