"""Phase stamps of one workgroup of the batched meta weight-gradient GEMM (COPO_RP_DBG=1024)."""
import os, sys, ctypes as C
os.environ["COPO_RP_DBG"] = "1024"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import torch
import bench
tr = bench.make_trainer(256, 40, graphs=True)
for _ in range(4):
    tr.train()
torch.cuda.synchronize()
from copo_amd import _capi
buf = (C.c_ulonglong * 16)()
_capi.lib.copo_debug_rowpass_stamps.argtypes = [C.c_void_p]
_capi.lib.copo_debug_rowpass_stamps(buf)
t = list(buf)
names = ["row tables", "first fetch issued", "slab 0 (stash, sync, MFMA)", "slabs 1..7", "epilogue stores"]
for i, n in enumerate(names):
    print("%-28s %7.2f us" % (n, (t[i + 1] - t[i]) / 100.0))
print("%-28s %7.2f us" % ("workgroup total", (t[5] - t[0]) / 100.0))
tr.stop()
