"""Sequential LCF kernel (phase B of the batched meta pass): time per call of 90 LCF steps with the rows of `n_seg` ranks, on one
workgroup and with one workgroup per rank's rows (device-side hand-over of the partial sums)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from copo_amd import _capi  # noqa: E402

dev = torch.device("cuda")
mb, n_mb = 512, 90
for n_seg in (1, 2, 4, 8):
    g = torch.Generator(device="cuda").manual_seed(1)
    en = torch.randn(n_seg, n_mb, mb, 2, device=dev, generator=g)
    w = torch.ones(n_seg, n_mb, mb, device=dev)
    eps = torch.randn(n_seg, n_mb, mb, device=dev, generator=g, dtype=torch.float64)
    denom = w.sum((0, 2)).contiguous()
    gv = torch.randn(n_mb, device=dev, generator=g, dtype=torch.float64)
    stats_in = torch.zeros(n_mb, 2, 8, device=dev)
    raw = torch.tensor([0.1, 1.3], dtype=torch.float64, device=dev)
    xchg = torch.zeros(256, dtype=torch.float64, device=dev)
    p = torch.tensor([0.05, -2.3], dtype=torch.float64, device=dev)
    adam = torch.zeros(5, dtype=torch.float64, device=dev)
    st = torch.zeros(7, dtype=torch.float64, device=dev)
    for wgs in sorted({1, n_seg, max(n_seg, 8)}):
        def call():
            _capi.check(_capi.lib.copo_meta_batch_lcf_f64(
                None, 0, 0, 0, None, en.data_ptr(), n_seg, w.data_ptr(), eps.data_ptr(), denom.data_ptr(), mb, n_mb, gv.data_ptr(),
                stats_in.data_ptr(), p.data_ptr(), raw.data_ptr(), adam.data_ptr(), 1e-4, st.data_ptr(), wgs, xchg.data_ptr(),
                _capi.current_stream()))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("rows of %d rank(s), %2d workgroup(s): %7.1f us per pass of %d LCF steps = %.2f us per step" % (n_seg, wgs, us, n_mb, us / n_mb))
