"""Micro-benchmark of the fused learner step (eager launches, HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("COPO_RP_DBG"):          # phase stamps live in the profiling build only (make -C copo_amd/csrc prof SKIP=0)
    import copo_amd._libsel as _S
    _S.PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "copo_amd", "lib", "libcopo_hip_prof_0.so")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused_learner import _make, _dense_batch

R, mb, odim = 82000, int(os.environ.get("COPO_BENCH_MB", "512")), 92      # COPO_BENCH_MB: rows per minibatch (512 = the reference's)
pol = _make(os.environ.get("COPO_BENCH_ALGO", "copo"), "none", odim, fused=True, mb=mb)      # ippo: two nets -> 210 weight-gradient tiles, one per CU
batch = _dense_batch(pol, R, odim)
idx = torch.arange(R, device="cuda")
pol.prepare_sgd(batch, R, mb)
pol.plan_epoch(idx, R, [R], mb)
RS = pol._row_sources if os.environ.get("COPO_BENCH_TABLE") else pol.fused.gather_epoch(pol._row_sources, 160)      # the trainer's way: rows in minibatch order
if os.environ.get("COPO_BENCH_NOK"):
    RS = dict(RS, k=None)          # experiment: no device-side minibatch counter (every step takes minibatch 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for _ in range(10):
    pol.fused.step(RS, stats=pol.fused.stats)
torch.cuda.synchronize()
pol._row_sources["k"].zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
KC = os.environ.get("COPO_BENCH_KCONST")      # experiment: the minibatch index of step i read from a table that is never written (bump off)
kc = torch.arange(128, dtype=torch.int64, device="cuda") if KC else None
rs_i = [dict(RS, k=kc[i:i + 1]) for i in range(128)] if KC else None
for i in range(n):
    if KC:
        pol.fused.step(rs_i[i % 128], stats=pol.fused.stats, bump_index=False)
        continue
    if i % 128 == 0:
        pol._row_sources["k"].zero_()          # stay inside the epoch plan (161 minibatches)
    pol.fused.step(RS, stats=pol.fused.stats)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("fused sgd step: %.1f us gpu, %.1f us host enqueue" % (e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n))
if int(os.environ.get("COPO_RP_DBG", "0")) & 256:
    import ctypes as C
    from copo_amd import _capi
    buf = (C.c_ulonglong * 16)()
    _capi.lib.copo_debug_rowpass_stamps.argtypes = [C.c_void_p]
    _capi.lib.copo_debug_rowpass_stamps(buf)
    t = list(buf)
    names = ["prologue(loads X,w3)", "F1+tanh", "F2+tanh", "head+loss+stats", "dz2", "Bx+epilogue"]
    idx = [0, 1, 2, 3, 4, 5, 9]
    print("rowpass phases of workgroup (0,0), 100 MHz wall clock:")
    for i, nme in enumerate(names):
        print("  %-22s %6.2f us" % (nme, (t[idx[i + 1]] - t[idx[i]]) / 100.0))
    print("  total                  %6.2f us" % ((t[9] - t[0]) / 100.0))
if int(os.environ.get("COPO_RP_DBG", "0")) & 512:
    import ctypes as C
    from copo_amd import _capi
    buf = (C.c_ulonglong * 16)()
    _capi.lib.copo_debug_rowpass_stamps.argtypes = [C.c_void_p]
    _capi.lib.copo_debug_rowpass_stamps(buf)
    t = list(buf)
    names = ["state loads issued, k, (srow)", "MFMA loop (ring)", "LDS partials + sync", "fold + Adam + stores", "mirror write"]
    print("wgrad phases of workgroup (5,0), 100 MHz wall clock:")
    for i, nme in enumerate(names):
        print("  %-30s %6.2f us" % (nme, (t[i + 1] - t[i]) / 100.0))
    print("  total                          %6.2f us" % ((t[5] - t[0]) / 100.0))

if int(os.environ.get("COPO_RP_DBG", "0")) & 2048:
    import ctypes as C
    import numpy as np
    from copo_amd import _capi
    buf = (C.c_ulonglong * 4096)()
    _capi.lib.copo_debug_wg_times.argtypes = [C.c_void_p]
    _capi.lib.copo_debug_wg_times(buf)
    t = np.array(list(buf), dtype=np.float64).reshape(2, 1024, 2) / 100.0      # us
    for k, name in enumerate(("rowpass", "wgrad+adam")):
        m = t[k, :, 0] > 0
        st, en = t[k, m, 0], t[k, m, 1]
        t0 = st.min()
        print("%-11s %4d workgroups: first start 0.0, last start %.2f, first end %.2f, last end %.2f us; lifetime mean %.2f max %.2f" % (
            name, int(m.sum()), st.max() - t0, en.min() - t0, en.max() - t0, (en - st).mean(), (en - st).max()))
        print("   start-time quantiles (us):", [round(float(np.quantile(st - t0, q)), 2) for q in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)],
              " workgroups started within 1.5 us: %d" % int((st - t0 < 1.5).sum()))
        order = np.argsort(en)[::-1][:5]
        print("   slowest:", [(int(np.nonzero(m)[0][i]), round(float(st[i] - t0), 2), round(float(en[i] - t0), 2)) for i in order])
    print("rowpass start -> wgrad start %.2f us; rowpass last end -> wgrad first start %.2f us; wgrad last end -> (next) " % (
        t[1][t[1, :, 0] > 0, 0].min() - t[0][t[0, :, 0] > 0, 0].min(), t[1][t[1, :, 0] > 0, 0].min() - t[0][t[0, :, 0] > 0, 1].max()))
