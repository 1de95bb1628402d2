"""Feasibility of running the nets' SGD chains independently (round 6): the four nets of a CoPO step share no data, so their row-pass ->
weight-gradient chains could run on separate streams and overlap one net's memory-bound weight gradients with another's row pass.  Stand-in
that needs no kernel change: two 2-net learners (IPPO: policy + value) stepping CONCURRENTLY on two streams against ONE 4-net learner (CoPO).
    python scripts/bench_fused_streams.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused_learner import _make, _dense_batch

R, mb, odim = 82000, 512, 92
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def prep(algo):
    pol = _make(algo, "none", odim, fused=True, mb=mb)
    batch = _dense_batch(pol, R, odim)
    idx = torch.arange(R, device="cuda")
    pol.prepare_sgd(batch, R, mb)
    pol.plan_epoch(idx, R, [R], mb)
    rs = pol.fused.gather_epoch(pol._row_sources, 160)
    return pol, rs


def run(pols, streams, steps):
    for (pol, rs), st in zip(pols, streams):
        with torch.cuda.stream(st):
            for _ in range(10):
                pol.fused.step(rs, stats=pol.fused.stats)
    torch.cuda.synchronize()
    for pol, _ in pols:
        pol._row_sources["k"].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for st in streams:
        st.wait_event(e0)
    for i in range(steps):
        for (pol, rs), st in zip(pols, streams):
            if i % 128 == 0:
                with torch.cuda.stream(st):
                    pol._row_sources["k"].zero_()
            with torch.cuda.stream(st):
                pol.fused.step(rs, stats=pol.fused.stats)
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps, (time.perf_counter() - t0) * 1e6 / steps


copo = prep("copo")
a, b = prep("ippo"), prep("ippo")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
print("one 4-net learner, one stream:                 %.1f us per step (host %.1f)" % run([copo], [s0], n))
print("one 2-net learner, one stream:                 %.1f us per step (host %.1f)" % run([a], [s0], n))
print("two 2-net learners, ONE stream (back to back): %.1f us per pair of steps (host %.1f)" % run([a, b], [s0, s0], n))
print("two 2-net learners, TWO streams (concurrent):  %.1f us per pair of steps (host %.1f)" % run([a, b], [s0, s1], n))
