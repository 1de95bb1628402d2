import torch, sys
sys.path.insert(0, "/root/repo")
dev = torch.device("cuda")
main = torch.cuda.current_stream(dev)
probe = torch.zeros(64, device=dev)
big = torch.zeros(1 << 28, device=dev)
keep = []
for spin in ("sleep", "fill"):
    for i in range(10):
        cand = torch.cuda.Stream(device=dev); keep.append(cand)
        torch.cuda.synchronize()
        e0, em, es = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        if spin == "sleep":
            torch.cuda._sleep(400_000)
        else:
            for _ in range(4): big.fill_(1.0)
        em.record(main)
        with torch.cuda.stream(cand):
            cand.wait_event(e0)
            probe.add_(1.0)
            es.record(cand)
        torch.cuda.synchronize()
        print(spin, i, "stream", hex(cand.cuda_stream), "main %.3f ms side %.3f ms" % (e0.elapsed_time(em), e0.elapsed_time(es)))
