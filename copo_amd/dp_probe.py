"""Child process of `dist.probe_tile_exchange`: does the data-parallel tile exchange work between the GPUs of this node?

Joins its own process group (the parent shifted MASTER_PORT), builds a learner of the parent's shape (hidden width, input width, number of nets) twice from the same seed, gives every
rank different rows and takes the same minibatch steps (a) with the tile exchange inside the weight-gradient kernel, in
captured chains exactly like the trainer's, and (b) with the gradient all-reduce of torch.distributed + the flat Adam kernel.
Exit code 0 = (a) left bit-identical parameters on all ranks, they agree with (b) to rounding, and no wait timed out.
A hang stays inside this process: the parent kills it by PID after its timeout and keeps the RCCL loop."""
import os
import sys


def _learner(mode, rank):
    import torch
    from copo_amd.engine import Box, Postprocessing, SampleBatch, TorchDiagGaussian
    from copo_amd.torch_copo import algo_copo, algo_ippo
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_rllib_compatible_env
    os.environ["COPO_DP_EXCHANGE"] = mode
    nets = int(os.environ.get("COPO_DP_PROBE_NETS", "2"))      # policy + value nets of the parent's learner: 2 (IPPO / CCPPO) or 4 (CoPO)
    cfg = algo_copo.CoPOConfig() if nets >= 4 else algo_ippo.IPPOConfig()
    # the learner's own shape (the parent passes it on): the weight-gradient grid -- how many workgroups wait for peers at once --
    # is what has to work, not a toy
    hidden, obs_dim = int(os.environ.get("COPO_DP_PROBE_HIDDEN", "64")), int(os.environ.get("COPO_DP_PROBE_OBS", "20"))
    mb = _mb()
    cfg.update_from_dict(dict(env=get_rllib_compatible_env(MultiAgentIntersectionEnv), seed=11, sgd_minibatch_size=mb,
                              model={"fcnet_hiddens": [hidden, hidden]}))
    cfg.validate()
    pol = (algo_copo.CoPOPolicy if nets >= 4 else algo_ippo.IPPOPolicy)(Box(-1, 1, (obs_dim,)), Box(-1, 1, (2,)), cfg)
    R = (700 + 90 * rank) * mb // 128    # unequal shards: the smaller ranks pad with zero-weight rows (~6-7 minibatches each)
    g = torch.Generator().manual_seed(100 + rank)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()  # noqa: E731
    b = SampleBatch()
    b[SampleBatch.OBS] = rn(R, obs_dim) * 0.5
    b[SampleBatch.ACTIONS] = rn(R, 2) * 0.8
    di = torch.cat([rn(R, 2) * 0.3, rn(R, 2) * 0.2 - 0.2], 1)
    b[SampleBatch.ACTION_DIST_INPUTS] = di
    b[SampleBatch.ACTION_LOGP] = TorchDiagGaussian(di).logp(b[SampleBatch.ACTIONS])
    b[Postprocessing.ADVANTAGES] = rn(R)
    b[SampleBatch.VF_PREDS] = rn(R)
    b[Postprocessing.VALUE_TARGETS] = b[SampleBatch.VF_PREDS] + rn(R)
    if nets >= 4:
        for k in (algo_copo.NEI_VALUES, algo_copo.GLOBAL_VALUES, algo_copo.NEI_ADVANTAGE, algo_copo.GLOBAL_ADVANTAGES, "normalized_advantages"):
            b[k] = rn(R)
        b[algo_copo.NEI_TARGET] = b[algo_copo.NEI_VALUES] + rn(R)
        b[algo_copo.GLOBAL_TARGET] = b[algo_copo.GLOBAL_VALUES] + rn(R)
    return pol, b, R


def _mb():
    """The parent's minibatch size: 512-row minibatches take the buffer-load instantiation of the weight-gradient kernel, and it
    is that kernel whose exchange has to work."""
    return int(os.environ.get("COPO_DP_PROBE_MB", "128"))


def main():
    import torch
    import torch.distributed as td
    from copo_amd import dist as D
    rank, local, world = D.init_from_env("cuda")
    assert td.is_initialized() and world > 1
    torch.cuda.set_device(local)
    outs = {}
    for mode in ("tile", "rccl"):
        pol, batch, R = _learner(mode, rank)
        assert pol.fused is not None
        pol.prepare_sgd(batch, R, _mb())
        idx = torch.arange(R, device="cuda")
        B_all = D.all_gather_int(R, "cuda")
        torch.manual_seed(77)                 # the same shuffle keys in both modes
        t0 = __import__("time").perf_counter()
        st = pol.run_sgd(idx, R, B_all, _mb(), 6)      # 6 epochs x 7 minibatches: captured chains of 16 + single steps
        if os.environ.get("COPO_DP_PROBE_VERBOSE"):
            print("rank %d mode %s: %d steps in %.2f s" % (rank, mode, st["num_sgd_steps"], __import__("time").perf_counter() - t0), flush=True)
        assert (pol._dp_mode == "tile") == (mode == "tile") and st["num_sgd_steps"] > 32
        torch.cuda.synchronize()
        outs[mode] = pol.fused.flat.flat.clone()
        if pol._tile is not None:
            pol._tile.status()
            pol._tile.close()
    flat = outs["tile"]
    every = [torch.empty_like(flat) for _ in range(world)]
    td.all_gather(every, flat)
    same = all(bool(torch.equal(every[0], e)) for e in every)
    moved = float((flat - outs["rccl"]).abs().max()) <= 2e-4 and float(flat.abs().sum()) > 0
    if os.environ.get("COPO_DP_PROBE_VERBOSE"):
        print("rank %d: identical on all ranks %s, max |tile - rccl| %.3g" % (rank, same, float((flat - outs["rccl"]).abs().max())), flush=True)
    td.barrier()
    td.destroy_process_group()
    return 0 if (same and moved) else 3


if __name__ == "__main__":
    sys.exit(main())
