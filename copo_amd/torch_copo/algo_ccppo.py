"""CCPPO / MAPPO: PPO with a centralised critic over ego + neighbour observations (and actions).

Counterpart of the reference's `copo/torch_copo/algo_ccppo.py`: CCPPOConfig (:37-52),
get_centralized_critic_obs_dim (:55-71), CCModel (:74-219), concat / mean-field fusion (:225-311),
CCPPOPolicy.postprocess_trajectory / loss (:322-472).  The per-sample Python triple loop of the fusion is a
HIP gather kernel here (`copo_cc_fuse_{mf,concat}_f32`), applied to all [T, E, N] rows at once.
"""
import torch

from copo_amd.engine import SampleBatch
from copo_amd.torch_copo.algo_ippo import FullyConnectedModel, IPPOConfig, IPPOPolicy, IPPOTrainer

CENTRALIZED_CRITIC_OBS = "centralized_critic_obs"
COUNTERFACTUAL = "counterfactual"


class CCPPOConfig(IPPOConfig):
    def __init__(self, algo_class=None):
        super().__init__(algo_class=algo_class or CCPPOTrainer)
        self.counterfactual = True
        self.num_neighbours = 4
        self.fuse_mode = "mf"          # "concat" | "mf" | "none"
        self.mf_nei_distance = 10
        self.old_value_loss = True
        self.update_from_dict({"model": {"custom_model": "cc_model"}})

    def validate(self):
        assert self["fuse_mode"] in ("mf", "concat", "none")
        cmc = dict(self.model.get("custom_model_config") or {})
        cmc.update(fuse_mode=self["fuse_mode"], counterfactual=self["counterfactual"],
                   num_neighbours=self["num_neighbours"])
        self.model = {**self.model, "custom_model_config": cmc}
        # the simulator evaluates the mean-field radius with the neighbour list (exact fp64 compare)
        self.env_config = {**self.env_config, "mf_distance": float(self["mf_nei_distance"]),
                           "nbr_k": max(int(self.env_config.get("nbr_k", 8)), int(self["num_neighbours"]))}
        super().validate()
        return self


def get_centralized_critic_obs_dim(observation_space_shape, action_space_shape, counterfactual, num_neighbours,
                                   fuse_mode):
    """O (none) | 2O [+A] (mf) | O + k(O [+A]) (concat) -- e.g. 92 / 186 / 468 for O=92, A=2 (algo_ccppo.py:55-71)."""
    if fuse_mode == "concat":
        k = num_neighbours
    elif fuse_mode == "mf":
        k = 1
    elif fuse_mode == "none":
        k = 0
    else:
        raise ValueError("Unknown fuse mode: ", fuse_mode)
    dim = (k + 1) * observation_space_shape.shape[0]
    if counterfactual:
        dim += k * action_space_shape.shape[0]
    return dim


class CCModel(FullyConnectedModel):
    """Policy MLP on the agent's own obs; value MLP on the centralised-critic obs."""

    def value_input_dim(self):
        return self.get_centralized_critic_obs_dim()

    def get_centralized_critic_obs_dim(self):
        c = self.model_config["custom_model_config"]
        return get_centralized_critic_obs_dim(self.obs_space, self.action_space, c["counterfactual"],
                                              c["num_neighbours"], c["fuse_mode"])

    def value_function(self):
        raise ValueError("Centralized Value Function should not be called directly! "
                         "Call central_value_function(cobs) instead!")

    def central_value_function(self, obs):
        return self._value_branch(self._value_branch_separate(obs)).reshape(-1)


def concat_ccppo_process(policy, sample_batch, out=None):
    """Neighbour k of the distance-sorted list -> slot k (not compacted; absent neighbour => zeros)."""
    from copo_amd import ops
    b = sample_batch
    obs = b[SampleBatch.OBS]
    T, E, N, O = obs.shape
    return ops.cc_fuse("concat", obs.reshape(T * E, N, O), b[SampleBatch.ACTIONS].reshape(T * E, N, -1),
                       b[SampleBatch.FLAGS].reshape(T * E, N), b["nbr_idx"].reshape(T * E, N, -1),
                       b["nbr_cnt"].reshape(T * E, N), policy.config[COUNTERFACTUAL], policy.config["num_neighbours"],
                       out=out).view(T, E, N, -1)


def mean_field_ccppo_process(policy, sample_batch, out=None):
    """Mean obs (and action) over the neighbours within `mf_nei_distance` that have a row at the same step."""
    from copo_amd import ops
    b = sample_batch
    obs = b[SampleBatch.OBS]
    T, E, N, O = obs.shape
    return ops.cc_fuse("mf", obs.reshape(T * E, N, O), b[SampleBatch.ACTIONS].reshape(T * E, N, -1),
                       b[SampleBatch.FLAGS].reshape(T * E, N), b["nbr_idx"].reshape(T * E, N, -1),
                       b["mf_cnt"].reshape(T * E, N), policy.config[COUNTERFACTUAL], out=out).view(T, E, N, -1)


def get_ccppo_env(env_class):
    from copo_amd.torch_copo.utils.env_wrappers import get_ccenv, get_rllib_compatible_env
    return get_rllib_compatible_env(get_ccenv(env_class))


class CCPPOPolicy(IPPOPolicy):
    model_class = CCModel

    def __init__(self, observation_space, action_space, config):
        super().__init__(observation_space, action_space, config)
        self.centralized_critic_obs_dim = self.model.get_centralized_critic_obs_dim()
        self._cc_buf = None

    # algo_ccppo.py:362-365 / algo_copo.py:492-496 bootstrap a cut trajectory with VF_PREDS[-1], the value of its last ROW:
    # the centralised critic observation of the next step does not exist yet.  Critics that read nothing but the agent's own
    # observation (fuse_mode "none": CoPO's three heads by default) use the exact bootstrap instead -- the shortcut touches a
    # trajectory once or twice with the reference's 200-step fragments, but every row with the 8-step fragments of 256
    # lockstep scenes (trainer.PPOPolicyBase.bootstrap_next_obs; `bootstrap_next_obs: False` restores the shortcut).
    def bootstrap_next_obs(self):
        return super().bootstrap_next_obs() and int(self.model.value_input_dim()) == int(self.observation_space.shape[0])

    def critic_obs_dense(self, batch):
        mode = self.config["fuse_mode"]
        if mode == "none":
            return batch[SampleBatch.OBS]
        obs = batch[SampleBatch.OBS]
        shape = tuple(obs.shape[:3]) + (self.centralized_critic_obs_dim,)
        if self._cc_buf is None or tuple(self._cc_buf.shape) != shape:
            self._cc_buf = torch.empty(shape, dtype=torch.float32, device=obs.device)
        out = self._cc_buf.view(shape[0] * shape[1], shape[2], shape[3])
        return (concat_ccppo_process if mode == "concat" else mean_field_ccppo_process)(self, batch, out)

    def values_for(self, model, train_batch):
        return model.central_value_function(train_batch[CENTRALIZED_CRITIC_OBS])


class CCPPOTrainer(IPPOTrainer):
    _name = "CCPPO"

    @classmethod
    def get_default_config(cls):
        return CCPPOConfig()

    def get_default_policy_class(self, config):
        assert config["framework"] == "torch"
        return CCPPOPolicy


# ========== Test scripts ==========
def _test(stop=2000, local_dir=None):
    """The reference's `_test()` (algo_ccppo.py:485-528): a tiny configuration through `train()`, "does it run" end to end."""
    from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv
    from copo_amd.torch_copo.utils.train import train
    from copo_amd.torch_copo.utils.utils import get_train_parser
    args, _ = get_train_parser().parse_known_args()
    config = dict(env=get_ccppo_env(MultiAgentIntersectionEnv), env_config=dict(num_agents=8), num_envs=4, train_batch_size=100,
                  rollout_fragment_length=20, sgd_minibatch_size=30, fuse_mode="mf")
    return train(CCPPOTrainer, config=config, checkpoint_freq=0, keep_checkpoints_num=0, stop={"timesteps_total": stop},
                 num_gpus=args.num_gpus, num_seeds=1, max_failures=0, exp_name=args.exp_name or "test_ccppo",
                 custom_callback=MultiAgentDrivingCallbacks, test_mode=True, local_mode=True, local_dir=local_dir)


if __name__ == "__main__":
    _test()
