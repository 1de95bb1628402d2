"""IPPO: independent PPO with one shared policy ("default") for every agent.

Counterpart of the reference's `copo/torch_copo/algo_ippo.py` (IPPOConfig :17-75, IPPOPolicy.loss :78-172,
IPPOTrainer :175-182) with RLlib's PPO machinery (sampling, SGD epochs, KL adaptation, Adam) replaced
by a thin host loop over the HIP simulator and torch-ROCm: see `copo_amd/trainer.py`.
"""
import torch
import torch.nn as nn

from copo_amd.engine import AlgorithmConfig, Box, Postprocessing, SampleBatch, build_mlp, reduce_mean_valid_fn
from copo_amd.trainer import PPOPolicyBase, VecTrainer


class IPPOConfig(AlgorithmConfig):
    """Defaults of algo_ippo.py:17-42: minibatch 512, fragment 200, batch 2000 env-steps, 5 epochs, lr 3e-4,
    clip 0.2, lambda 0.95, vf_clip 100, old-style clipped value loss."""

    def __init__(self, algo_class=None):
        super().__init__(algo_class=algo_class or IPPOTrainer)
        self.sgd_minibatch_size = 512
        self.rollout_fragment_length = 200
        self.train_batch_size = 2000
        self.num_sgd_iter = 5
        self.lr = 3e-4
        self.clip_param = 0.2
        self.lambda_ = 0.95
        self.num_cpus_per_worker = 0.2
        self.num_cpus_for_local_worker = 1
        self.num_rollout_workers = 5
        self.framework_str = "torch"
        self.vf_clip_param = 100
        self.old_value_loss = True

    def validate(self):
        """Reads the spaces from the env (algo_ippo.py:44-75) and pins the single shared policy "default"."""
        super().validate()
        from copo_amd.torch_copo.utils.env_wrappers import lookup_env
        env_cls = lookup_env(self["env"])
        obs_space, act_space = env_cls.spaces_for(self["env_config"])
        assert isinstance(obs_space, Box)
        self.update_from_dict({"multiagent": dict(policies={"default": (None, obs_space, act_space, {})},
                                                  policy_mapping_fn=lambda x: "default")})
        self.observation_space, self.action_space = obs_space, act_space
        return self


class FullyConnectedModel(nn.Module):
    """Policy MLP obs -> 256 -> 256 -> 2A (tanh; normc 1.0 / 0.01) plus a separate value MLP on the same obs:
    the RLlib default net IPPO uses; key names as in best_checkpoints/ippo_*.npz."""

    def __init__(self, obs_space, action_space, num_outputs, model_config, name="fc_model"):
        super().__init__()
        self.obs_space, self.action_space, self.num_outputs = obs_space, action_space, num_outputs
        self.model_config, self.name = model_config, name
        hiddens = list(model_config.get("fcnet_hiddens", [256, 256]))
        act = model_config.get("fcnet_activation", "tanh")
        odim = int(obs_space.shape[0])
        self._hidden_layers, self._logits, _ = build_mlp(odim, hiddens, act, num_outputs, 0.01)
        self._value_branch_separate, self._value_branch, _ = build_mlp(self.value_input_dim(), hiddens, act, 1, 0.01)
        self.tower_stats = {}
        self._last_obs = None

    def value_input_dim(self):
        return int(self.obs_space.shape[0])

    def forward(self, input_dict, state=None, seq_lens=None):
        obs = input_dict["obs"] if isinstance(input_dict, dict) else input_dict
        # (the parameters' dtype: fp32 in every trainer; a float64 copy of the model is what the tests' exact reference uses)
        obs = obs.to(self._logits._model[0].weight.dtype).reshape(obs.shape[0], -1)
        self._last_obs = obs
        return self._logits(self._hidden_layers(obs)), (state or [])

    def value_function(self):
        return self._value_branch(self._value_branch_separate(self._last_obs)).reshape(-1)

    def policy_parameters(self):
        return list(self._hidden_layers.parameters()) + list(self._logits.parameters())


def clipped_value_loss(current_vf, prev_vf, value_target, vf_clip_param, old_value_loss):
    """`old_value_loss=True`: max((v-T)^2, (v_prev + clip(v-v_prev, +-c) - T)^2); else clamp((v-T)^2, 0, c)
    (algo_ippo.py:139-151)."""
    if old_value_loss:
        l1 = torch.pow(current_vf - value_target, 2.0)
        clipped = prev_vf + torch.clamp(current_vf - prev_vf, -vf_clip_param, vf_clip_param)
        return torch.max(l1, torch.pow(clipped - value_target, 2.0))
    return torch.clamp(torch.pow(current_vf - value_target, 2.0), 0, vf_clip_param)


class IPPOPolicy(PPOPolicyBase):
    model_class = FullyConnectedModel

    def critic_obs(self, train_batch):
        return train_batch[SampleBatch.OBS]

    def values_for(self, model, train_batch):
        return model.value_function()

    def loss(self, model, dist_class, train_batch):
        """PPO-clip surrogate + clipped value loss + KL penalty (algo_ippo.py:78-172)."""
        mean = reduce_mean_valid_fn(train_batch)
        logits, _ = model(train_batch)
        curr = dist_class(logits, model)
        prev = dist_class(train_batch[SampleBatch.ACTION_DIST_INPUTS], model)
        ratio = torch.exp(curr.logp(train_batch[SampleBatch.ACTIONS]) - train_batch[SampleBatch.ACTION_LOGP])
        use_kl = self.config["kl_coeff"] > 0.0
        mean_kl = mean(prev.kl(curr)) if use_kl else torch.zeros((), device=ratio.device)
        entropy = curr.entropy()
        adv = train_batch[Postprocessing.ADVANTAGES]
        clip = self.config["clip_param"]
        surrogate = torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip))
        assert self.config["use_critic"]
        value_out = self.values_for(model, train_batch)
        vf_loss = clipped_value_loss(value_out, train_batch[SampleBatch.VF_PREDS],
                                     train_batch[Postprocessing.VALUE_TARGETS], self.config["vf_clip_param"],
                                     self.config["old_value_loss"])
        total = mean(-surrogate + self.config["vf_loss_coeff"] * vf_loss - self.entropy_coeff * entropy)
        if use_kl:
            total = total + self.kl_coeff * mean_kl
        st = model.tower_stats
        st["total_loss"], st["mean_policy_loss"], st["mean_vf_loss"] = total, mean(-surrogate), mean(vf_loss)
        st["vf_explained_var"] = torch.zeros((), device=ratio.device)
        st["mean_entropy"], st["mean_kl_loss"] = mean(entropy), mean_kl
        return total


class IPPOTrainer(VecTrainer):
    _name = "IPPO"

    @classmethod
    def get_default_config(cls):
        return IPPOConfig()

    def get_default_policy_class(self, config):
        assert config["framework"] == "torch"
        return IPPOPolicy


# ========== Test scripts ==========
def _test(stop=2000, local_dir=None):
    """The reference's `_test()` (algo_ippo.py:186-234): a tiny configuration through `train()`, "does it run" end to end."""
    from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
    from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_rllib_compatible_env
    from copo_amd.torch_copo.utils.train import train
    from copo_amd.torch_copo.utils.utils import get_train_parser
    args, _ = get_train_parser().parse_known_args()
    config = dict(env=get_rllib_compatible_env(MultiAgentIntersectionEnv), env_config=dict(num_agents=8), num_envs=4, train_batch_size=100,
                  rollout_fragment_length=20, sgd_minibatch_size=30)
    return train(IPPOTrainer, config=config, checkpoint_freq=0, keep_checkpoints_num=0, stop={"timesteps_total": stop},
                 num_gpus=args.num_gpus, num_seeds=1, max_failures=0, exp_name=args.exp_name or "test_ippo",
                 custom_callback=MultiAgentDrivingCallbacks, test_mode=True, local_mode=True, local_dir=local_dir)


if __name__ == "__main__":
    _test()
