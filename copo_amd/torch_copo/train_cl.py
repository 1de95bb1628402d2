"""Curriculum-learning baseline launch script: IPPO on `ChangeNEnv` with the population schedule of
`copo/algo_ippo/ippo_cl.py:41-78` (launch shape of `copo/train_all_cl.py`)."""
from copo_amd.engine import grid_search
from copo_amd.torch_copo.algo_ippo import IPPOTrainer
from copo_amd.torch_copo.utils.callbacks import get_change_n_callback
from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv, get_change_n_env, get_rllib_compatible_env
from copo_amd.torch_copo.utils.train import train
from copo_amd.torch_copo.utils.utils import get_train_parser

if __name__ == "__main__":
    parser = get_train_parser()
    parser.add_argument("--num-envs", type=int, default=256)
    parser.add_argument("--stop", type=int, default=100_0000)
    args = parser.parse_args()
    config = dict(
        env=grid_search([get_rllib_compatible_env(get_change_n_env(MultiAgentIntersectionEnv))]),
        env_config=dict(),
        num_gpus=0.25 if args.num_gpus != 0 else 0,
        num_envs=args.num_envs,
    )
    train(IPPOTrainer, exp_name=args.exp_name or "TEST", keep_checkpoints_num=5, stop=args.stop, config=config,
          num_gpus=args.num_gpus, num_seeds=1, custom_callback=get_change_n_callback(args.stop), test_mode=args.test)
