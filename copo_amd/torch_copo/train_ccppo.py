"""CCPPO launch script, same shape as the reference's `copo/torch_copo/train_ccppo.py:10-54`."""
from copo_amd.engine import grid_search
from copo_amd.torch_copo.algo_ccppo import COUNTERFACTUAL, CCPPOTrainer, get_ccppo_env
from copo_amd.torch_copo.utils.callbacks import MultiAgentDrivingCallbacks
from copo_amd.torch_copo.utils.env_wrappers import MultiAgentIntersectionEnv
from copo_amd.torch_copo.utils.train import train
from copo_amd.torch_copo.utils.utils import get_train_parser

if __name__ == "__main__":
    parser = get_train_parser()
    parser.add_argument("--num-envs", type=int, default=256)
    parser.add_argument("--stop", type=int, default=100_0000)
    args = parser.parse_args()
    config = dict(
        env=grid_search([get_ccppo_env(MultiAgentIntersectionEnv)]),
        env_config=dict(neighbours_distance=40),
        num_gpus=0.25 if args.num_gpus != 0 else 0,
        num_envs=args.num_envs,
        fuse_mode=grid_search(["mf", "concat"]),
        **{COUNTERFACTUAL: grid_search([True])},
    )
    train(CCPPOTrainer, exp_name=args.exp_name or "TEST", keep_checkpoints_num=5, stop=args.stop, config=config,
          num_gpus=args.num_gpus, num_seeds=1, custom_callback=MultiAgentDrivingCallbacks, test_mode=args.test)
