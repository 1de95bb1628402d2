"""CLI / misc helpers with the reference's names (`copo/torch_copo/utils/utils.py:182-235`); no Ray underneath."""
import argparse
import logging
import os


def get_train_parser():
    """The six flags of the reference's launch scripts (utils/utils.py:206-214)."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--exp-name", type=str, default="")
    parser.add_argument("--num-gpus", type=int, default=0)
    parser.add_argument("--num-seeds", type=int, default=3)
    parser.add_argument("--num-cpus-per-worker", type=float, default=0.5)
    parser.add_argument("--num-gpus-per-trial", type=float, default=0.25)
    parser.add_argument("--test", action="store_true")
    return parser


def initialize_ray(local_mode=False, num_gpus=None, test_mode=False, **kwargs):
    """There is no Ray runtime in this build; kept so that reference-shaped scripts run unchanged.  One process
    drives one GPU; launch with `python -m torch.distributed.run --nproc-per-node N` for N GPUs."""
    os.environ["OMP_NUM_THREADS"] = "1"
    from copo_amd import dist as D
    D.init_from_env()
    print("Successfully initialize the MI355X host loop (world size %d)." % D.world_size())


def setup_logger(debug=False):
    logging.basicConfig(level=logging.DEBUG if debug else logging.WARNING,
                        format="%(asctime)s - %(filename)s[line:%(lineno)d] - %(levelname)s: %(message)s")


def pretty_print(result):
    import json

    def clean(v):
        if isinstance(v, dict):
            return {k: clean(x) for k, x in v.items()}
        try:
            return float(v)
        except (TypeError, ValueError):
            return str(v)
    return json.dumps(clean(result), indent=2, sort_keys=True)
