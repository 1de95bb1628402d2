"""Module path of the reference (`copo/eval/get_policy_function_from_checkpoint.py:15-63`); the code lives in
`checkpoint_io`."""
from .checkpoint_io import get_lcf_from_checkpoint, get_policy_function_from_checkpoint  # noqa: F401
