"""Evaluation path: the reference's policy-function and checkpoint wire formats
(`copo/eval/get_policy_function*.py`) and a vectorised evaluation on the HIP simulator in place of `RecorderEnv` +
`evaluate_population.py`."""
