"""MI355X-native CoPO rollout-and-update engine.  Package-level helpers mirror `copo/__init__.py:8-41` of the reference."""
import json
import numbers


class SafeFallbackEncoder(json.JSONEncoder):
    """Result dicts to JSON: NaN -> `nan_str`, numpy arrays -> lists, numpy scalars -> python numbers, the rest -> str."""

    def __init__(self, nan_str="null", **kwargs):
        super().__init__(**kwargs)
        self.nan_str = nan_str

    def default(self, value):
        import numpy as np
        try:
            if np.isnan(value):
                return self.nan_str
            if isinstance(value, np.ndarray):
                return value.tolist()
            if isinstance(value, numbers.Integral):
                return int(value)
            if isinstance(value, numbers.Number):
                return float(value)
            return super().default(value)
        except Exception:
            return str(value)


def pretty_print(result):
    """YAML text of a result dict without its `config` / `hist_stats` entries and without None values."""
    import yaml
    kept = {k: v for k, v in result.items() if v is not None and k not in ("config", "hist_stats")}
    return yaml.safe_dump(json.loads(json.dumps(kept, cls=SafeFallbackEncoder)), default_flow_style=False)
