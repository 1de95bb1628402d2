"""MI355X-native CoPO rollout-and-update engine (the hot path only; the reference's package-level
result printers -- `copo/__init__.py` -- are out of scope)."""
