"""Library path override for profiling scripts (see _capi.LIB_PATH); None = the shipped libcopo_hip.so."""
PATH = None
