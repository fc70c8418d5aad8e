"""Mirror of the reference package of the same name: the modules present here are the MI355X-native
ones, every other submodule falls through to the reference's package further along sys.path."""
try:
    from _fallthrough import extend as _extend          # dropin/ on sys.path (INTEGRATION.md route 1)
    __path__ = _extend(list(__path__), __name__, __file__)
    if __name__ == "datasets":
        # DataLoader workers of the unmodified scripts call the HIP library: they must not be forked children of a
        # process that has initialised the GPU (see _fallthrough.worker_start_method)
        from _fallthrough import worker_start_method as _worker_start_method
        _worker_start_method()
except ImportError:                                      # imported as <package>.dropin.<name>: nothing to fall through to
    pass
