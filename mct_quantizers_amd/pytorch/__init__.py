

def __getattr__(name):
    # the reference's sub-modules are attributes of its packages (its top level imports them all); here they resolve on
    # first access, so dotted paths like mct_quantizers.pytorch.quantize_wrapper work without an explicit import
    import importlib
    try:
        return importlib.import_module(f"{__name__}.{name}")
    except ModuleNotFoundError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
