"""The two nethook helpers the hot path uses (R/editor/nethook.py:415-432): lookup of a module or
parameter by its dotted HF name.  The Trace/TraceDict hook machinery of the reference is torch-hook
based and is not part of the native path (DESIGN.md 'out of scope')."""


def get_module(model, name):
    for n, m in model.named_modules():
        if n == name:
            return m
    raise LookupError(name)


def get_parameter(model, name):
    for n, p in model.named_parameters():
        if n == name:
            return p
    raise LookupError(name)
