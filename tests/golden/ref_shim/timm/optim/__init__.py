from . import optim_factory  # noqa: F401
