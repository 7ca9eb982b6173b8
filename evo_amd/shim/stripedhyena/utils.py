from evo_amd.sh.utils import dotdict  # noqa: F401
