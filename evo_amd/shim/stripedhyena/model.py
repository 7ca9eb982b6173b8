from evo_amd.sh.model import StripedHyena  # noqa: F401
