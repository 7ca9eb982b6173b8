from evo_amd.sh.cache import InferenceParams, RecurrentInferenceParams  # noqa: F401
