"""evo_amd: a MI355X-native (gfx950) StripedHyena forward engine behind evo-design/evo's API.

    import evo_amd as evo
    m = evo.Evo("evo-1-131k-base", device="cuda:0", weights="synthetic")
    evo.score_sequences(["ACGT..."], m.model, m.tokenizer)

Same public names as the reference package [REF evo/__init__.py:3-6].  `install_shim()` makes
`import stripedhyena` resolve to this engine so the unmodified reference `evo` package runs on it.
"""
import os as _os
import sys as _sys

from .version import version as __version__
from .models import Evo
from .generation import generate
from .scoring import score_sequences, positional_entropies

SHIM_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "shim")


def install_shim() -> str:
    """Put the in-repo `stripedhyena` drop-in package at the front of sys.path."""
    if SHIM_PATH not in _sys.path:
        _sys.path.insert(0, SHIM_PATH)
    return SHIM_PATH
