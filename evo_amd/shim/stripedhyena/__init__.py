"""Drop-in `stripedhyena` package: put `<repo>/evo_amd/shim` on PYTHONPATH (or call `evo_amd.install_shim()`)
and the UNMODIFIED evo-design/evo host code imports the MI355X-native engine where it expects
stripedhyena==0.2.2 [REF evo/models.py:8-9; evo/scoring.py:5; evo/generation.py:6-7]."""
