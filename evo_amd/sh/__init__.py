"""Host-side mirror of the `stripedhyena` package surface that evo-design/evo imports
[REF evo/models.py:8-9; evo/scoring.py:5; evo/generation.py:6-7]."""
