from evo_amd.sh.sample import sample, modify_logits_for_top_k_filtering, modify_logits_for_top_p_filtering  # noqa: F401
