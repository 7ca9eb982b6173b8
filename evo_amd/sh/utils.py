"""`stripedhyena.utils` mirror: dotdict [REF evo/models.py:8,142]."""


class dotdict(dict):
    """dict with attribute access; a missing key reads as None (the reference relies on that for every
    hyper-parameter its yml leaves out).  Extra keyword arguments become keys, which is how the
    reference's `dotdict(config, Loader=yaml.FullLoader)` ends up with a stray 'Loader' entry
    [REF evo/models.py:142]."""

    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __getstate__(self):
        return dict(self)

    def __setstate__(self, state):
        self.update(state)
