"""Ordered parameter bounds -- the shape ``latent_bounds`` has at the plugin boundary
(turbo/bounds.py:4-29: ``.ordered`` is ``[(name, min, max), ...]``).  Any object with an
``ordered`` attribute of that form works, including the reference's own ``turbo.Bounds``."""


class Bounds:
    def __init__(self, ordered):
        self.ordered = list(ordered)
        self.params = set(b[0] for b in self.ordered)
        self.associative = {b[0]: (b[1], b[2]) for b in self.ordered}

    def __len__(self):
        return len(self.ordered)

    def get(self, param):
        return self.associative[param]

    def get_param_index(self, param):
        for i, b in enumerate(self.ordered):
            if param == b[0]:
                return i
        raise KeyError()
