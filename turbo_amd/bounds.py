"""Ordered parameter bounds: the shape ``latent_bounds`` has at the plugin boundary.

The auxiliary optimiser only reads ``latent_bounds.ordered`` -- ``[(name, min, max), ...]`` in
parameter order (turbo/bounds.py:4-29) -- so any object with that attribute works, the reference's
own ``turbo.Bounds`` included.  This one exists so the plugins can be driven without the
reference installed (tests, bench, stand-alone use); it offers the same lookups.
"""


class Bounds:
    def __init__(self, ordered):
        triples = []
        position = {}
        for entry in ordered:
            name, low, high = entry
            if name in position:
                raise ValueError('parameter {!r} listed twice'.format(name))
            if not low <= high:
                raise ValueError('parameter {!r}: lower bound {} above upper bound {}'.format(name, low, high))
            position[name] = len(triples)
            triples.append((name, low, high))
        self.ordered = triples
        self._position = position

    # the reference's attribute names, derived on demand
    @property
    def params(self):
        return set(self._position)

    @property
    def associative(self):
        return {name: (low, high) for name, low, high in self.ordered}

    def __len__(self):
        return len(self.ordered)

    def __iter__(self):
        return iter(self.ordered)

    def __repr__(self):
        return 'Bounds({!r})'.format(self.ordered)

    def get(self, param):
        """(min, max) of one parameter"""
        _, low, high = self.ordered[self._position[param]]
        return low, high

    def get_param_index(self, param):
        """position of a parameter in a point (1, D)"""
        return self._position[param]

    def low_high(self):
        """two D-vectors, for the device-side candidate generator"""
        return [b[1] for b in self.ordered], [b[2] for b in self.ordered]
