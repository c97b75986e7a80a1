class Parameterizable(object):
    def __init__(self, name=None, *a, **kw):
        self.name = name
        self.parameters = []

    def link_parameter(self, p, index=None):
        if not hasattr(self, 'parameters'):
            self.parameters = []
        if index is None:
            self.parameters.append(p)
        else:
            self.parameters.insert(index, p)

    def link_parameters(self, *ps):
        for p in ps:
            self.link_parameter(p)

    def unlink_parameter(self, p):
        self.parameters = [q for q in self.parameters if q is not p]

    def add_index_operation(self, *a, **k):
        pass

    def copy(self):
        """paramz's Parameterized.copy: a deep copy detached from any parent (used by GPy.kern.Add, add.py:28)."""
        import copy
        return copy.deepcopy(self)

    def __getstate__(self):
        return dict(self.__dict__)

    def __setstate__(self, state):
        self.__dict__.update(state)

    @property
    def is_fixed(self):
        return False

    def parameters_changed(self):
        pass

    @property
    def size(self):
        return sum(int(p.size) for p in self.parameters)


class Observable(object):
    pass
