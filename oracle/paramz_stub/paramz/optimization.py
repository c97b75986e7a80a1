class Optimizer(object): pass
