class Pickleable(object): pass
