class Transformation(object):
    def __init__(self, *a, **k): pass
class Logexp(Transformation): pass
class Logistic(Transformation): pass
class Exponent(Transformation): pass
class NormalTheta(Transformation): pass
class NormalNaturalAntti(Transformation): pass
class NormalNaturalThroughTheta(Transformation): pass
class NormalEta(Transformation): pass
class NegativeLogexp(Transformation): pass
class LogexpNeg(Transformation): pass
class Square(Transformation): pass
class LogexpClipped(Transformation): pass
class NegativeExponent(Transformation): pass
__fixed__ = "fixed"
