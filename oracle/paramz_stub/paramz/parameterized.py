class ParametersChangedMeta(type):
    pass
