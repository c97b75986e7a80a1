class ParameterIndexOperations(object): pass
