from . import parameter_core, index_operations, pickleable, lists_and_dicts, observable_array, observable
