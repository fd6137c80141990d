class ModelCatalog:
    _registry = {}

    @staticmethod
    def register_custom_model(name, cls):
        ModelCatalog._registry[name] = cls
