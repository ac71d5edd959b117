"""Plugin lookup by name, like the reference's models/__init__.py:18-20 (`model = <name>` ini key)."""
from importlib import import_module


def get_model_class(name):
    return import_module(__name__ + '.' + name).Model
