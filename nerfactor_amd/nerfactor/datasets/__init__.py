"""Dataset plugin lookup by name (`dataset = <name>` ini key), like the reference's
datasets/__init__.py:18-20."""
from importlib import import_module


def get_dataset_class(name):
    return import_module(__name__ + '.' + name).Dataset
