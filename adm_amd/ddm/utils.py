"""Host utilities the reference's drivers import from ddm.utils (/root/reference/ddm/utils.py:94-161):
dynamic construction by dotted name, so YAML `class_name:` entries keep working."""
import importlib


def get_obj_by_name(name: str):
    parts = name.split(".")
    for i in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        try:
            for p in parts[i:]:
                obj = getattr(obj, p)
            return obj
        except AttributeError:
            continue
    raise ImportError(name)


def construct_class_by_name(*args, class_name: str = None, **kwargs):
    assert class_name is not None
    return get_obj_by_name(class_name)(*args, **kwargs)


def exists(x):
    return x is not None


def default(val, d):
    return val if val is not None else (d() if callable(d) else d)


def cycle(dl):
    while True:
        for data in dl:
            yield data
