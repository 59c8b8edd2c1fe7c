"""Deprecation helpers with the call signatures of flow/utils/flow_warnings.py.

``deprecated(base, new_path)`` marks a class or function kept under an old module path: using it emits a
``PendingDeprecationWarning`` that names the new location.  Unlike the reference's decorator (which replaces a class by
a plain function, :33-68), a decorated class stays a class here -- ``issubclass`` / ``isinstance`` against the new
class keep working, which the env registry relies on."""
import functools
import inspect
import warnings


def _warn(message, stacklevel=3):
    warnings.simplefilter('always', PendingDeprecationWarning)
    warnings.warn(message, category=PendingDeprecationWarning, stacklevel=stacklevel)
    warnings.simplefilter('default', PendingDeprecationWarning)


def deprecated_attribute(obj, dep_from, dep_to):
    """flow_warnings.py:10-27: an attribute was renamed."""
    _warn("The attribute {} in {} is deprecated, use {} instead.".format(dep_from, obj.__class__.__name__, dep_to))


def deprecated(base, new_path):
    """flow_warnings.py:30-68: decorator for a class / function that moved to ``new_path``."""
    def decorator(obj):
        kind = "class" if inspect.isclass(obj) else "function"
        message = "The {} {}.{} is deprecated, use {} instead.".format(kind, base, obj.__name__, new_path)
        if inspect.isclass(obj):
            original_init = obj.__init__

            @functools.wraps(original_init)
            def init(self, *args, **kwargs):
                _warn(message)
                original_init(self, *args, **kwargs)
            obj.__init__ = init
            return obj

        @functools.wraps(obj)
        def wrapper(*args, **kwargs):
            _warn(message)
            return obj(*args, **kwargs)
        return wrapper
    return decorator
