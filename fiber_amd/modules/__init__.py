from .fiber_module import FIBERTransformerSS  # noqa: F401  (reference: fiber/modules/__init__.py:1)
