"""`from fiber.modules import FIBERTransformerSS` (reference coarse_grained/fiber/modules/__init__.py:1), served by the
MI355X-native implementation.  The sub-module names callers use (`fiber.modules.fiber_module`, `.objectives`,
`.fiber_utils`, `.heads`, `.swin_transformer`, `.roberta`, `.swin_helpers`) resolve to the fiber_amd modules."""
import sys

from fiber_amd.modules import FIBERTransformerSS, fiber_module, fiber_utils, heads, objectives, roberta, swin_helpers, swin_transformer

for _name, _mod in (("fiber_module", fiber_module), ("fiber_utils", fiber_utils), ("heads", heads), ("objectives", objectives),
                    ("roberta", roberta), ("swin_helpers", swin_helpers), ("swin_transformer", swin_transformer)):
    sys.modules[f"{__name__}.{_name}"] = _mod

__all__ = ["FIBERTransformerSS"]
