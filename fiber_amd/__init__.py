"""fiber_amd -- MI355X-native (gfx950) implementation of FIBER's coarse-grained fused-backbone path.

Only the hot path lives here: HIP kernels + C ABI (csrc/, include/fiber_hip.h), their ctypes binding (lib.py),
autograd wrappers (ops.py) and the host-side mirror of the reference's module surface (modules/).
"""
__all__ = ["lib", "ops", "modules"]
