"""Import-surface alias: `fiber` is the package name the reference's entry points import (coarse_grained/run.py:7-9
`from fiber.modules import FIBERTransformerSS`).  Everything lives in `fiber_amd`; this package only re-exports the module
surface of the hot path so that run.py's model import resolves unedited when this repository is on sys.path."""
