"""Alias so that `import TorchRegister as tr` (ref:README.md:26) resolves to the MI355X-native package."""
from torchregister_amd import *  # noqa: F401,F403
from torchregister_amd import __version__  # noqa: F401
