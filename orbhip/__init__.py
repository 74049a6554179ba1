"""Importable alias of the product package.

The package directory is named after the project (`awesome-orb-slam3-3dvisioncraft-version_amd/`), which is not
a valid Python identifier; `import orbhip` resolves to that directory."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "awesome-orb-slam3-3dvisioncraft-version_amd"))
from ._pkg import *  # noqa: F401,F403,E402
from ._pkg import __all__  # noqa: F401,E402
