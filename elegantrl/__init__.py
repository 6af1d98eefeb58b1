"""`import elegantrl` from a checkout of this repository: hands the name to `elegantrl_amd` (see elegantrl_amd/compat.py)."""
from elegantrl_amd.compat import install as _install

_install()          # sys.modules["elegantrl"] is elegantrl_amd from here on; submodules resolve through the import hook
