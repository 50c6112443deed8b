"""`python sicelore-2.1_amd <sub-command> ...`: the jar's command line (cli.py).  The directory name is not an identifier, so the
package is imported under the name the rest of the repository uses for it (`sicelore_amd`)."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if "sicelore_amd" not in sys.modules:
    spec = importlib.util.spec_from_file_location("sicelore_amd", os.path.join(_HERE, "__init__.py"), submodule_search_locations=[_HERE])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sicelore_amd"] = mod
    spec.loader.exec_module(mod)
from sicelore_amd import cli  # noqa: E402

sys.exit(cli.main())
