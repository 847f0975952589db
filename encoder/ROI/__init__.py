"""Import surface of the reference (implicit namespace package there).  See INTEGRATION.md.
The package spans the same-named directories further along sys.path, so modules this repository does not replace
resolve to the reference's own files."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
