"""Mirror of the reference's `models` package (lib/models/__init__.py:1-4): `models.GATOR / GAT / MDR .get_model`."""
from . import GAT, MDR  # noqa: F401  (order matters: GATOR imports both)
from . import GATOR  # noqa: F401
