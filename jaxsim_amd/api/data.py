"""``jaxsim.api.data`` mirror (``src/jaxsim/api/data.py``)."""

from ..data import JaxSimModelData, random_model_data  # noqa: F401
