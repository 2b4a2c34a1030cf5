"""Namespace mirroring ``jaxsim.api`` for the step path: ``import jaxsim_amd.api as js``
then ``js.model.step(model, data)``, ``js.data.JaxSimModelData.build(...)``,
``js.contact.estimate_good_contact_parameters(...)``."""

from . import contact, data, model, references  # noqa: F401
