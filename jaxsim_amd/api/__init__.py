"""Namespace mirroring ``jaxsim.api`` for the step path: ``import jaxsim_amd.api as js``
then ``js.model.step(model, data)``, ``js.data.JaxSimModelData.build(...)``,
``js.contact.estimate_good_contact_parameters(...)``, ``js.ode.system_dynamics(model, data)``."""

from . import contact, data, model, ode, references  # noqa: F401
