"""Host-side ``JaxSimModel`` mirror: immutable model container + parameter classes.

Mirrors the reference's public surface for the step path
(``src/jaxsim/api/model.py:46-330,674-799``; parameter classes:
``rbda/contacts/soft.py:24-123``, ``rbda/actuation/common.py:10-19``,
``terrain/terrain.py:65-124``).  The object is plain Python + NumPy; the device copy
of the tables lives behind the C-ABI (``include/jaxsim_amd.h``: ``jxs_model_create``)
and is created lazily the first time a kernel entry point needs it.
"""

from __future__ import annotations

import contextlib
import copy
import dataclasses
import enum
import pathlib

import numpy as np

from .kin_dyn_parameters import KinDynParameters
from .parsers import urdf as urdf_parser

STANDARD_GRAVITY = 9.81  # src/jaxsim/math/__init__.py:14


class VelRepr(enum.IntEnum):
    """Velocity representations (``src/jaxsim/api/common.py:39-47``); values are the C-ABI enum."""

    Inertial = 0
    Body = 1
    Mixed = 2


@dataclasses.dataclass(frozen=True)
class SoftContactsParams:
    """``SoftContactsParams`` (``src/jaxsim/rbda/contacts/soft.py:24-123``)."""

    K: float = 1e6
    D: float = 2000.0
    mu: float = 0.5
    p: float = 0.5
    q: float = 0.5

    @classmethod
    def build(cls, *, K=1e6, D=2_000, mu=0.5, p=0.5, q=0.5, **kwargs):
        return cls(K=float(K), D=float(D), mu=float(mu), p=float(p), q=float(q))

    def valid(self) -> bool:
        return all(v >= 0.0 for v in (self.K, self.D, self.mu, self.p, self.q))

    @classmethod
    def build_default_from_jaxsim_model(
        cls,
        model: "JaxSimModel",
        *,
        stiffness=None,
        damping=None,
        standard_gravity=STANDARD_GRAVITY,
        static_friction_coefficient=0.5,
        max_penetration=0.001,
        number_of_active_collidable_points_steady_state=1,
        damping_ratio=1.0,
        p=0.5,
        q=0.5,
    ):
        """``build_default_from_jaxsim_model`` (``rbda/contacts/common.py:88-168``)."""
        m = float(np.sum(model.kin_dyn_parameters.link_mass))
        if stiffness is None:
            f_average = m * standard_gravity / number_of_active_collidable_points_steady_state
            stiffness = float(np.clip(f_average / np.power(max_penetration, 1 + p), 0, 1e6))
        if damping is None:
            damping = float(np.clip(damping_ratio * 2 * np.sqrt(stiffness * m), 0, 1e4))
        return cls.build(K=stiffness, D=damping, mu=static_friction_coefficient, p=p, q=q)


@dataclasses.dataclass(frozen=True)
class ActuationParams:
    """``ActuationParams`` (``src/jaxsim/rbda/actuation/common.py:10-19``)."""

    torque_max: float = 3000.0
    omega_th: float = 30.0
    omega_max: float = 100.0
    enable_friction: bool = True


@dataclasses.dataclass(frozen=True)
class FlatTerrain:
    """``FlatTerrain`` (``src/jaxsim/terrain/terrain.py:65-124``): constant height, normal +z."""

    _height: float = 0.0

    @staticmethod
    def build(height: float = 0.0) -> "FlatTerrain":
        return FlatTerrain(_height=float(height))

    def height(self, x, y):
        return np.full(np.shape(x), self._height, dtype=float)

    def normal(self, x, y):
        n = np.zeros(np.shape(x) + (3,), dtype=float)
        n[..., 2] = 1.0
        return n


@dataclasses.dataclass(frozen=True)
class PlaneTerrain(FlatTerrain):
    """``PlaneTerrain`` (``src/jaxsim/terrain/terrain.py:127-238``): plane ``A x + B y + C z + D = 0``
    with unit normal ``(A, B, C)`` and height ``-D / C`` over the origin."""

    _normal: tuple = (0.0, 0.0, 1.0)

    @staticmethod
    def build(height: float = 0.0, *, normal) -> "PlaneTerrain":
        n = np.asarray(normal, dtype=float)
        if n.shape != (3,):
            raise ValueError(f"Expected a 3D vector for the plane normal, got '{n.shape}'.")
        n = n / np.linalg.norm(n)
        return PlaneTerrain(_height=float(height), _normal=tuple(n.tolist()))

    def normal(self, x=None, y=None):
        shape = np.shape(x) if x is not None else ()
        return np.broadcast_to(np.array(self._normal, dtype=float), shape + (3,)).copy()

    def height(self, x, y):
        A, B, Cc = self._normal
        if np.allclose(Cc, 0.0):
            raise ValueError("The z component of the normal cannot be zero.")
        D = -Cc * self._height
        return np.asarray(-(A * np.asarray(x) + B * np.asarray(y) + D) / Cc, dtype=float)


@dataclasses.dataclass(frozen=True, eq=False)
class HeightFieldTerrain:
    """[round 6] The reference's generic ``Terrain`` (``src/jaxsim/terrain/terrain.py:15-62``: any ``height(x, y)``, the
    normal by central differences with ``delta = 0.010``) in the form the C-ABI can carry: heights sampled on a regular
    grid -- ``heights[ix, iy]`` at ``origin + (ix dx, iy dy)`` -- with ``height(x, y)`` the BILINEAR interpolant (clamped
    to the border samples outside the grid) and ``normal(x, y)`` the reference's central difference of that function.
    A user of the reference who subclasses ``Terrain`` samples their ``height`` onto a grid fine enough for their
    terrain; between samples the two agree to the interpolation error, at the samples exactly."""

    _heights: np.ndarray = None
    _origin: tuple = (0.0, 0.0)
    _spacing: tuple = (1.0, 1.0)
    delta: float = 0.010  # Terrain.delta (terrain.py:23)

    @staticmethod
    def build(heights, *, origin=(0.0, 0.0), spacing=(1.0, 1.0), delta: float = 0.010) -> "HeightFieldTerrain":
        h = np.array(heights, dtype=np.float64)
        if h.ndim != 2 or min(h.shape) < 2:
            raise ValueError(f"Expected a 2D grid of at least 2 x 2 heights, got '{h.shape}'.")
        if not np.all(np.isfinite(h)):
            raise ValueError("The height field contains non-finite values.")
        sp = tuple(float(v) for v in np.broadcast_to(np.asarray(spacing, dtype=float), (2,)))
        if not (sp[0] > 0 and sp[1] > 0):
            raise ValueError("The grid spacing must be positive.")
        h.setflags(write=False)  # (frozen like every other model constant: a changed terrain is a new object)
        return HeightFieldTerrain(_heights=h, _origin=tuple(float(v) for v in origin), _spacing=sp, delta=float(delta))

    @staticmethod
    def from_function(height_fn, *, x_range, y_range, spacing, delta: float = 0.010) -> "HeightFieldTerrain":
        """Sample ``height_fn(x, y)`` (vectorised over NumPy arrays) on the grid covering the two ranges."""
        sp = tuple(float(v) for v in np.broadcast_to(np.asarray(spacing, dtype=float), (2,)))
        nx = int(np.ceil((x_range[1] - x_range[0]) / sp[0])) + 1
        ny = int(np.ceil((y_range[1] - y_range[0]) / sp[1])) + 1
        X, Y = np.meshgrid(x_range[0] + sp[0] * np.arange(nx), y_range[0] + sp[1] * np.arange(ny), indexing="ij")
        return HeightFieldTerrain.build(height_fn(X, Y), origin=(x_range[0], y_range[0]), spacing=sp, delta=delta)

    @property
    def _height(self) -> float:  # (what model-level code reads of a flat terrain: not used by the device for a grid)
        return 0.0

    def height(self, x, y):
        h = self._heights
        nx, ny = h.shape
        fx = np.clip((np.asarray(x, dtype=float) - self._origin[0]) / self._spacing[0], 0.0, nx - 1)
        fy = np.clip((np.asarray(y, dtype=float) - self._origin[1]) / self._spacing[1], 0.0, ny - 1)
        ix = np.minimum(fx, nx - 2).astype(int)
        iy = np.minimum(fy, ny - 2).astype(int)
        tx, ty = fx - ix, fy - iy
        a = h[ix, iy] + (h[ix, iy + 1] - h[ix, iy]) * ty
        c = h[ix + 1, iy] + (h[ix + 1, iy + 1] - h[ix + 1, iy]) * ty
        return a + (c - a) * tx

    def normal(self, x, y):
        x, y, d = np.asarray(x, dtype=float), np.asarray(y, dtype=float), self.delta
        n = np.stack([(self.height(x - d, y) - self.height(x + d, y)) / (2 * d),
                      (self.height(x, y - d) - self.height(x, y + d)) / (2 * d), np.ones(np.shape(x))], axis=-1)  # fmt: skip
        return n / np.linalg.norm(n, axis=-1, keepdims=True)


class SoftContacts:
    """Marker for the Hunt-Crossley soft-contact model (``rbda/contacts/soft.py:126-444``)."""

    _parameters_class = SoftContactsParams

    @classmethod
    def build(cls, **kwargs):
        return cls()


@dataclasses.dataclass(frozen=True)
class RigidContactsParams:
    """``RigidContactsParams`` (``src/jaxsim/rbda/contacts/rigid.py:28-98``): friction coefficient
    and the Baumgarte gains of the contact constraint (both 0 by default)."""

    mu: float = 0.5
    K: float = 0.0
    D: float = 0.0

    @classmethod
    def build(cls, *, mu=None, K=None, D=None, **kwargs):
        return cls(mu=0.5 if mu is None else float(mu), K=0.0 if K is None else float(K), D=0.0 if D is None else float(D))

    def valid(self) -> bool:
        return self.mu >= 0.0 and self.K >= 0.0 and self.D >= 0.0


@dataclasses.dataclass(frozen=True)
class RigidContacts:
    """``RigidContacts`` (``src/jaxsim/rbda/contacts/rigid.py:95-174``): contact forces from a QP
    on the Delassus matrix, velocity reset at impacts.  ``solver_options`` accepts ``solver_tol``
    (the reference forwards the dict to ``qpax.solve_qp``)."""

    regularization_delassus: float = 1e-6
    solver_tol: float = 1e-3
    _parameters_class = RigidContactsParams

    @classmethod
    def build(cls, regularization_delassus=None, solver_options=None, **kwargs):
        opts = {"solver_tol": 1e-3} | (dict(solver_options) if solver_options is not None else {})
        unknown = set(opts) - {"solver_tol"}
        if unknown:
            raise ValueError(f"unsupported solver options: {sorted(unknown)}")
        return cls(
            regularization_delassus=1e-6 if regularization_delassus is None else float(regularization_delassus),
            solver_tol=float(opts["solver_tol"]),
        )

    @property
    def solver_options(self) -> dict:
        return {"solver_tol": self.solver_tol}


@dataclasses.dataclass(frozen=True)
class RelaxedRigidContactsParams:
    """``RelaxedRigidContactsParams`` (``src/jaxsim/rbda/contacts/relaxed_rigid.py:29-75``).  ``K`` and
    ``D`` are accepted and stored like in the reference, where they do not influence the forces
    (``_regularizers`` recomputes both from the time constant, ``:567-568``)."""

    time_constant: float = 0.02
    damping_coefficient: float = 1.0
    d_min: float = 0.9
    d_max: float = 0.95
    width: float = 0.001
    midpoint: float = 0.5
    power: float = 2.0
    K: float = 0.0
    D: float = 0.0
    mu: float = 0.005

    @classmethod
    def build(cls, **kwargs):
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: float(v) for k, v in kwargs.items() if k in names and v is not None})

    def valid(self) -> bool:
        return bool(
            self.time_constant >= 0.0 and self.damping_coefficient > 0.0 and self.d_min >= 0.0
            and self.d_max <= 1.0 and self.d_min <= self.d_max and self.width >= 0.0
            and self.midpoint >= 0.0 and self.power >= 0.0 and self.mu >= 0.0
        )  # fmt: skip


@dataclasses.dataclass(frozen=True)
class RelaxedRigidContacts:
    """``RelaxedRigidContacts`` (``src/jaxsim/rbda/contacts/relaxed_rigid.py:203-488``): contact forces
    from the regularised linear system ``(J M^-1 J^T + R) f = a_ref - a_free``, no velocity reset.
    ``solver_options`` keeps the reference's L-BFGS keys (``tol``, ``maxiter``, ``memory_size``,
    ``scale_init_precond``); the device solves the system the reference hands to
    ``custom_linear_solve`` directly, so they select nothing (HISTORY.md section 4e)."""

    _solver_options_keys: tuple = ("tol", "maxiter", "memory_size", "scale_init_precond")
    _solver_options_values: tuple = (1e-6, 50, 10, False)
    _parameters_class = RelaxedRigidContactsParams

    @classmethod
    def build(cls, solver_options=None, **kwargs):
        opts = dict(zip(cls._solver_options_keys, cls._solver_options_values)) | (
            dict(solver_options) if solver_options is not None else {}
        )
        try:
            hash(tuple(opts.values()))
        except TypeError as exc:
            raise ValueError("The values of the solver options must be hashable.") from exc
        return cls(_solver_options_keys=tuple(opts.keys()), _solver_options_values=tuple(opts.values()))

    @property
    def solver_options(self) -> dict:
        return dict(zip(self._solver_options_keys, self._solver_options_values))


class IntegratorType(enum.IntEnum):
    """``IntegratorType`` (``src/jaxsim/api/model.py:32-40``); values are the C-ABI enum.
    ``RungeKutta4Fast`` (``api/integrators.py:170-276``: contact forces and position derivatives frozen
    at the initial state) is built for RigidContacts / RelaxedRigidContacts, where the reference's version
    is well defined (HISTORY.md section 4c)."""

    SemiImplicitEuler = 0
    RungeKutta4 = 1
    RungeKutta4Fast = 2


class JaxSimModel:
    """Immutable-by-convention model container (``src/jaxsim/api/model.py:46-90``)."""

    def __init__(
        self,
        model_name: str,
        kin_dyn_parameters: KinDynParameters,
        *,
        floating_base: bool,
        time_step: float = 0.001,
        gravity: float = -STANDARD_GRAVITY,
        terrain: FlatTerrain | None = None,
        contact_model=None,
        contact_params: SoftContactsParams | None = None,
        actuation_params: ActuationParams | None = None,
        integrator: IntegratorType = IntegratorType.SemiImplicitEuler,
    ):
        self.model_name = model_name
        self.kin_dyn_parameters = kin_dyn_parameters
        self._floating_base = bool(floating_base)
        self.time_step = float(time_step)
        #: signed z acceleration; ``build_from_model_description`` negates its positive
        #: argument like the reference (``api/model.py:206``).
        self.gravity = float(gravity)
        self.terrain = terrain if terrain is not None else FlatTerrain.build()
        self.contact_model = contact_model if contact_model is not None else SoftContacts.build()
        #: default parameters follow the contact model (``api/model.py:283-286``)
        self.contact_params = contact_params if contact_params is not None else self.contact_model._parameters_class()
        self.actuation_params = actuation_params if actuation_params is not None else ActuationParams()
        self.integrator = integrator
        self._device = {}  # dtype name -> device handle (see _lib.DeviceModel)

    # -- construction -----------------------------------------------------------------------
    @staticmethod
    def build_from_model_description(
        model_description: str | pathlib.Path,
        *,
        model_name: str | None = None,
        time_step: float | None = None,
        terrain: FlatTerrain | None = None,
        contact_model=None,
        contact_params: SoftContactsParams | None = None,
        actuation_params: ActuationParams | None = None,
        integrator: IntegratorType | None = None,
        gravity: float = STANDARD_GRAVITY,
        considered_joints=None,
        locked_joint_positions: dict | None = None,
    ) -> "JaxSimModel":
        """Build from a URDF path or string (``src/jaxsim/api/model.py:128-223``).

        ``gravity`` is the positive magnitude; the stored value is ``-gravity``.
        ``considered_joints`` keeps only the listed joints, the others are locked (at
        ``locked_joint_positions``, default 0) and their links lumped (``:807-878``).
        """
        desc = urdf_parser.parse_urdf(
            str(model_description), considered_joints=considered_joints, locked_joint_positions=locked_joint_positions
        )
        kdp = KinDynParameters.build(desc)
        model = JaxSimModel(
            model_name or desc.name,
            kdp,
            floating_base=not desc.fixed_base,
            time_step=0.001 if time_step is None else time_step,
            gravity=-float(gravity),
            terrain=terrain,
            contact_model=contact_model,
            contact_params=contact_params,
            actuation_params=actuation_params,
            integrator=IntegratorType.SemiImplicitEuler if integrator is None else integrator,
        )
        model.__dict__["built_from"] = str(model_description)  # origin, for `js.model.reduce`
        return model

    # -- reference accessors (``src/jaxsim/api/model.py:674-799``) ---------------------------
    def name(self) -> str:
        return self.model_name

    def number_of_links(self) -> int:
        return self.kin_dyn_parameters.number_of_links()

    def number_of_joints(self) -> int:
        return self.kin_dyn_parameters.number_of_joints()

    def dofs(self) -> int:
        return self.kin_dyn_parameters.number_of_joints()

    def floating_base(self) -> bool:
        return self._floating_base

    def base_link(self) -> str:
        return self.kin_dyn_parameters.link_names[0]

    def link_names(self) -> tuple[str, ...]:
        return self.kin_dyn_parameters.link_names

    def joint_names(self) -> tuple[str, ...]:
        return self.kin_dyn_parameters.joint_names

    def frame_names(self) -> tuple[str, ...]:
        return self.kin_dyn_parameters.frame_names

    def total_mass(self) -> float:
        return float(np.sum(self.kin_dyn_parameters.link_mass))

    @contextlib.contextmanager
    def editable(self, validate: bool = True):
        """The ``with model.editable(validate=False) as model:`` idiom
        (``src/jaxsim/utils/jaxsim_dataclass.py:28-49``): yields a shallow copy whose
        fields may be reassigned; device copies are rebuilt on next use."""
        clone = copy.copy(self)
        clone._device = {}
        yield clone

    def _invalidate_device(self) -> None:
        self._device = {}

    def __setattr__(self, key, value):
        # Any change of a model-level constant invalidates the cached device tables.
        if key != "_device" and "_device" in self.__dict__:
            self.__dict__["_device"] = {}
        super().__setattr__(key, value)
