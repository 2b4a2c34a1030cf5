#!/usr/bin/env python3
"""Analytic fixtures that do NOT come from the oracle: closed-form equations of motion, derived by hand
from the Lagrangian of each system and evaluated with plain NumPy.  Nothing of ``oracle/`` or of the
reference is imported.  The fixtures pin the oracle (tests/test_oracle_independent.py), the kernel core in
emulation and the HIP kernels (``-m gpu``) against an independent statement of the physics.

    python tests/golden/make_analytic.py     -> tests/golden/analytic_{cartpole,serial_double_pendulum,two_pendulums}.npz

Systems (constants are restated here on purpose -- a parser or table bug must show up as a mismatch):

* cartpole (reference example ``examples/assets/cartpole.urdf``): cart M on a prismatic joint along +y,
  pole (m, CoM at l along its +z, inertia I about x) on a revolute joint about +x.
      y_p = y - l sin(th),  z_p = l cos(th)
      T = 1/2 (M+m) yd^2 - m l cos(th) yd thd + 1/2 (I + m l^2) thd^2,   V = m g l cos(th)
      (M+m) ydd - m l cos(th) thdd + m l sin(th) thd^2 = F
      (I + m l^2) thdd - m l cos(th) ydd - m g l sin(th) = tau
* serial double pendulum (``robots.serial_double_pendulum_urdf``), angles about +x, links along local +z,
  th2 relative to link 1:
      M11 = I1 + I2 + m1 c1^2 + m2 (L1^2 + c2^2 + 2 L1 c2 cos th2)
      M12 = I2 + m2 (c2^2 + L1 c2 cos th2),  M22 = I2 + m2 c2^2,  h = m2 L1 c2 sin th2
      M11 thdd1 + M12 thdd2 - h (2 thd1 thd2 + thd2^2) - (m1 c1 + m2 L1) g sin th1 - m2 g c2 sin(th1+th2) = tau1
      M12 thdd1 + M22 thdd2 + h thd1^2 - m2 g c2 sin(th1+th2) = tau2
* two independent pendulums on one base (the reference's ``tests/assets/double_pendulum.sdf``): joint
  frames rolled by r = -3.1415 about x, CoM at l = 0.5 along the rolled +z, I = 1 about the axis:
      (I + m l^2) thdd - m g l sin(th + r) = tau
For `step` fixtures the joint torque is tau_ref - kv sd - kc sign(sd) (api/actuation_model.py:75-89 as
physics: viscous + Coulomb friction) and the integrator is semi-implicit Euler: sd+ = sd + dt sdd,
s+ = s + dt sd+.
"""
import pathlib

import numpy as np

G = 9.81
HERE = pathlib.Path(__file__).resolve().parent

CARTPOLE = dict(M=1.0, m=0.5, l=0.5, I=0.04166979166666667)
SERIAL = dict(m1=1.3, m2=0.7, L1=0.45, c1=0.2, c2=0.3, I1=0.021, I2=0.013)
TWO = dict(m=1.0, l=0.5, I=1.0, roll=-3.1415, kv=1.0)


def cartpole_acc(s, sd, tau, M, m, l, I):
    y, th = s[:, 0], s[:, 1]
    yd, thd = sd[:, 0], sd[:, 1]
    a11, a12, a22 = M + m, -m * l * np.cos(th), I + m * l * l
    b1 = tau[:, 0] - m * l * np.sin(th) * thd**2
    b2 = tau[:, 1] + m * G * l * np.sin(th)
    det = a11 * a22 - a12 * a12
    return np.stack([(a22 * b1 - a12 * b2) / det, (a11 * b2 - a12 * b1) / det], axis=1)


def serial_acc(s, sd, tau, m1, m2, L1, c1, c2, I1, I2):
    t1, t2 = s[:, 0], s[:, 1]
    d1, d2 = sd[:, 0], sd[:, 1]
    M11 = I1 + I2 + m1 * c1**2 + m2 * (L1**2 + c2**2 + 2 * L1 * c2 * np.cos(t2))
    M12 = I2 + m2 * (c2**2 + L1 * c2 * np.cos(t2))
    M22 = I2 + m2 * c2**2
    h = m2 * L1 * c2 * np.sin(t2)
    b1 = tau[:, 0] + h * (2 * d1 * d2 + d2**2) + (m1 * c1 + m2 * L1) * G * np.sin(t1) + m2 * G * c2 * np.sin(t1 + t2)
    b2 = tau[:, 1] - h * d1**2 + m2 * G * c2 * np.sin(t1 + t2)
    det = M11 * M22 - M12 * M12
    return np.stack([(M22 * b1 - M12 * b2) / det, (M11 * b2 - M12 * b1) / det], axis=1)


def two_pendulums_acc(s, sd, tau, m, l, I, roll, kv):
    return (tau + m * G * l * np.sin(s + roll)) / (I + m * l * l)


def euler(s, sd, acc, dt):
    sd1 = sd + dt * acc
    return s + dt * sd1, sd1


def main():
    rng = np.random.default_rng(20260927)
    N, dt = 64, 1e-3
    for name, fn, par, kv, kc in (
        ("cartpole", cartpole_acc, CARTPOLE, 0.0, 0.0),
        ("serial_double_pendulum", serial_acc, SERIAL, 0.3, 0.05),
        ("two_pendulums", two_pendulums_acc, TWO, 1.0, 0.0),
    ):
        s = rng.uniform(-1.5, 1.5, (N, 2))
        sd = rng.uniform(-2.0, 2.0, (N, 2))
        tau = rng.uniform(-3.0, 3.0, (N, 2))
        # forward dynamics: the joint forces are applied as given (no friction: forward_dynamics_aba)
        sdd = fn(s, sd, tau, **par)
        # one step: friction enters through the actuation model
        tau_step = tau - kv * sd - kc * np.sign(sd)
        s1, sd1 = euler(s, sd, fn(s, sd, tau_step, **par), dt)
        np.savez(HERE / f"analytic_{name}.npz", s=s, sd=sd, tau=tau, sdd=sdd, s_next=s1, sd_next=sd1, dt=dt,
                 kv=kv, kc=kc, **{f"par_{k}": v for k, v in par.items()})
        print(name, "ok", float(np.abs(sdd).max()))


if __name__ == "__main__":
    main()
