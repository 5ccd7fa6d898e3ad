"""Small numpy SE(3) toolbox in the conventions the reference's callers use.

* pose storage = Sophus ``SE3d::data()`` order ``[qx qy qz qw tx ty tz]`` -- the 7 numbers Ceres
  hands to ``NIDCost::operator()`` (nid_cost.hpp:38, visual_camera_calibration.cpp:215-229);
* ``plus(x, delta)`` = ``T * exp(delta)``, delta = [upsilon; omega] -- ``Sophus::Manifold<SE3>::Plus``
  (visual_camera_calibration.cpp:216);
* ``plus_jacobian(x)`` = ``Dx_this_mul_exp_x_at_0`` (7x6), which Ceres right-multiplies onto the
  ambient gradient;
* ``pose3_expmap(xi)`` = GTSAM ``Pose3::Expmap``, xi = [omega; v] (Nelder-Mead path,
  visual_camera_calibration.cpp:104,129);
* TUM order ``[tx ty tz qx qy qz qw]`` for ``calib.json`` (calibrate.cpp:72-76,128-133).
"""
import numpy as np


def quat_mul(a, b):
    """Hamilton product, storage [x y z w]."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array(
        [
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz,
        ]
    )


def quat_to_rot(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def rot_to_quat(R):
    """Rotation matrix -> unit quaternion [x y z w] (Shepperd's method)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def so3_exp_quat(omega):
    theta = np.linalg.norm(omega)
    if theta < 1e-10:
        half = 0.5 - theta * theta / 48.0
        w = 1.0 - theta * theta / 8.0
    else:
        half = np.sin(0.5 * theta) / theta
        w = np.cos(0.5 * theta)
    q = np.array([half * omega[0], half * omega[1], half * omega[2], w])
    return q / np.linalg.norm(q)


def _left_jacobian(omega):
    theta = np.linalg.norm(omega)
    W = hat(omega)
    if theta < 1e-8:
        return np.eye(3) + 0.5 * W + W @ W / 6.0
    return np.eye(3) + (1 - np.cos(theta)) / theta**2 * W + (theta - np.sin(theta)) / theta**3 * (W @ W)


def se3_exp(delta):
    """Sophus SE3::exp, delta = [upsilon(3); omega(3)] -> 7-vector."""
    ups, omega = np.asarray(delta[:3], float), np.asarray(delta[3:], float)
    q = so3_exp_quat(omega)
    t = _left_jacobian(omega) @ ups
    return np.concatenate([q, t])


def compose(a, b):
    """a * b for 7-vectors."""
    qa, ta = a[:4], a[4:]
    qb, tb = b[:4], b[4:]
    q = quat_mul(qa, qb)
    q = q / np.linalg.norm(q)
    t = quat_to_rot(qa) @ tb + ta
    return np.concatenate([q, t])


def inverse(a):
    q = np.array([-a[0], -a[1], -a[2], a[3]])
    t = -(quat_to_rot(q) @ a[4:])
    return np.concatenate([q, t])


def plus(x, delta):
    """Sophus::Manifold<SE3>::Plus: x * exp(delta)."""
    return compose(np.asarray(x, float), se3_exp(delta))


def plus_jacobian(x):
    """Sophus ``SE3::Dx_this_mul_exp_x_at_0`` (7x6): d(x * exp(delta))/d(delta) at 0, rows in
    storage order [qx qy qz qw tx ty tz], columns [upsilon; omega]."""
    qx, qy, qz, qw = x[:4]
    J = np.zeros((7, 6))
    J[0:4, 3:6] = 0.5 * np.array([[qw, -qz, qy], [qz, qw, -qx], [-qy, qx, qw], [-qx, -qy, -qz]])
    J[4:7, 0:3] = quat_to_rot(x[:4])
    return J


def to_matrix(x):
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(x[:4])
    T[:3, 3] = x[4:]
    return T


def from_matrix(T):
    return np.concatenate([rot_to_quat(T[:3, :3]), T[:3, 3]])


def pose3_expmap(xi):
    """GTSAM Pose3::Expmap, xi = [omega(3); v(3)] -> 4x4 matrix (full SE(3) exponential)."""
    omega, v = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(so3_exp_quat(omega))
    T[:3, 3] = _left_jacobian(omega) @ v
    return T


def from_tum(v):
    """calib.json order [tx ty tz qx qy qz qw] -> 7-vector (quaternion normalised as
    calibrate.cpp:74 does)."""
    v = np.asarray(v, float)
    q = v[3:7] / np.linalg.norm(v[3:7])
    return np.concatenate([q, v[0:3]])


def to_tum(x):
    return np.concatenate([x[4:7], x[0:4]])


def delta_trans_rot(a, b):
    """|translation| and rotation angle of a^-1 * b (the convergence / parity measures)."""
    d = compose(inverse(a), b)
    ang = 2.0 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))
    return float(np.linalg.norm(d[4:])), float(ang)
