"""Host-side camera mirror used by the Python test/bench harness.

Mirrors, in float64 like the JavaScript original:
  Camera           src/cameras/Camera.ts:22-92   (update -> projection/view/viewProj)
  rotation_from_quaternion   src/math/Matrix3.ts:67-80
  quaternion_from_euler      src/math/Quaternion.ts:65-83
  mat4_multiply    src/math/Matrix4.ts:32-53
  orbit_pose       src/controls/OrbitControls.ts:275-283 (the pose formula only)

Matrix buffers are flat 16-element lists laid out exactly like Matrix4.buffer
(uploaded with transpose=false, i.e. buffer[c*4+r] = element(row r, col c)).
The float32 arrays handed to the device are produced the way the reference
does it: `new Float32Array(m.buffer)` (Worker.ts:37, WebGLRenderer.ts:147,159).
"""
import math

import numpy as np


def rotation_from_quaternion(x, y, z, w):
    return [
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
        2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
        2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y,
    ]


def quaternion_from_euler(ex, ey, ez):
    hx, hy, hz = ex / 2, ey / 2, ez / 2
    cy, sy = math.cos(hy), math.sin(hy)
    cp, sp = math.cos(hx), math.sin(hx)
    cz, sz = math.cos(hz), math.sin(hz)
    return (
        cy * sp * cz + sy * cp * sz,
        sy * cp * cz - cy * sp * sz,
        cy * cp * sz - sy * sp * cz,
        cy * cp * cz + sy * sp * sz,
    )


def mat4_multiply(a, b):
    """Matrix4.multiply: this=a, m=b."""
    r = []
    for i in range(4):
        for j in range(4):
            r.append(b[4 * i + 0] * a[j] + b[4 * i + 1] * a[4 + j] + b[4 * i + 2] * a[8 + j] + b[4 * i + 3] * a[12 + j])
    return r


class Camera:
    def __init__(self, position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), fx=1132.0, fy=1132.0, near=0.01, far=1000.0):
        self.position = tuple(position)
        self.rotation = tuple(rotation)  # x, y, z, w
        self.fx, self.fy, self.near, self.far = fx, fy, near, far
        self.projectionMatrix = self.viewMatrix = self.viewProj = None

    def update(self, width, height):
        fx, fy, near, far = self.fx, self.fy, self.near, self.far
        self.projectionMatrix = [
            2 * fx / width, 0, 0, 0,
            0, -2 * fy / height, 0, 0,
            0, 0, far / (far - near), 1,
            0, 0, -(far * near) / (far - near), 0,
        ]
        R = rotation_from_quaternion(*self.rotation)
        t = self.position
        self.viewMatrix = [
            R[0], R[1], R[2], 0,
            R[3], R[4], R[5], 0,
            R[6], R[7], R[8], 0,
            -t[0] * R[0] - t[1] * R[3] - t[2] * R[6],
            -t[0] * R[1] - t[1] * R[4] - t[2] * R[7],
            -t[0] * R[2] - t[1] * R[5] - t[2] * R[8],
            1,
        ]
        self.viewProj = mat4_multiply(self.projectionMatrix, self.viewMatrix)
        return self

    def f32(self):
        """(view, projection, viewProj) as float32[16], rounded like Float32Array(buffer)."""
        return (np.asarray(self.viewMatrix, dtype=np.float32),
                np.asarray(self.projectionMatrix, dtype=np.float32),
                np.asarray(self.viewProj, dtype=np.float32))


def orbit_pose(alpha, beta=0.3, radius=8.0, target=(0.0, 0.0, 0.0)):
    """Position + rotation quaternion of OrbitControls.update for (alpha, beta, radius, target)."""
    x = target[0] + radius * math.sin(alpha) * math.cos(beta)
    y = target[1] - radius * math.sin(beta)
    z = target[2] - radius * math.cos(alpha) * math.cos(beta)
    dx, dy, dz = target[0] - x, target[1] - y, target[2] - z
    ln = math.sqrt(dx * dx + dy * dy + dz * dz)
    dx, dy, dz = dx / ln, dy / ln, dz / ln
    rx = math.asin(-dy)
    ry = math.atan2(dx, dz)
    return (x, y, z), quaternion_from_euler(rx, ry, 0.0)


def orbit_camera(k, frames=120, width=1920, height=1080, fx=1132.0, fy=None, beta=0.3, radius=8.0):
    """Camera k of the bench's 120-frame orbit."""
    pos, rot = orbit_pose(2.0 * math.pi * k / frames, beta, radius)
    return Camera(pos, rot, fx, fx if fy is None else fy).update(width, height)
