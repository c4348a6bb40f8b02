"""ScalarTransform4f mirror (/root/reference/include/mitsuba/core/transform.h:23-330).

A transform carries its matrix and the inverse transpose, exactly as the reference does, so no
numeric inversion happens downstream (Transform::inverse is a pair of transposes, transform.h:59-61).
All arithmetic is float32 and follows the reference's formulas.
"""
import math
import numpy as np

f32 = np.float32


def _m(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(4, 4))


def _fma32(a, b, c):
    """float32 fused multiply-add (the product of two float32 is exact in float64; one rounding to float32 at the end, up to
    the rare double rounding of the float64 sum)."""
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def normalize32(v):
    """enoki's normalize for a float32 3-vector: v * rsqrt(squared_norm(v)), squared_norm as the fma chain of dot()
    (the same convention as oracle/oracle_math.h and csrc/dmath.h)."""
    v = np.asarray(v, dtype=np.float32)
    sq = _fma32(v[2], v[2], _fma32(v[1], v[1], np.float32(v[0] * v[0])))
    r = np.float32(1.0) / np.sqrt(sq, dtype=np.float32)
    return (v * r).astype(np.float32)


class ScalarTransform4f:
    __slots__ = ("matrix", "inverse_transpose")

    def __init__(self, matrix=None, inverse_transpose=None):
        if matrix is None:
            matrix = np.eye(4, dtype=np.float32)
        if isinstance(matrix, ScalarTransform4f):
            inverse_transpose = matrix.inverse_transpose
            matrix = matrix.matrix
        self.matrix = _m(matrix)
        if inverse_transpose is None:
            # Transform(Matrix): inverse_transpose = enoki::inverse_transpose(value) (transform.h:48-50)
            inverse_transpose = np.linalg.inv(self.matrix.astype(np.float64)).T
        self.inverse_transpose = _m(inverse_transpose)

    # transform.h:53-56
    def __matmul__(self, other):
        if isinstance(other, ScalarTransform4f):
            return ScalarTransform4f((self.matrix @ other.matrix).astype(np.float32),
                                     (self.inverse_transpose @ other.inverse_transpose).astype(np.float32))
        return NotImplemented

    def __mul__(self, other):
        if isinstance(other, ScalarTransform4f):
            return self.__matmul__(other)
        v = np.asarray(other, dtype=np.float32)
        if v.shape == (3,):
            return self.transform_point(v)
        return NotImplemented

    def inverse(self):
        return ScalarTransform4f(self.inverse_transpose.T.copy(), self.matrix.T.copy())

    def transform_point(self, p):
        p = np.asarray(p, dtype=np.float32)
        r = self.matrix @ np.append(p, f32(1)).astype(np.float32)
        return (r[:3] / r[3]).astype(np.float32)

    def transform_vector(self, v):
        return (self.matrix[:3, :3] @ np.asarray(v, dtype=np.float32)).astype(np.float32)

    def transform_normal(self, n):
        return (self.inverse_transpose[:3, :3] @ np.asarray(n, dtype=np.float32)).astype(np.float32)

    def translation(self):
        return self.matrix[:3, 3].copy()

    # transform.h:160-170
    @staticmethod
    def translate(v):
        v = np.asarray(v, dtype=np.float32)
        m = np.eye(4, dtype=np.float32); m[:3, 3] = v
        it = np.eye(4, dtype=np.float32); it[3, :3] = -v
        return ScalarTransform4f(m, it)

    @staticmethod
    def scale(v):
        v = np.asarray(v, dtype=np.float32)
        if v.shape == ():
            v = np.array([v, v, v], dtype=np.float32)
        m = np.diag(np.append(v, f32(1))).astype(np.float32)
        it = np.diag(np.append(f32(1) / v, f32(1))).astype(np.float32)
        return ScalarTransform4f(m, it)

    # transform.h:173-177 (angle in degrees; rotation matrices are their own inverse transpose)
    @staticmethod
    def rotate(axis, angle):
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        t = math.radians(float(angle))
        c, s = math.cos(t), math.sin(t)
        x, y, z = a
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                      [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                      [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                      [0, 0, 0, 1]], dtype=np.float64)
        return ScalarTransform4f(r.astype(np.float32), r.astype(np.float32))

    # transform.h:241-269
    @staticmethod
    def look_at(origin, target, up):
        origin = np.asarray(origin, dtype=np.float32)
        target = np.asarray(target, dtype=np.float32)
        up = np.asarray(up, dtype=np.float32)

        normalize = normalize32
        d = normalize(target - origin)
        d = normalize(d)
        left = normalize(np.cross(up, d).astype(np.float32))
        new_up = np.cross(d, left).astype(np.float32)
        m = np.eye(4, dtype=np.float32)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, origin
        inv = np.eye(4, dtype=np.float32)
        inv[0, :3], inv[1, :3], inv[2, :3] = left, new_up, d
        inv[:, 3] = inv @ np.append(-origin, f32(1)).astype(np.float32)
        return ScalarTransform4f(m, inv.T.copy())

    @staticmethod
    def from_frame(s, t, n):
        m = np.eye(4, dtype=np.float32)
        m[0, :3], m[1, :3], m[2, :3] = s, t, n
        return ScalarTransform4f(m, m)

    def __repr__(self):
        return "ScalarTransform4f(\n%s)" % np.array2string(self.matrix)

    def __eq__(self, o):
        return isinstance(o, ScalarTransform4f) and np.array_equal(self.matrix, o.matrix) and \
            np.array_equal(self.inverse_transpose, o.inverse_transpose)


# enoki::sign / mulsign helpers and coordinate_system (vector.h:116-136) for the "direction" parameters
def coordinate_system(n):
    n = np.asarray(n, dtype=np.float32)
    sign = np.copysign(f32(1), n[2])
    a = f32(-1) / (sign + n[2])
    b = n[0] * n[1] * a

    def mulsign(x, y):
        return np.copysign(f32(1), y) * x
    s = np.array([mulsign(n[0] * n[0] * a, n[2]) + f32(1), mulsign(b, n[2]), -mulsign(n[0], n[2])], dtype=np.float32)
    t = np.array([b, sign + n[1] * n[1] * a, -n[1]], dtype=np.float32)
    return s, t
