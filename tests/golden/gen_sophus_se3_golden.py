"""Generates tests/golden/sophus_se3_golden.json.

Runs ONLY in the authoring container: imports the reference's own sympy model
of Sophus (thirdparty/Sophus/py/sophus: Se3.exp, Se3.__mul__, Se3.matrix) from
/root/reference and evaluates it at high precision on
  * the 7 tangent vectors of thirdparty/Sophus/test/core/test_se3.cpp:30-43,
  * small tracker-scale increments (|v| ~ cm, |w| ~ 0.01 rad),
  * products exp(a) * exp(b) (the `exp(inc) * referenceToFrame` of
    system/optimizer.cpp:266).
The JSON holds inputs and expected 4x4 matrices only (data, not code).
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference/thirdparty/Sophus/py")
import sympy  # noqa: E402
import sophus  # noqa: E402
from sophus.se3 import Se3  # noqa: E402


def exp_matrix(v):
    vec = sympy.Matrix(6, 1, [sympy.Float(x, 40) for x in v])
    if all(abs(x) < 1e-300 for x in v[3:]):
        # theta == 0: the sympy closed form divides by theta; use the limit
        # (V = I, R = I), which is what se3.hpp:737-739 returns for theta < eps
        T = sympy.eye(4)
        T[0, 3], T[1, 3], T[2, 3] = vec[0], vec[1], vec[2]
        return T, None
    g = Se3.exp(vec)
    return g.matrix().evalf(30), g


def to_list(M):
    return [[float(M[r, c]) for c in range(4)] for r in range(4)]


tangents = [
    [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0],
    [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0],
]
small = [
    [0.01, -0.02, 0.005, 0.01, 0.002, -0.015], [1e-3, 2e-3, -1e-3, 1e-4, -2e-4, 3e-4],
    [0.03, 0.0, 0.0, 0.0, 0.026, 0.0], [-0.004, 0.011, 0.02, -0.02, 0.01, 0.005],
    [1e-5, 1e-5, 1e-5, 1e-6, 2e-6, -1e-6],
]
cases = []
for v in tangents + small:
    M, _ = exp_matrix(v)
    cases.append({"tangent": v, "matrix": to_list(M)})
products = []
pairs = [(small[0], small[3]), (small[1], tangents[4]), (tangents[2], small[2]), (small[3], small[0])]
for a, b in pairs:
    Ma, ga = exp_matrix(a)
    Mb, gb = exp_matrix(b)
    P = (ga * gb).matrix().evalf(30)
    products.append({"a": a, "b": b, "matrix": to_list(P)})
# SO3: unit quaternion <-> rotation matrix (so3.hpp:280-282 `matrix()`, so3.hpp:419-424 `SO3(R)`), from the reference's
# sympy So3: q = So3.exp(w).q and its matrix(); the matrix -> quaternion direction is pinned as the inverse of this map
from sophus.so3 import So3  # noqa: E402
rot = []
for w in [[0.3, -0.2, 0.1], [1e-3, 2e-3, -1e-3], [3.0, 0.2, -0.1], [0.0, 3.1, 0.0], [-2.0, -2.0, 0.5], [0.01, 0.0, 0.0],
          [0.5, 2.9, 0.4], [-0.1, 0.2, 3.0], [2.2, -2.1, 0.3]]:
    g = So3.exp(sympy.Matrix(3, 1, [sympy.Float(x, 40) for x in w]))
    q = g.q
    M = g.matrix().evalf(30)
    rot.append({"omega": w, "q_wxyz": [float(q.real.evalf(30))] + [float(q.vec[i].evalf(30)) for i in range(3)],
                "matrix": [[float(M[r, c]) for c in range(3)] for r in range(3)]})
out = {"source": "thirdparty/Sophus/py/sophus (sympy), test/core/test_se3.cpp:30-43",
       "exp": cases, "mul": products, "so3": rot}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sophus_se3_golden.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(cases), "exp cases,", len(products), "products,", len(rot), "rotations")
