"""elast_quad4 restated from first principles: 4-node bilinear quad, plane stress, 2x2 Gauss.

dof order [u1x,u1y,...,u4x,u4y]; nodes counter-clockwise; shape functions
N = 1/4 [(1-r)(1-s), (1+r)(1-s), (1+r)(1+s), (1-r)(1+s)].
"""
import numpy as np


def elast_quad4(coord, params):
    E, nu = float(params[0]), float(params[1])
    C = E / (1.0 - nu ** 2) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, (1.0 - nu) / 2.0]])
    g = 1.0 / np.sqrt(3.0)
    k = np.zeros((8, 8))
    m = np.zeros((8, 8))
    coord = np.asarray(coord, dtype=float)
    for r in (-g, g):
        for s in (-g, g):
            dN = 0.25 * np.array([[-(1 - s), (1 - s), (1 + s), -(1 + s)],
                                  [-(1 - r), -(1 + r), (1 + r), (1 - r)]])
            J = dN @ coord
            det = np.linalg.det(J)
            dNdx = np.linalg.solve(J, dN)
            B = np.zeros((3, 8))
            B[0, 0::2] = dNdx[0]
            B[1, 1::2] = dNdx[1]
            B[2, 0::2] = dNdx[1]
            B[2, 1::2] = dNdx[0]
            k += det * (B.T @ C @ B)
    return k, m
