"""Contour types and quadrature rules (host side).  Mirrors src/contour.jl of
the reference: only ``nodes`` / ``weights`` and the region parameters cross the
C ABI (feasthip_set_contour)."""
from __future__ import annotations

import numpy as np


class Contour:
    """abstract type Contour (src/contour.jl:1)"""
    nodes: np.ndarray
    weights: np.ndarray


class CircularContour(Contour):
    """src/contour.jl:3-8"""

    def __init__(self, c, r, nodes, weights):
        self.c = complex(c)
        self.r = float(r)
        self.nodes = np.ascontiguousarray(nodes, dtype=np.complex128)
        self.weights = np.ascontiguousarray(weights, dtype=np.complex128)

    def region(self):
        return 0, np.array([self.c.real, self.c.imag, self.r, 0.0])


class RectangularContour(Contour):
    """src/contour.jl:10-16; the constructor rejects unordered corners."""

    def __init__(self, bottom_left, top_right, nodes, weights):
        bl, tr = complex(bottom_left), complex(top_right)
        if not (bl.real < tr.real and bl.imag < tr.imag):
            raise ValueError("Invalid corners")
        self.bottom_left, self.top_right = bl, tr
        self.nodes = np.ascontiguousarray(nodes, dtype=np.complex128)
        self.weights = np.ascontiguousarray(weights, dtype=np.complex128)

    def region(self):
        return 1, np.array([self.bottom_left.real, self.bottom_left.imag, self.top_right.real, self.top_right.imag])


def circular_contour_trapezoidal(c, r, N=16):
    """src/contour.jl:26-31: theta_k = pi/N + (k-1) 2pi/N, z_k = c + r e^{i theta_k}, w_k = r e^{i theta_k}/N."""
    theta = np.linspace(np.pi / N, 2 * np.pi - np.pi / N, N)      # LinRange(pi/N, 2pi - pi/N, N)
    e = r * np.exp(1j * theta)
    return CircularContour(c, r, e + c, e / N)


def circular_contour_gauss(c, r, N=16):
    """src/contour.jl:33-45"""
    if N % 2:
        raise ValueError("Number of nodes must be multiple of 2")
    h = N // 2
    x, w = np.polynomial.legendre.leggauss(h)
    t = 0.5 * np.pi * (x + 1.0)
    tt = np.concatenate([t, t + np.pi])
    ww = np.concatenate([w, w])
    e = r * np.exp(1j * tt)
    return CircularContour(c, r, e + c, e * ww / 4.0)


def _rect_setup(bl, tr, N):
    bl, tr = complex(bl), complex(tr)
    if N % 4:
        raise ValueError("Number of nodes must be multiple of 4")
    return bl, tr, N // 4, tr.real - bl.real, tr.imag - bl.imag


def rectangular_contour_gauss(bottom_left, top_right, N=16):
    """src/contour.jl:47-65: Gauss-Legendre on each side, clockwise top/right/bottom/left."""
    bl, tr, q, wlen, hlen = _rect_setup(bottom_left, top_right, N)
    x, w = np.polynomial.legendre.leggauss(q)
    fwd, rev = x + 1.0, x[::-1] + 1.0
    sides = [
        (fwd * (wlen / 2) + complex(bl.real, tr.imag), w * wlen),
        (fwd * (1j * hlen / 2) + complex(tr.real, bl.imag), -1j * w * hlen),
        (rev * (wlen / 2) + bl, -w * wlen),
        (rev * (1j * hlen / 2) + bl, 1j * w * hlen),
    ]
    nodes = np.concatenate([s[0] for s in sides])
    weights = np.concatenate([s[1] for s in sides]) / (-4.0 * np.pi * 1j)
    return RectangularContour(bl, tr, nodes, weights)


def rectangular_contour_trapezoidal(bottom_left, top_right, N=16):
    """src/contour.jl:67-86: trapezoid rule, corner weights are the half sums of the two sides."""
    bl, tr, q, wlen, hlen = _rect_setup(bottom_left, top_right, N)
    f = np.arange(q) / q
    nodes = np.concatenate([
        bl.real + f * wlen + 1j * tr.imag,
        tr.real + 1j * (tr.imag - f * hlen),
        tr.real - f * wlen + 1j * bl.imag,
        bl.real + 1j * (bl.imag + f * hlen),
    ])
    steps = [wlen / q, -1j * hlen / q, -wlen / q, 1j * hlen / q]   # dz along top, right, bottom, left
    weights = np.zeros(N, dtype=np.complex128)
    for s in range(4):
        weights[s * q + 1:(s + 1) * q] = steps[s]
        weights[s * q] = 0.5 * (steps[s] + steps[s - 1])
    return RectangularContour(bl, tr, nodes, weights / (-2.0 * np.pi * 1j))


def in_contour(lam, contour, r=None):
    """src/contour.jl:88-100: closed disc, open rectangle."""
    lam = np.asarray(lam)
    if r is not None:
        return np.abs(lam - contour) <= r
    if isinstance(contour, CircularContour):
        return np.abs(lam - contour.c) <= contour.r
    bl, tr = contour.bottom_left, contour.top_right
    return (bl.real < lam.real) & (lam.real < tr.real) & (bl.imag < lam.imag) & (lam.imag < tr.imag)
