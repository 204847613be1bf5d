"""Composite wind-extrusion operators (host float64 precompute for ``aog_upload_layer_composite``).

hcipy's ``InfiniteAtmosphericLayer.evolve_until`` (AO_env.py:125) shifts the screen one pixel at a time: every shift draws a new
row / column ``new = A z + sqrt(Cn^2) B n`` from a stencil ``z`` of the CURRENT screen — which, from the second shift of a step on,
contains samples the step itself has just created.  k successive shifts along one axis are nevertheless ONE linear map of the screen as
it stood before the first of them and of the k N normals they draw:

    [R_1; ...; R_k] = A_k z_old + sqrt(Cn^2) B_k [n_1; ...; n_k]          (R_j = the slice shift j creates, j = 1 first)

with ``A_k`` [k N, U_k] over the UNION of old samples any of the k stencils reaches and ``B_k`` [k N, k N] block lower triangular.  The
composition is exact (no approximation: the same samples given the same normals, up to float64 rounding of the matrix products here), it
removes the k-fold sequential dependence — all k N new values of an env come out of one matrix product, whose rows are independent —
and it is what the device's int8 matrix-core extrusion applies (``csrc/k_extrude_i8.h``).

Coordinates: "slice" = the axis that is extruded (rows for the vertical stencil, columns for the horizontal one), "along" = the other.
Stencil samples are logical indices on the screen as hcipy's ``_extrude`` sees it (for the 'top' / 'right' directions that is the
180-degree rotated screen; the device gathers accordingly), new slices are prepended at slice 0.
"""
from __future__ import annotations

import numpy as np


def compose_extrusions(stencil, A, B, n: int, k: int, vertical: bool):
    """k one-pixel extrusions along one axis as one operator.

    stencil  [nz] flat logical indices (sy * n + sx) of the single-shift stencil (``build_layer_tables``)
    A, B     [n, nz], [n, n] of the single shift (B without the sqrt(Cn^2) factor)
    Returns (old_yx [U] int32 = (sy << 16 | sx) on the screen BEFORE the first shift, A_k [k n, U], B_k [k n, k n]);
    row (j - 1) n + i of A_k / B_k = sample i of the slice shift j creates; column (j' - 1) n + i' of B_k = normal i' of shift j'."""
    stencil = np.asarray(stencil, dtype=np.int64)
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    nz = stencil.size
    sy, sx = stencil // n, stencil % n
    sl, al = (sy, sx) if vertical else (sx, sy)          # (slice, along) of every stencil sample
    # union of the old samples: shift j sees current slice c as old slice c - (j - 1) when c >= j - 1
    old = set()
    for j in range(1, k + 1):
        for m in range(nz):
            c = int(sl[m]) - (j - 1)
            if c >= 0:
                old.add((c, int(al[m])))
    old = sorted(old)
    index = {p: i for i, p in enumerate(old)}
    U = len(old)
    Ak = np.zeros((k * n, U))
    Bk = np.zeros((k * n, k * n))
    for j in range(1, k + 1):
        # S [nz, U + k n]: every stencil sample of shift j as a combination of old samples and normals
        S = np.zeros((nz, U + k * n))
        for m in range(nz):
            c = int(sl[m])
            if c >= j - 1:
                S[m, index[(c - (j - 1), int(al[m]))]] = 1.0
            else:                                        # the slice created by shift j - 1 - c, its sample al[m]
                r = (j - 1 - c - 1) * n + int(al[m])
                S[m, :U] = Ak[r]
                S[m, U:] = Bk[r]
        C = A @ S
        rows = slice((j - 1) * n, j * n)
        Ak[rows] = C[:, :U]
        Bk[rows] = C[:, U:]
        Bk[rows, (j - 1) * n:j * n] += B
    if vertical:
        yx = np.array([(s << 16) | a for s, a in old], dtype=np.int32)
    else:
        yx = np.array([(a << 16) | s for s, a in old], dtype=np.int32)
    return yx, Ak, Bk


def apply_composite(screen, yx, Ak, Bk, normals, sqrt_cn2: float, vertical: bool, flipped: bool):
    """Float64 host application of one composite operator to one logical [n, n] screen (test / documentation aid; the device does this on
    the int8 matrix cores).  ``normals`` [k, n].  Returns the screen after the k shifts, exactly as k calls of hcipy's ``_extrude`` in
    direction 'bottom' / 'left' (``flipped`` False) or 'top' / 'right' (True: they act on the 180-degree rotated screen) would leave it."""
    n = screen.shape[0]
    k = Ak.shape[0] // n
    s = screen[::-1, ::-1] if flipped else screen
    z = s[yx >> 16, yx & 0xFFFF]
    new = (Ak @ z + sqrt_cn2 * (Bk @ np.asarray(normals, dtype=np.float64).reshape(-1))).reshape(k, n)
    if vertical:
        out = np.vstack([new[::-1], s[:n - k]])          # slice j ends up at row k - j
    else:
        out = np.hstack([new[::-1].T, s[:, :n - k]])
    return out[::-1, ::-1] if flipped else out
