"""Element-block operator containers (host side, NumPy).

Mirrors the reference's structured operator types (reference:
src/BlockMatrices.jl:17-62): a ``BlockDiag`` holds one dense ``p x q`` block per
element in a Fortran-ordered ``(p, q, N)`` array, so the memory image is exactly
the Julia ``Array{T,3}`` the reference stores (element blocks contiguous, row index
fastest).  That image is what the C ABI uploads to HBM unchanged.

Only data layout and the cheap host conversions live here; every matvec / triple
product / R'HR assembly on the solve path runs in the HIP library.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class BlockDiag:
    """Block-diagonal operator, ``data[r, c, e]`` (reference: src/BlockMatrices.jl:17-27)."""

    data: np.ndarray  # (p, q, N), Fortran order

    def __post_init__(self):
        d = np.asarray(self.data, dtype=np.float64)
        if d.ndim != 3:
            raise ValueError("BlockDiag data must be a (p, q, N) array")
        self.data = np.asfortranarray(d)

    @property
    def p(self) -> int:
        return self.data.shape[0]

    @property
    def q(self) -> int:
        return self.data.shape[1]

    @property
    def N(self) -> int:
        return self.data.shape[2]

    @property
    def shape(self):
        return (self.p * self.N, self.q * self.N)

    def is_identity(self) -> bool:
        """Every block the identity?  Answered once per object (the blocks are not modified after construction)."""
        cached = self.__dict__.get("_identity")
        if cached is None:
            if self.p != self.q:
                cached = False
            else:
                eye = np.eye(self.p)[:, :, None]
                cached = bool(np.array_equal(self.data, np.broadcast_to(eye, self.data.shape)))
            self.__dict__["_identity"] = cached
        return cached

    def to_sparse(self) -> sp.csr_matrix:
        """Sparse image (reference: src/BlockMatrices.jl:690-710, zeros dropped)."""
        p, q, N = self.data.shape
        e = np.repeat(np.arange(N), p * q)
        c = np.tile(np.repeat(np.arange(q), p), N)
        r = np.tile(np.arange(p), q * N)
        v = self.data.reshape(-1, order="F")
        keep = v != 0
        return sp.csr_matrix(
            (v[keep], (e[keep] * p + r[keep], e[keep] * q + c[keep])), shape=self.shape
        )

    def matvec(self, z: np.ndarray) -> np.ndarray:
        """Host block matvec (setup/diagnostics only; reference: src/BlockMatrices.jl:583-601)."""
        p, q, N = self.data.shape
        zz = np.asarray(z, dtype=np.float64).reshape(N, q)
        return np.einsum("rce,ec->er", self.data, zz).reshape(-1)


@dataclass
class BlockColumn:
    """``D_fine[k]``: one BlockDiag in column block ``active_col`` of ``nu`` equal blocks
    (reference: src/BlockMatrices.jl:38-46, :673-674).  ``active_col`` is 0-based here."""

    active_block: BlockDiag
    active_col: int
    nu: int

    @property
    def shape(self):
        m, n = self.active_block.shape
        return (m, n * self.nu)

    def to_sparse(self) -> sp.csr_matrix:
        m, n = self.active_block.shape
        blocks = [None] * self.nu
        for j in range(self.nu):
            blocks[j] = self.active_block.to_sparse() if j == self.active_col else sp.csr_matrix((m, n))
        return sp.hstack(blocks, format="csr")


def block_column(op, active: int, nu: int):
    """reference: src/BlockMatrices.jl:666-674 (`_block_column`)."""
    if isinstance(op, BlockDiag):
        return BlockColumn(op, active, nu)
    # dense (spectral) operators: hcat of zeros with the operator in slot `active`
    op = np.asarray(op, dtype=np.float64)
    n = op.shape[0]
    out = np.zeros((n, n * nu))
    out[:, active * n:(active + 1) * n] = op
    return out
