"""Pre-split GEMM operands ("planes", csrc/gemm_pl.hip / include/dgcnn_hip.h).

A PlaneSet holds a (rows x cols) fp32 tensor as NPL planes of 16-bit terms in the chunk-major layout
    element (row, c) of plane p:  base + p * plane_stride + ((c / 8) * rows_alloc + row) * 16 + (c % 8) * 2   [bytes]
written once by the tensor's producer; the plane GEMM then moves the planes into LDS by DMA and does no split work.
torch tensors are only the memory holder (uint8 storage)."""
from __future__ import annotations

import torch

from . import _hip as H

F16X2 = 1                     # include/dgcnn_hip.h: DGCNN_PLANES_F16X2 (the only plane format)
KC, TR = 0, 1                 # DGCNN_PL_KC / DGCNN_PL_TR
NPLANES = {F16X2: 2}


def rows_alloc(rows):
    return (int(rows) + 63) // 64 * 64


class PlaneSet(object):
    """rows x cols (cols % 8 == 0) in `fmt`; `buf` = uint8 storage of NPL x (cols/8) x rows_alloc x 16 bytes."""

    def __init__(self, rows, cols, fmt=F16X2, device=None, zero=False):
        if cols % 8:
            raise ValueError("PlaneSet: cols must be a multiple of 8, got %d" % cols)
        self.rows, self.cols, self.fmt = int(rows), int(cols), int(fmt)
        self.ra = rows_alloc(rows)
        self.noct = cols // 8
        self.plane_stride = self.noct * self.ra * 16
        n = NPLANES[fmt] * self.plane_stride
        mk = torch.zeros if zero else torch.empty
        self.buf = mk(n, dtype=torch.uint8, device=device)
        self.c0 = 0                   # first channel of this view (views share `buf`)
        self.scale = None             # F16X2: device float holding the power-of-two scale the planes were multiplied by

    def ptr(self):
        return self.buf.data_ptr() + (self.c0 // 8) * self.ra * 16

    def cols_view(self, c0, c1):
        """The channel range [c0, c1) of the same rows (c0, c1 multiples of 8): shares the storage."""
        if c0 % 8 or c1 % 8 or not (0 <= c0 < c1 <= self.cols):
            raise ValueError("PlaneSet.cols_view: bad channel range [%d, %d) of %d" % (c0, c1, self.cols))
        v = PlaneSet.__new__(PlaneSet)
        v.__dict__.update(self.__dict__)
        v.c0, v.cols = self.c0 + c0, c1 - c0
        return v

    def fill_from(self, src, transpose=False):
        """Split the fp32 2-D view `src` into this set: plane (row, c) = src[row, c], or src[c, row] when transpose."""
        H.require_gpu(src)
        H.f32(src)
        r, c = (src.shape[1], src.shape[0]) if transpose else (src.shape[0], src.shape[1])
        if (r, c) != (self.rows, self.cols):
            raise ValueError("PlaneSet.fill_from: source is %s, the set holds (%d, %d)" % (tuple(src.shape), self.rows, self.cols))
        rs, cs = (1, H.ld2(src)) if transpose else (H.ld2(src), 1)
        if self.fmt == F16X2 and self.scale is None:
            self.measure_scale(src)
        H.call("dgcnn_split_planes_f32", src.data_ptr(), rs, cs, self.rows, self.cols, self.fmt, H._p(self.scale), self.ptr(),
               self.plane_stride, self.ra, tag="split_planes_kernel",
               work=4.0 * self.rows * self.cols + 2.0 * NPLANES[self.fmt] * self.ra * self.cols)
        return self


    def measure_scale(self, src, bound_mul=1.0):
        """F16X2: scale <- the power of two that brings max |src| (times bound_mul) into [2^14, 2^15), computed on the device."""
        self.scale = torch.empty(1, dtype=torch.float32, device=src.device)
        ws = torch.empty(1, dtype=torch.int32, device=src.device)
        if src.shape[1] % 4 == 0 and H.ld2(src) % 4 == 0 and src.data_ptr() % 16 == 0 and src.stride(1) == 1:
            v = src
        else:                                   # (small odd-shaped tensors: weights with an unaligned view)
            v = src.contiguous().view(1, -1)
            pad = (-v.shape[1]) % 4
            if pad:
                v = torch.cat([v, v.new_zeros(1, pad)], 1)
        H.call("dgcnn_planes_scale_f32", v.data_ptr(), H.ld2(v), v.shape[0], v.shape[1], float(bound_mul), self.scale.data_ptr(),
               ws.data_ptr())
        return self


def from_f32(src, fmt=F16X2, transpose=False):
    r, c = (src.shape[1], src.shape[0]) if transpose else (src.shape[0], src.shape[1])
    return PlaneSet(r, c, fmt, device=src.device).fill_from(src, transpose)


def gemm(form, A, B, C, beta=0.0, gbias=None, rpg=0, stats=None, ws=None, colmax=None, colmax_rpg=0):
    """C (+)= A B^T-style product of two plane sets.  KC: C is (A.rows, B.rows), reduction over the channels;
    TR: C is (A.cols, B.cols), reduction over the rows (X^T dY)."""
    if A.fmt != B.fmt:
        raise ValueError("plane GEMM: operands in different formats")
    if form == KC:
        M, N, K = A.rows, B.rows, A.cols
        if B.cols != K:
            raise ValueError("plane GEMM (KC): A has %d channels, B %d" % (K, B.cols))
    else:
        M, N, K = A.cols, B.cols, A.rows
        if B.rows != K:
            raise ValueError("plane GEMM (TR): A has %d rows, B %d" % (K, B.rows))
    assert tuple(C.shape) == (M, N), (tuple(C.shape), M, N)
    H.call("dgcnn_gemm_planes_f32", form, A.fmt, M, N, K, A.ptr(), A.plane_stride, A.ra, B.ptr(), B.plane_stride, B.ra,
           H._p(A.scale), H._p(B.scale), C.data_ptr(), H.ld2(C), float(beta), H._p(gbias), 0 if gbias is None else H.ld2(gbias), int(rpg),
           H._p(stats), H._p(colmax), int(colmax_rpg), H._p(ws), 0 if ws is None else ws.numel(),
           tag="gemm_pl_kernel<%s,%s>" % ("KC" if form == KC else "TR", "f16x2"),
           work=2.0 * M * N * K)
